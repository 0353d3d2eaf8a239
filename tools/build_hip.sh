#!/bin/bash
# Rebuild librtr_hip.so for gfx950 and print the per-kernel register report.
cd "$(dirname "$0")/.." || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value \
  -Iinclude -Iray_tracing-rendering_amd/csrc ray_tracing-rendering_amd/csrc/rtr_capi.hip \
  -o ray_tracing-rendering_amd/librtr_hip.so -Rpass-analysis=kernel-resource-usage "$@" 2>/tmp/rtr_build.log
rc=$?
grep -E "error|warning: v" -A6 /tmp/rtr_build.log | head -40
python3 tools/kres.py /tmp/rtr_build.log
exit $rc
