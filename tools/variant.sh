#!/bin/bash
# Build a variant of librtr_hip.so that differs only in the wavefront translation unit:
#   tools/variant.sh <name> [hipcc flags, e.g. -DRTR_MACHINE_WAIT=32]  ->  ray_tracing-rendering_amd/variants/librtr_hip_<name>.so
# (use with RTR_HIP_LIBRARY=... ; the other objects come from the last full build in build/obj)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p ray_tracing-rendering_amd/variants build/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -Iinclude -Iray_tracing-rendering_amd/csrc \
  -c ray_tracing-rendering_amd/csrc/rtr_wavefront.hip -o build/obj/wavefront_$name.o -Rpass-analysis=kernel-resource-usage 2>/tmp/variant_$name.log || { tail -20 /tmp/variant_$name.log; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/capi.o build/obj/mega_mis.o build/obj/mega_rr_path.o build/obj/mega_pbr_nee.o build/obj/wavefront_$name.o \
  -o ray_tracing-rendering_amd/variants/librtr_hip_$name.so || exit 1
python3 tools/kres.py /tmp/variant_$name.log | grep "wf_extend\|wf_connect" | sed "s/^/$name: /"
