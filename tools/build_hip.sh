#!/bin/bash
# Rebuild librtr_hip.so for gfx950 (translation units in parallel) and print the per-kernel register report.
#   tools/build_hip.sh [extra hipcc flags, e.g. -DRTR_MACHINE_WAIT=16]
cd "$(dirname "$0")/.." || exit 1
python3 - "$@" <<'PY'
import importlib, sys
b = importlib.import_module("ray_tracing-rendering_amd.build")
try:
    out, log = b.build_hip(force=True, verbose=True, extra_flags=tuple(sys.argv[1:]))
except RuntimeError as e:
    print(e); sys.exit(1)
open("/tmp/rtr_build.log", "w").write(log)
PY
rc=$?
[ $rc -eq 0 ] && python3 tools/kres.py /tmp/rtr_build.log
exit $rc
