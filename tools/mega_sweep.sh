#!/bin/bash
# tools/mega_sweep.sh <tag> "<variants>" "<workloads>" [bench args]: time megakernel variants
tag=$1; variants=$2; wl=$3; shift; shift; shift
for v in $variants; do
  export RTR_HIP_LIBRARY=$(pwd)/ray_tracing-rendering_amd/variants/librtr_hip_$v.so
  [ "$v" = "base" ] && unset RTR_HIP_LIBRARY
  for w in $wl; do
    timeout -k 10 200 python bench.py --workload $w --extras none --steps 2 --warmup 1 --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%-8s %-16s %9.1f Msamples/s  %8.2f ms' % ('$v', '$w', d['value'], d['ms_per_step']))
" | tee -a gpurun_out/$tag.txt
  done
done
