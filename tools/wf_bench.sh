#!/bin/bash
# quick timing of the wavefront pipeline on the BASELINE workloads (GPU box): tools/wf_bench.sh <tag> [workloads...]
tag=${1:-wf}; shift
wl=${@:-final_rr final_mis cornell_mis mis_spheres cornell_literal}
mkdir -p gpurun_out
for w in $wl; do
  timeout -k 10 200 python bench.py --workload $w --extras none --pipeline wavefront --steps 2 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('%-16s %9.1f Msamples/s  %8.2f ms  chunks %d  hbm_eq %.3f' % ('$w', d['value'], d['ms_per_step'], d['config']['spp_chunks'], d['roofline'].get('frac',0)))
" | tee -a gpurun_out/$tag.txt
done
