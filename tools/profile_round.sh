#!/bin/bash
# Round evidence, run on the GPU box through gpurun from the repo root:
#   tools/profile_round.sh <tag> [workloads...]   -> gpurun_out/<tag>/...  and  gpurun_out/<tag>/<tag>_counts.json
# One rocprofv3 mode per run (never --pmc together with a trace), program directly after "--".
# Counter passes render every workload exactly once (--steps 1 --warmup 0, no parity crops, no CPU leg), so the
# k-th render dispatch of a pass belongs to the k-th workload.
tag=${1:-r02}
shift
wl=${@:-cornell_mis cornell_literal final_rr final_mis mis_spheres c5_shard}
first=$(echo $wl | cut -d' ' -f1)
rest=$(echo $wl | cut -d' ' -f2- | tr ' ' ',')
[ "$first" = "$(echo $wl)" ] && rest=none
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
set -e
once="--workload $first --extras $rest --steps 1 --warmup 0 --extra-steps 1 --extra-warmup 0 --no-cpu-baseline --no-parity"
# 1. kernel trace of the default headline command (its kernel_stats.csv average is what roofline.kernel_ms must agree with)
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --extras none --no-cpu-baseline --no-parity > $out/kt_bench.json 2>/dev/null
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
echo "kernel trace (headline) done"
# 2. kernel trace of one render per workload (per-dispatch durations)
rm -rf /tmp/kt2 && rocprofv3 --kernel-trace -d /tmp/kt2 -o kt --output-format csv -- python3 $root/bench.py $once > $out/once_bench.json 2>/dev/null
cp $(find /tmp/kt2 -name '*kernel_trace.csv' | head -1) $out/kernel_trace_once.csv
echo "kernel trace (all workloads) done"
# 3. counter passes (SQ: 8 slots per pass; FETCH_SIZE / WRITE_SIZE each on their own)
pass() { # name, counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n && rocprofv3 --pmc "$@" -d /tmp/pmc_$n -o pmc --output-format csv -- python3 $root/bench.py $once > /dev/null 2>$out/pmc_$n.err
  cp $(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1) $out/pmc_$n.csv
  echo "pmc pass $n done"
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM
pass f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA || true
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# 4. the wavefront pipeline on the headline and on scene 9: kernel trace, SQ, FETCH, WRITE
for w in cornell_mis final_rr; do
  wfonce="--workload $w --extras none --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline --no-parity"
  rm -rf /tmp/ktw && rocprofv3 --kernel-trace --stats -d /tmp/ktw -o kt --output-format csv -- python3 $root/bench.py $wfonce > $out/wf_${w}_bench.json 2>/dev/null
  cp $(find /tmp/ktw -name '*kernel_stats.csv' | head -1) $out/wf_${w}_kernel_stats.csv
  for cn in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcw && rocprofv3 --pmc $cn -d /tmp/pmcw -o pmc --output-format csv -- python3 $root/bench.py $wfonce > /dev/null 2>&1
    cp $(find /tmp/pmcw -name '*counter_collection.csv' | head -1) $out/wf_${w}_$cn.csv
  done
  rm -rf /tmp/pmcw && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU -d /tmp/pmcw -o pmc --output-format csv -- python3 $root/bench.py $wfonce > /dev/null 2>&1
  cp $(find /tmp/pmcw -name '*counter_collection.csv' | head -1) $out/wf_${w}_sq.csv
  echo "wavefront passes $w done"
done
# 5. FETCH_SIZE / WRITE_SIZE calibration on a known 8-byte-per-lane stream (the wavefront stages' access shape)
for cn in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal && rocprofv3 --pmc $cn -d /tmp/cal -o pmc --output-format csv -- python3 $root/tools/calibrate_stream.py > /dev/null 2>&1
  cp $(find /tmp/cal -name '*counter_collection.csv' | head -1) $out/calib_$cn.csv
done
echo "calibration done"
cd $root && python3 tools/issue_rates.py > $out/issue_rates.txt 2>&1 && cp gpurun_out/issue_rates.json $out/${tag}_issue_rates.json
python3 $root/tools/pmc_counts.py $out $tag $wl
