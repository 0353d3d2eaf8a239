#!/bin/bash
# tools/pmc_detail.sh <tag> <workload> [bench args]: latency / cache counters of one megakernel render
tag=$1; w=$2; shift; shift
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  rm -rf /tmp/pd_$n && rocprofv3 --pmc "$@" -d /tmp/pd_$n -o pmc --output-format csv -- python3 $root/bench.py --workload $w --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-parity $EXTRA > /dev/null 2>$root/gpurun_out/${tag}_$n.err
  cp $(find /tmp/pd_$n -name '*counter_collection.csv' | head -1) $root/gpurun_out/${tag}_$n.csv || echo "pass $n failed"
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU
run b SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run d SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACCUM_PREV_HIRES SQ_WAVE_CYCLES
python3 - $tag $root <<'PY'
import csv, sys, collections, glob
tag, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
for f in glob.glob('%s/gpurun_out/%s_[abcd].csv' % (root, tag)):
    for r in csv.DictReader(open(f)):
        if 'k_mega' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
with open('%s/gpurun_out/%s_detail.txt' % (root, tag), 'w') as o:
    for k in sorted(acc): o.write('%-32s %.6g\n' % (k, acc[k]))
print(open('%s/gpurun_out/%s_detail.txt' % (root, tag)).read())
PY
