#!/bin/bash
# tools/pmc_ab.sh <tag> "<variants>" <workload>: SQ counter passes of one megakernel render per library variant
# ("base" = the in-tree library, anything else = ray_tracing-rendering_amd/variants/librtr_hip_<name>.so)
tag=$1; variants=$2; w=$3; shift; shift; shift
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { v=$1; n=$2; shift; shift
  rm -rf /tmp/pab_${v}_$n && rocprofv3 --pmc "$@" -d /tmp/pab_${v}_$n -o pmc --output-format csv -- python3 $root/bench.py --workload $w --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>$root/gpurun_out/${tag}_${v}_$n.err
  cp $(find /tmp/pab_${v}_$n -name '*counter_collection.csv' | head -1) $root/gpurun_out/${tag}_${v}_$n.csv || echo "pass $v $n failed"
}
for v in $variants; do
  export RTR_HIP_LIBRARY=$root/ray_tracing-rendering_amd/variants/librtr_hip_$v.so
  [ "$v" = "base" ] && unset RTR_HIP_LIBRARY
  run $v a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
  run $v b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM
  run $v c SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
  run $v d GRBM_GUI_ACTIVE GRBM_COUNT
  rm -rf /tmp/pab_${v}_kt && rocprofv3 --kernel-trace --stats -d /tmp/pab_${v}_kt -o kt --output-format csv -- python3 $root/bench.py --workload $w --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
  cp $(find /tmp/pab_${v}_kt -name '*kernel_stats.csv' | head -1) $root/gpurun_out/${tag}_${v}_kt.csv
done
python3 - $tag $root "$variants" <<'PY'
import csv, sys, collections, glob
tag, root, variants = sys.argv[1], sys.argv[2], sys.argv[3].split()
names = []
table = {}
for v in variants:
    acc = collections.defaultdict(float)
    for f in glob.glob('%s/gpurun_out/%s_%s_[abcd].csv' % (root, tag, v)):
        for r in csv.DictReader(open(f)):
            if 'k_mega' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value'])
    for f in glob.glob('%s/gpurun_out/%s_%s_kt.csv' % (root, tag, v)):
        for r in csv.DictReader(open(f)):
            if 'k_mega' in r['Name']:
                acc['kernel_ns'] = float(r['AverageNs'])
    table[v] = acc
    for k in acc:
        if k not in names: names.append(k)
with open('%s/gpurun_out/%s_ab.txt' % (root, tag), 'w') as o:
    o.write('%-28s' % 'counter' + ''.join('%16s' % v for v in variants) + '\n')
    for k in sorted(names):
        o.write('%-28s' % k + ''.join('%16.6g' % table[v].get(k, float('nan')) for v in variants) + '\n')
print(open('%s/gpurun_out/%s_ab.txt' % (root, tag)).read())
PY
