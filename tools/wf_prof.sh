#!/bin/bash
# kernel-trace stats + SQ pass of one wavefront render: tools/wf_prof.sh <tag> <workload> [bench args]
tag=$1; w=$2; shift; shift
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag && rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt --output-format csv -- python3 $root/bench.py --workload $w --extras none --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > $root/gpurun_out/${tag}_bench.json 2>/dev/null
cp $(find /tmp/kt_$tag -name '*kernel_stats.csv' | head -1) $root/gpurun_out/${tag}_kernel_stats.csv
cut -c1-150 $root/gpurun_out/${tag}_kernel_stats.csv | head -12
rm -rf /tmp/sq_$tag && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU -d /tmp/sq_$tag -o pmc --output-format csv -- python3 $root/bench.py --workload $w --extras none --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline --no-parity "$@" > /dev/null 2>&1
python3 - $tag $root <<'PY'
import csv, glob, sys, collections
tag, root = sys.argv[1], sys.argv[2]
f = glob.glob('/tmp/sq_%s/**/*counter_collection.csv' % tag, recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']] += float(r['Counter_Value'])
with open('%s/gpurun_out/%s_sq.txt' % (root, tag), 'w') as o:
    for k, c in acc.items():
        wc = c.get('SQ_WAVE_CYCLES', 0)
        if wc < 1e6: continue
        line = '%s\n  wait %.1f%%  wait_inst %.1f%%  issue %.1f%%  valu_active %.1f%%  INSTS_VALU %.4g  lane_util %.1f%%\n' % (
            k, 100*c['SQ_WAIT_ANY']/wc, 100*c['SQ_WAIT_INST_ANY']/wc, 100*c['SQ_ACTIVE_INST_ANY']/wc,
            100*c['SQ_ACTIVE_INST_VALU']/wc, c['SQ_INSTS_VALU'], 100*c['SQ_THREAD_CYCLES_VALU']/max(c["SQ_INSTS_VALU"],1)/64)
        o.write(line); print(line)
PY
