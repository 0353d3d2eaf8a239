#!/bin/bash
# SQ counter pass over one bench workload (run on the GPU box through gpurun):
#   tools/pmc_sq.sh <tag> <bench args...>   ->  gpurun_out/pmc_<tag>.txt
tag=$1; shift
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU \
  -d /tmp/pmc_$tag -o pmc --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $root/gpurun_out/pmc_${tag}_bench.log 2>&1 || exit 1
python3 - $tag $root <<'PY'
import csv, glob, sys, collections
tag, root = sys.argv[1], sys.argv[2]
f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % tag, recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
with open('%s/gpurun_out/pmc_%s.txt' % (root, tag), 'w') as o:
    for k, c in acc.items():
        if c.get('SQ_WAVE_CYCLES', 0) < 1e6: continue
        wc = c['SQ_WAVE_CYCLES']
        line = '%s\n  wait %.1f%%  wait_inst %.1f%%  active %.1f%%  valu_active %.1f%%  INSTS_VALU %.4g  lane_util %.1f%%\n' % (
            k, 100*c['SQ_WAIT_ANY']/wc, 100*c['SQ_WAIT_INST_ANY']/wc, 100*c['SQ_ACTIVE_INST_ANY']/wc,
            100*c['SQ_ACTIVE_INST_VALU']/wc, c['SQ_INSTS_VALU'], 100*c['SQ_THREAD_CYCLES_VALU']/max(c["SQ_INSTS_VALU"],1)/64)
        o.write(line); print(line)
PY
tail -1 $root/gpurun_out/pmc_${tag}_bench.log
