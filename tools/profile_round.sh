#!/bin/bash
# Round-end evidence, run on the GPU box through gpurun from the repo root:
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{bench_*.json, kernel_stats.csv, pmc_sq.txt, traffic_*.csv}
# One rocprofv3 mode per run (never --pmc together with a trace), program directly after "--".
tag=${1:-r01_final}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
set -e
# 1. the default bench line (with cpu_baseline) and the other workloads
python3 $root/bench.py --steps 3 --warmup 1 > $out/bench_cornell_mis.json 2> $out/bench_cornell_mis.err
echo "bench default done"
for w in cornell_rr cornell_literal mis_spheres final_rr final_mis; do
  python3 $root/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$w.json 2>/dev/null
done
python3 $root/bench.py --pipeline wavefront --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_cornell_mis_wavefront.json 2>/dev/null
echo "bench lines done"
# 2. kernel trace of the same default command
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/kt_bench.json 2>/dev/null
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
echo "kernel trace done"
# 3. SQ counters (own pass)
rm -rf /tmp/sq && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU \
  -d /tmp/sq -o sq --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/sq -name '*counter_collection.csv' | head -1) $out/pmc_sq_raw.csv
# 4. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for cn in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr && rocprofv3 --pmc $cn -d /tmp/tr -o tr --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  cp $(find /tmp/tr -name '*counter_collection.csv' | head -1) $out/pmc_${cn}_raw.csv
  rm -rf /tmp/tr && rocprofv3 --pmc $cn -d /tmp/tr -o tr --output-format csv -- python3 $root/bench.py --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  cp $(find /tmp/tr -name '*counter_collection.csv' | head -1) $out/pmc_${cn}_wavefront_raw.csv
done
rm -rf /tmp/ktw && rocprofv3 --kernel-trace --stats -d /tmp/ktw -o kt --output-format csv -- python3 $root/bench.py --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/ktw -name '*kernel_stats.csv' | head -1) $out/kernel_stats_wavefront.csv
echo "pmc done"
python3 - $out <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
def agg(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0][:48]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
    return acc, n
acc, n = agg(out + '/pmc_sq_raw.csv')
with open(out + '/pmc_sq.txt', 'w') as f:
    for k, c in acc.items():
        wc = c.get('SQ_WAVE_CYCLES', 0)
        if wc < 1e6: continue
        f.write('%s\n  wait %.1f%%  issue %.1f%%  valu %.1f%%  INSTS_VALU %.4g  lane_util %.1f%%\n' % (
            k, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_VALU'] / wc,
            c['SQ_INSTS_VALU'], 100 * c['SQ_THREAD_CYCLES_VALU'] / max(c['SQ_INSTS_VALU'], 1) / 64))
tr = collections.defaultdict(float)
with open(out + '/pmc_hbm_traffic.csv', 'w') as f:
    f.write('pipeline,kernel,dispatches,counter,sum_KB\n')
    for pipe, suffix in (('megakernel', ''), ('wavefront', '_wavefront')):
        for cn in ('FETCH_SIZE', 'WRITE_SIZE'):
            a, nn = agg(out + '/pmc_%s%s_raw.csv' % (cn, suffix))
            for k, c in a.items():
                f.write('%s,%s,%d,%s,%.0f\n' % (pipe, k.replace(',', ';'), nn[(k, cn)], cn, c[cn]))
                if k.startswith('void k_mega') or k.startswith('k_resolve') or 'wf_' in k:
                    tr[pipe] += c[cn] * 1024
json.dump({'cornell_mis': dict(tr)}, open(out + '/traffic.json', 'w'))
print(open(out + '/pmc_sq.txt').read())
print(open(out + '/traffic.json').read())
PY
head -4 $out/kernel_stats.csv
cat $out/bench_*.json | cut -c1-150
