#!/bin/bash
# tools/final_round.sh <tag>: the measurements of a round that are not counter passes (run on the GPU box from the repo root,
# after tools/profile_round.sh <tag> has written profiles/<tag>_counts.json for the same sources):
#   the default bench line, the many-instance and hollow-sphere scenes, the region profiles (-DRTR_REGION_PROFILE variant)
tag=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err; echo "bench rc=$?"
for spp in 16 64; do
  echo "== tools/time_random.py, 640x360, spp $spp"
  SPP=$spp timeout -k 10 300 python tools/time_random.py 2>&1 | grep objects
done > gpurun_out/${tag}_time_random.txt
echo "== RTR_TOP_MIN=100000 (every sub-scene scanned in instance order), spp 64" >> gpurun_out/${tag}_time_random.txt
RTR_TOP_MIN=100000 OBJECTS=120,400,800 SPP=64 timeout -k 10 300 python tools/time_random.py 2>&1 | grep objects >> gpurun_out/${tag}_time_random.txt
timeout -k 10 300 python tools/bench_scenes.py 2>&1 | grep scene > gpurun_out/${tag}_bench_scenes.txt
bash tools/region_profile.sh ${tag}_regions "cornell_mis cornell_literal final_rr final_mis mis_spheres" > /dev/null 2>&1
export RTR_HIP_LIBRARY=$(pwd)/ray_tracing-rendering_amd/variants/librtr_hip_prof.so RTR_REGION_PROFILE=1
(echo "== tools/time_random.py OBJECTS=800 SPP=64 (305 instances: the per-lane top tree), integrators 1 and 4"
 OBJECTS=800 SPP=64 timeout -k 10 300 python tools/time_random.py 2>&1 | awk '/wave cycles in all/ {last = ""} /region profile\]/ {last = last $0 "\n"} /^objects/ {printf "%s%s\n", last, $0; last = ""}') > gpurun_out/${tag}_top_regions.txt
(echo "== tools/bench_scenes.py (scenes 1 / 35 / 30 / 40 / 24 / 23)"
 timeout -k 10 300 python tools/bench_scenes.py 2>&1 | awk '/wave cycles in all/ {last = ""} /region profile\]/ {last = last $0 "\n"} /^scene/ {printf "%s%s\n", last, $0; last = ""}') > gpurun_out/${tag}_scene_regions.txt
tail -3 gpurun_out/${tag}_time_random.txt; cat gpurun_out/${tag}_bench_scenes.txt; python - <<PY
import json
for l in open("gpurun_out/bench_${tag}.json"):
    j = json.loads(l)
    print(j["metric"], j["value"], j["roofline"]["frac"], j["roofline"].get("stale"), j["parity_ok"], j["cpu_baseline"]["value"])
    for e in j["extra"]:
        print(" ", e["metric"], e["value"], e["roofline"].get("stale"), e.get("parity", {}).get("rel_l2_timed_framebuffer_vs_oracle"))
PY
