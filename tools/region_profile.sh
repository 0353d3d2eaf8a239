#!/bin/bash
# tools/region_profile.sh <tag> "<workloads>": where the megakernel's wave cycles go, per code region
# (a -DRTR_REGION_PROFILE variant library: tools/variant_mega.sh prof -DRTR_REGION_PROFILE; see rt_device.h RT_REGION)
tag=$1; wl=$2
mkdir -p gpurun_out
export RTR_HIP_LIBRARY=$(pwd)/ray_tracing-rendering_amd/variants/librtr_hip_prof.so
export RTR_REGION_PROFILE=1
: > gpurun_out/$tag.txt
for w in $wl; do
  echo "== $w" >> gpurun_out/$tag.txt
  timeout -k 10 300 python bench.py --workload $w --extras none --steps 1 --warmup 0 --no-cpu-baseline --no-parity 2>&1 >/dev/null | grep "region profile" >> gpurun_out/$tag.txt
done
cat gpurun_out/$tag.txt
