export RTR_HIP_LIBRARY=$(pwd)/ray_tracing-rendering_amd/variants/librtr_hip_stats.so
for w in final_rr cornell_mis; do echo == $w; python bench.py --workload $w --extras none --pipeline wavefront --steps 1 --warmup 0 --no-cpu-baseline --no-parity --spp 100 2>&1 | grep -v "^{" | grep -v amdgpu.ids; done
