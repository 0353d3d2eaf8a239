#!/bin/bash
# like tools/variant.sh, but rebuilds the three megakernel units with the given flags
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p ray_tracing-rendering_amd/variants build/obj
for g in 0 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -DRTR_MEGA_GROUP=$g -Iinclude -Iray_tracing-rendering_amd/csrc \
    -c ray_tracing-rendering_amd/csrc/rtr_mega.hip -o build/obj/mega${g}_$name.o -Rpass-analysis=kernel-resource-usage 2>/tmp/variant_${name}_$g.log &
done
wait
grep -l "error" /tmp/variant_${name}_*.log | head -3 | xargs -r tail -5
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/capi.o build/obj/mega0_$name.o build/obj/mega1_$name.o build/obj/mega2_$name.o build/obj/wavefront.o \
  -o ray_tracing-rendering_amd/variants/librtr_hip_$name.so || exit 1
cat /tmp/variant_${name}_*.log > /tmp/variant_$name.log
python3 tools/kres.py /tmp/variant_$name.log | grep "k_megaILi1ELi3ELi1\|k_megaILi4ELi3ELi2\|k_megaILi4ELi4ELi0\|k_megaILi4ELi4ELi2" | sed "s/^/$name: /"
