#!/bin/bash
# tools/variant_all.sh <name> [hipcc flags]: a variant library with EVERY translation unit rebuilt under the flags
# (flags that change what rtr_upload_scene builds as well as what the kernels do)
cd "$(dirname "$0")/.." || exit 1
name=$1; shift
mkdir -p ray_tracing-rendering_amd/variants build/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Iinclude -Iray_tracing-rendering_amd/csrc"
hipcc $F "$@" -c ray_tracing-rendering_amd/csrc/rtr_capi.hip -o build/obj/capi_$name.o 2>/tmp/variant_${name}_c.log &
hipcc $F "$@" -c ray_tracing-rendering_amd/csrc/rtr_wavefront.hip -o build/obj/wavefront_$name.o 2>/tmp/variant_${name}_w.log &
for g in 0 1 2; do
  hipcc $F "$@" -DRTR_MEGA_GROUP=$g -c ray_tracing-rendering_amd/csrc/rtr_mega.hip -o build/obj/mega${g}_$name.o -Rpass-analysis=kernel-resource-usage 2>/tmp/variant_${name}_$g.log &
done
wait
grep -l " error" /tmp/variant_${name}_*.log | head -3 | xargs -r tail -5
hipcc --offload-arch=gfx950 -shared -fPIC build/obj/capi_$name.o build/obj/mega0_$name.o build/obj/mega1_$name.o build/obj/mega2_$name.o build/obj/wavefront_$name.o \
  -o ray_tracing-rendering_amd/variants/librtr_hip_$name.so || exit 1
cat /tmp/variant_${name}_[012].log > /tmp/variant_$name.log
python3 tools/kres.py /tmp/variant_$name.log | grep "k_megaILi1ELi3ELi1\|k_megaILi4ELi3ELi2\|k_megaILi4ELi4ELi0\|k_megaILi4ELi4ELi2" | sed "s/^/$name: /"
