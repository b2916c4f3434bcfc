#!/bin/bash
# Builds a variant of libvrt.so with extra compiler flags for A/B runs (select it with VRT_LIB):
#   tools/build_variant.sh <name> [flags...]   ->  raytracing_amd/libvrt_<name>.so
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/raytracing_amd/csrc
tmp=/tmp/vrt_variant_$name
mkdir -p $tmp
flags="${VRT_OPT:--O3} -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-value $*"
pids=()
for f in vrt_api vrt_kernels vrt_bvh vrt_wavefront vrt_blockwf vrt_multi; do
  /opt/rocm/bin/hipcc $flags -c $src/$f.hip -o $tmp/$f.o & pids+=($!)
done
for f in vrt_kernels vrt_wavefront vrt_blockwf; do   # the same sources once more, box test in its fused form
  /opt/rocm/bin/hipcc $flags -DVRT_SLAB_FMA=1 -c $src/$f.hip -o $tmp/${f}_fused.o & pids+=($!)
done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -c $src/vrt_sampler.cpp -o $tmp/vrt_sampler.o & pids+=($!)
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/raytracing_amd/libvrt_$name.so $tmp/*.o -ldl -lpthread && echo built libvrt_$name.so
