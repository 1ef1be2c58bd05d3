#!/bin/bash
# Builds a variant of the library with extra device flags, for A/B timing with scripts/gpu_variants.py:
#   scripts/build_variant.sh <name> "<extra hipcc flags>"   ->  raytracer_amd/csrc/build/variants/<name>.so
# Host objects and the kernels that do not change are shared with the main build (run make first).
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
name=$1; flags=$2
out=build/variants; mkdir -p $out/$name
DEV="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
/opt/rocm/bin/hipcc $DEV -ffp-contract=off -DRT_FLAVOUR_STRICT $flags -c rt_kernels.hip -o $out/$name/strict.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/$name.so $out/$name/strict.o build/rt_kernels_fast.o build/rt_noalloc.o build/rt_comm.o build/rt_host.o build/rt_bvh.o build/rt_scene.o -Wl,-rpath,/opt/rocm/lib -ldl
rm -rf $out/$name
echo built $out/$name.so
