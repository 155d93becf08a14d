#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...> -- A/B build of the library into build_variants/libcrt_<name>.so:
# kernels.hip recompiled with the extra flags, every other object taken from the in-tree build (run make first).
# Selected at run time with CRT_LIB=...; development experiments only.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../cudaraytracer_amd/csrc"
mkdir -p ../../build_variants/obj_$NAME
O=../../build_variants/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $*"
/opt/rocm/bin/hipcc $FLAGS -c kernels.hip -o $O/kernels.o -Rpass-analysis=kernel-resource-usage > $O/kernels.log 2>&1 || { grep -A6 error $O/kernels.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libcrt_$NAME.so $O/kernels.o lbvh.o refrank.o noalloc.o crt_api.o bvh_build.o scene.o mgpu.o compat.o -ldl -lpthread
echo built build_variants/libcrt_$NAME.so
