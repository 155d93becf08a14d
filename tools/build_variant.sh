#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...> -- A/B build of the library into build_variants/libcrt_<name>.so
# (selected at run time with CRT_LIB=...; development experiments only)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../cudaraytracer_amd/csrc"
mkdir -p ../../build_variants/obj_$NAME
O=../../build_variants/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value $*"
rm -f $O/kernels.o
/opt/rocm/bin/hipcc $FLAGS -c kernels.hip -o $O/kernels.o -Rpass-analysis=kernel-resource-usage > $O/kernels.log 2>&1 || { grep -A6 error $O/kernels.log; exit 1; }
grep -A12 "k_trace8ILi0ELb0" $O/kernels.log | grep -E "VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//' | tr '\n' ' '; echo
/opt/rocm/bin/hipcc $FLAGS -c lbvh.hip -o $O/lbvh.o
for f in crt_api bvh_build scene; do /opt/rocm/bin/hipcc $FLAGS -c $f.cpp -o $O/$f.o; done
/opt/rocm/bin/hipcc $FLAGS -c compat/compat.cpp -o $O/compat.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libcrt_$NAME.so $O/*.o
