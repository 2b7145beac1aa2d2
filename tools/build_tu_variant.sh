#!/bin/bash
# A kernel-variant library that differs from the default build only in ONE translation unit's compile flags:
#   bash tools/build_tu_variant.sh <unit, e.g. rt_walk> <tag> "<-D flags>"   ->  pathtracer_amd/lib/libptr_hip_<tag>.so   (load with PTR_HIP_LIB)
set -e
cd "$(dirname "$0")/../pathtracer_amd"
TU=$1; TAG=$2; FLAGS=$3
mkdir -p build_var
HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-invalid-offsetof"
/opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -Rpass-analysis=kernel-resource-usage -c -o build_var/${TU}_$TAG.o csrc/$TU.hip 2> build_var/${TU}_$TAG.log
grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" build_var/${TU}_$TAG.log | sed "s/.*remark: *//; s/ \[-R.*//" | paste - - - - - | sed "s/^/[$TAG] /"
OBJS=$(ls build/*.o | grep -v "build/$TU.o")
/opt/rocm/bin/hipcc $HIPFLAGS -shared -o lib/libptr_hip_$TAG.so $OBJS build_var/${TU}_$TAG.o
