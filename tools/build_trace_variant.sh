#!/bin/bash
# A kernel-variant library that differs from the default build only in rt_trace.hip's compile flags:
#   bash tools/build_trace_variant.sh <tag> "<-D flags>"   ->  pathtracer_amd/lib/libptr_hip_<tag>.so   (load with PTR_HIP_LIB)
set -e
cd "$(dirname "$0")/../pathtracer_amd"
TAG=$1; FLAGS=$2
mkdir -p build_var
HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-invalid-offsetof"
/opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -Rpass-analysis=kernel-resource-usage -c -o build_var/rt_trace_$TAG.o csrc/rt_trace.hip 2> build_var/rt_trace_$TAG.log
grep -A8 "k_ptraceILb0ELb0" build_var/rt_trace_$TAG.log | grep -E "VGPRs:|ScratchSize|VGPRs Spill|Occupancy" | sed "s/.*remark: *//; s/ \[-R.*//" | tr '\n' ' '; echo " [$TAG]"
OBJS=$(ls build/*.o | grep -v "build/rt_trace.o")
/opt/rocm/bin/hipcc $HIPFLAGS -shared -o lib/libptr_hip_$TAG.so $OBJS build_var/rt_trace_$TAG.o
