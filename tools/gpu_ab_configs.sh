#!/bin/bash
# bench.py --no-cpu per config and library variant, interleaved: bash tools/gpu_ab_configs.sh <outdir> "<configs>" tag ...   (tag default = the shipped library)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; CFGS=$2; shift; shift
mkdir -p $OUT
for C in $CFGS; do
  for TAG in "$@"; do
    LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$TAG.so
    [ "$TAG" = "default" ] && LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip.so
    PTR_HIP_LIB=$LIB timeout -k 10 280 python3 -u bench.py --config $C --no-cpu --steps ${STEPS:-6} --warmup 2 > $OUT/bench_c${C}_$TAG.json 2> $OUT/bench_c${C}_$TAG.err || { echo "config $C $TAG failed"; tail -5 $OUT/bench_c${C}_$TAG.err; exit 1; }
    python3 - $OUT/bench_c${C}_$TAG.json $C $TAG <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels_ms_exclusive_per_step"]
print("config %s %-8s %.1f Mrays/s %.1f ms/step | tracer %.3f ms/launch | %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], {a: round(b,1) for a,b in k.items() if b > 1}))
PY
  done
done
