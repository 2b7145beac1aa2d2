#!/bin/bash
# A/B of library variants on the bench line itself (six passes overlapping), interleaved and repeated:
#   bash tools/gpu_ab_bench.sh <outdir> <reps> tag[:ENV=VAL,...] ...     (BENCH_ARGS="--config 2 --steps 4" for another config)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; REPS=$2; shift; shift
mkdir -p $OUT
for R in $(seq 1 $REPS); do
  for V in "$@"; do
    TAG=${V%%:*}; ENVS=""
    if [[ "$V" == *:* ]]; then ENVS=$(echo "${V#*:}" | tr ',' ' '); fi
    NAME=$(echo "$V" | tr ':=,' '___')
    LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$TAG.so
    [ "$TAG" = "default" ] && LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip.so
    env PTR_HIP_LIB=$LIB $ENVS timeout -k 10 280 python3 bench.py --no-cpu ${BENCH_ARGS:-} > $OUT/${NAME}_$R.json 2> $OUT/${NAME}_$R.err || { echo "$V failed"; tail -3 $OUT/${NAME}_$R.err; exit 1; }
    python3 - $OUT/${NAME}_$R.json "$V" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%-28s %.1f Mrays/s %.2f ms/step | tracer %.3f ms/launch | shade launch %.3f ms | %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], d["roofline_shade"]["avg_launch_ms"],
    {k: round(v, 1) for k, v in d["kernels_ms_exclusive_per_step"].items() if v >= 1}))
PY
  done
done
