#!/bin/bash
# A/B of library variants on one serial pass, interleaved and repeated: bash tools/gpu_ab_serial.sh <outdir> <reps> tag ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; REPS=$2; shift; shift
mkdir -p $OUT
for R in $(seq 1 $REPS); do
  for TAG in "$@"; do
    LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$TAG.so
    [ "$TAG" = "default" ] && LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip.so
    PTR_HIP_LIB=$LIB timeout -k 10 200 python3 -u tools/gpu_serial_costs.py --spp ${SPP:-32} ${ARGS:-} > $OUT/serial_${TAG}_$R.json 2> $OUT/serial_${TAG}_$R.err || { echo "$TAG failed"; exit 1; }
    python3 - $OUT/serial_${TAG}_$R.json $TAG <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_pass"]
print("%-8s pass %.1f ms | closest %.1f shadow %.1f mis_any %.1f shade %.2f resolve %.1f miss %.1f" % (sys.argv[2], d["ms_per_pass"], k["trace_closest"][0], k["nee_shadow"][0], k.get("nee_mis_any",[0])[0], k["shade"][0], k["nee_resolve"][0], k["miss"][0]))
PY
  done
done
