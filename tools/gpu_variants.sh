#!/bin/bash
# A/B of kernel-variant libraries on the tracer microbench and on one serial pass: bash tools/gpu_variants.sh <outdir> tag[:ENV=VAL,...] ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
for V in "$@"; do
  TAG=${V%%:*}; ENVS=""
  if [[ "$V" == *:* ]]; then ENVS=$(echo "${V#*:}" | tr ',' ' '); fi
  LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$TAG.so
  [ "$TAG" = "default" ] && LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip.so
  env PTR_HIP_LIB=$LIB $ENVS timeout -k 10 200 python3 -u tools/gpu_tracepmc.py 3 > $OUT/tp_$TAG.log 2>&1 || { echo "$TAG tracepmc failed"; tail -3 $OUT/tp_$TAG.log; continue; }
  env PTR_HIP_LIB=$LIB $ENVS timeout -k 10 200 python3 -u tools/gpu_serial_costs.py --spp ${SPP:-32} > $OUT/serial_$TAG.json 2> $OUT/serial_$TAG.err || { echo "$TAG serial failed"; continue; }
  python3 - <<PY
import json,re
t=open("$OUT/tp_$TAG.log").read()
ms=re.findall(r"(closest|any) n \d+\s+([\d.]+) ms",t)
d=json.load(open("$OUT/serial_$TAG.json"))
k=d["kernels_ms_per_pass"]
print("%-10s micro %s | pass %.1f ms %.0f Mrays/s | closest %.1f shadow %.1f mis_any %.1f shade %.1f" % ("$TAG", " ".join("%s %s"%m for m in ms), d["ms_per_pass"], d["mrays_s"], k["trace_closest"][0], k["nee_shadow"][0], k.get("nee_mis_any",[0])[0], k["shade"][0]))
PY
done
