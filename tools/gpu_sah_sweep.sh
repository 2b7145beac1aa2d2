#!/bin/bash
# SAH leaf-rule sweep on one serial pass: bash tools/gpu_sah_sweep.sh <outdir> "ct:always" ...   (ARGS= extra gpu_serial_costs.py flags)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
for S in "$@"; do
  CT=${S%%:*}; AL=${S##*:}
  PTR_SAH_CT=$CT PTR_SAH_ALWAYS_LEAF=$AL timeout -k 10 200 python3 -u tools/gpu_serial_costs.py --spp ${SPP:-32} ${ARGS:-} > $OUT/serial_${CT}_$AL.json 2> $OUT/serial_${CT}_$AL.err || { echo "$S failed"; exit 1; }
  python3 - $OUT/serial_${CT}_$AL.json $S <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["kernels_ms_per_pass"]
print("%-8s pass %.1f ms | closest %.1f shadow %.1f mis_any %.1f shade %.2f resolve %.1f miss %.1f" % (sys.argv[2], d["ms_per_pass"], k["trace_closest"][0], k["nee_shadow"][0], k.get("nee_mis_any",[0])[0], k["shade"][0], k["nee_resolve"][0], k["miss"][0]))
PY
done
