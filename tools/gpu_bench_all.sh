#!/bin/bash
# one short bench line per BASELINE config (the default config 1 is benched by `python bench.py` itself): bash tools/gpu_bench_all.sh <tag>
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_all_$1
mkdir -p $OUT
for c in 0 2 3 4 5; do
  timeout -k 10 280 python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu > $OUT/config$c.json 2> $OUT/config$c.err || { echo "config $c failed"; tail -3 $OUT/config$c.err; exit 1; }
  python3 - $OUT/config$c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("config %d: value %.1f Mrays/s (all %.1f) %.1f ms/step | tracer %.3f ms/launch, %.2f boxes + %.2f items per ray, frac %.3f | %s" % (
    d["config"]["config_index"], d["value"], d["mrays_all"], d["ms_per_step"], r["avg_launch_ms"], r["nodes_per_ray"], r["items_per_ray"], r["frac"],
    {k: round(v, 1) for k, v in d["kernels_ms_exclusive_per_step"].items() if v >= 1}))
PY
done
