#!/bin/bash
# film hashes against tools/ref/film_hashes_r03_start.txt, the GPU tests, one bench line: bash tools/gpu_check.sh <tag> [notests]
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
PYTHONPATH=.:tests timeout -k 10 300 python3 tools/gpu_film_hash.py > $OUT/${TAG}_hashes.txt 2> $OUT/${TAG}_hashes.err || { echo "hash run failed"; tail -5 $OUT/${TAG}_hashes.err; exit 1; }
if diff -q $OUT/${TAG}_hashes.txt tools/ref/film_hashes_r03_start.txt > /dev/null; then echo "film hashes: identical"; else echo "film hashes: DIFFER"; diff $OUT/${TAG}_hashes.txt tools/ref/film_hashes_r03_start.txt | head -10; fi
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/${TAG}_gputests.log 2>&1 || { echo "gpu tests failed"; tail -30 $OUT/${TAG}_gputests.log; exit 1; }
  tail -1 $OUT/${TAG}_gputests.log
fi
timeout -k 10 200 python3 bench.py --no-cpu > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { echo "bench failed"; tail -5 $OUT/${TAG}_bench.err; exit 1; }
python3 - $OUT/${TAG}_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("value %.1f Mrays/s (all %.1f) %.2f ms/step | tracer %.3f ms/launch frac %.3f | shade %.3f ms frac %.3f | %s" % (
    d["value"], d["mrays_all"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["roofline_shade"]["avg_launch_ms"], d["roofline_shade"]["frac"],
    {k: round(v, 1) for k, v in d["kernels_ms_exclusive_per_step"].items() if v >= 1}))
PY
