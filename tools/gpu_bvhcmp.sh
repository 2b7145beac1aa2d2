#!/bin/bash
# host-SAH against device-LBVH geometry trees on the full-size configs: build time, depth, nodes/items per ray, step time
set -e
out=gpurun_out/${1:-bvhcmp}
cfgs=${2:-"1 2 3 4"}
mkdir -p $out
for cfg in $cfgs; do
  for b in host device; do
    timeout -k 10 280 python3 bench.py --config $cfg --bvh $b --steps 4 --warmup 1 --no-cpu > $out/c${cfg}_$b.json 2> $out/c${cfg}_$b.err
    python3 - $out/c${cfg}_$b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c, r = d["config"], d["roofline"]
print("cfg %d %-12s build %9.1f ms depth %2d nodes %8d | boxes/ray %6.2f items/ray %6.2f | %7.1f ms/step value %7.1f tracer %.3f ms" % (
    c["config_index"], c["bvh_builder"], c["bvh_build_ms"], c["bvh_depth"], c["bvh_nodes"], r["nodes_per_ray"], r["items_per_ray"], d["ms_per_step"], d["value"], r["avg_launch_ms"]))
PY
  done
done
