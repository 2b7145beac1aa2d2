#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of bench.py --roofline-only: usage tools/gpu_kstats.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --roofline-only "$@" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
cd $GRAFT_REPO_ROOT
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $F gpurun_out/ks_$TAG.csv
find $OUT -name "*.csv" -size +5M -delete
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ks_$TAG.csv")):
    print("  %-50s calls %5s total %9.1f ms avg %8.3f ms max %8.3f" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["MaxNs"])/1e6))
PY
