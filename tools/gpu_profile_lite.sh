#!/bin/bash
# kernel traces only (no counter passes) of a short bench run of one config: bash tools/gpu_profile_lite.sh <tag> <config> [steps]
set -u
TAG=$1; CFG=$2; STEPS=${3:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu --config $CFG --steps $STEPS --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_exclusive -o t -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --roofline-only > $OUT/trace_exclusive.log 2>&1 || { echo "exclusive trace failed"; tail -5 $OUT/trace_exclusive.log; exit 1; }
grep -h "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench.json
find $OUT -name "*.csv" -size +20M -delete
head -6 $OUT/trace_exclusive/t_kernel_stats.csv | cut -c1-160
