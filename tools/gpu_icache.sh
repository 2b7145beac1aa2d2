#!/bin/bash
# instruction-cache counters per kernel over a short bench run: bash tools/gpu_icache.sh <tag> [bench args]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/ic -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 "$@" > $OUT/ic.log 2>&1 || { echo "icache pass failed"; tail -5 $OUT/ic.log; exit 1; }
cd $GRAFT_REPO_ROOT && python3 tools/prof_summary.py $OUT/ic > $OUT/summary.txt
find $OUT -name "*.csv" -size +5M -delete
grep -E "k_shade|k_ptrace<false, false>|k_nee_resolve|k_miss" $OUT/summary.txt
