#!/bin/bash
# rocprofv3 passes over bench.py (kernel trace, then PMC passes each in its own run).
# usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu $*"   # bench.py defaults: the command the committed BENCH line comes from
CONFIG=1
PREV=""
for A in "$@"; do [ "$PREV" = "--config" ] && CONFIG=$A; PREV=$A; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
# the same program with the timed region skipped: only passes that run one at a time (bench.py's roofline leg)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_exclusive -o t -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS --roofline-only > $OUT/trace_exclusive.log 2>&1 || { echo "exclusive trace failed"; tail -5 $OUT/trace_exclusive.log; exit 1; }
for P in "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
         "sq2:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVES" \
         "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
         "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/$NAME -o p -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$NAME.log 2>&1 || { echo "pmc pass $NAME failed"; tail -5 $OUT/$NAME.log; exit 1; }
  echo "pass $NAME ok"
done
cd $GRAFT_REPO_ROOT && python3 tools/prof_summary.py --json $OUT/summary.json --config $CONFIG $OUT/trace_exclusive $OUT/trace $OUT/sq $OUT/sq2 $OUT/tcc $OUT/fetch $OUT/write > $OUT/summary.txt
# keep only the summaries small enough to merge back
find $OUT -name "*.csv" -size +20M -delete
tail -60 $OUT/summary.txt
