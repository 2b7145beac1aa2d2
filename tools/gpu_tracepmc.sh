#!/bin/bash
# counter passes over tools/gpu_tracepmc.py for one tracer: bash tools/gpu_tracepmc.sh <tag>   (PTR_TRACER etc. from the environment)
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" \
         "b:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
         "c:SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_VMEM" \
         "d:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/$NAME -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_tracepmc.py 2 > $OUT/$NAME.log 2>&1 || { echo "pmc pass $NAME failed"; tail -5 $OUT/$NAME.log; exit 1; }
done
cd $GRAFT_REPO_ROOT && python3 tools/prof_summary.py $OUT/a $OUT/b $OUT/c $OUT/d > $OUT/summary.txt
find $OUT -name "*.csv" -size +5M -delete
grep -E "trace|==" $OUT/summary.txt
