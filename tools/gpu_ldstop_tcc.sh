#!/bin/bash
# L2 requests of the tracer with and without the upper BVH levels staged in LDS (one TCC pass per variant library)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/ldstop
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for TAG in base lds48; do
  export PTR_HIP_LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$TAG.so
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc_$TAG -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_tracepmc.py 2 > $OUT/tcc_$TAG.log 2>&1 || { echo "pass $TAG failed"; tail -5 $OUT/tcc_$TAG.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq_$TAG -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_tracepmc.py 2 > $OUT/sq_$TAG.log 2>&1 || { echo "pass sq $TAG failed"; tail -5 $OUT/sq_$TAG.log; exit 1; }
  (cd $GRAFT_REPO_ROOT && python3 tools/prof_summary.py $OUT/tcc_$TAG $OUT/sq_$TAG > $OUT/summary_$TAG.txt)
  echo "== $TAG"; grep -E "k_ptrace" $OUT/summary_$TAG.txt
done
find $OUT -name "*.csv" -size +5M -delete
