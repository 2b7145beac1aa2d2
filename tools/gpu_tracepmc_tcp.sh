#!/bin/bash
# texture-path counters (TA / TCP / TD) of the persistent tracer over tools/gpu_tracepmc.py: bash tools/gpu_tracepmc_tcp.sh <tag>
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcp_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in "t1:TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "t2:TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "t3:TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE" \
         "t4:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVES TA_BUSY_avr TA_BUSY_max"; do
  NAME=${P%%:*}; CTRS=${P#*:}
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/$NAME -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_tracepmc.py 2 > $OUT/$NAME.log 2>&1 || { echo "pmc pass $NAME failed"; tail -5 $OUT/$NAME.log; exit 1; }
done
cd $GRAFT_REPO_ROOT && python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_ptrace" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print("==", k)
    for c in sorted(acc[k]): print("  %-44s %16.0f per launch (%d launches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
find $OUT -name "*.csv" -size +5M -delete
