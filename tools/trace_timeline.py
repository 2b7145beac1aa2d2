"""Print the kernels of a rocprofv3 kernel trace in start order, with duration and the gap to the
previous kernel's end, for the last `--last` milliseconds of the trace (or everything).
usage: python tools/trace_timeline.py <dir> [--last-ms 120]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else None
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].split("(")[0].replace("void ", ""),
                     row.get("Stream_Id", row.get("Queue_Id", "?"))))
rows.sort()
if not rows:
    sys.exit("no kernel_trace.csv under %s" % d)
t_end = max(r[1] for r in rows)
t0 = rows[0][0]
prev_end = None
for s, e, name, q in rows:
    if last_ms is not None and s < t_end - last_ms * 1e6:
        prev_end = e
        continue
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%10.3f ms  +%9.1f us  dur %9.1f us  q%-4s %s" % ((s - t0) / 1e6, gap, (e - s) / 1e3, q, name))
    prev_end = max(prev_end or e, e)
