"""Condense rocprofv3 CSV output (kernel trace / counter collection) into a per-kernel table.
usage: python tools/prof_summary.py [--json out.json] [--config N] <dir> [<dir> ...] > profiles/<name>.txt
--json also writes {kernel: {calls, avg_ns, counters: {name: sum}, dispatches}} for bench.py's roofline.traffic."""
import csv
import glob
import os
import json
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "")


args = sys.argv[1:]
json_out = None
config = None
while args and args[0] in ("--json", "--config"):
    if args[0] == "--json":
        json_out = args[1]
    else:
        config = int(args[1])
    args = args[2:]
summary = defaultdict(lambda: {"counters": {}})
for d in args:
    print("== %s" % d)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print("-- kernel stats (%s)" % os.path.basename(f))
        for row in csv.DictReader(open(f)):
            print("  %-34s calls %6s  total_ns %14s  avg_ns %12s  pct %6s" % (short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"),
                                                                               row.get("AverageNs"), row.get("Percentage")))
            summary[short(row.get("Name", ""))].update(calls=int(row.get("Calls") or 0), avg_ns=float(row.get("AverageNs") or 0))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print("-- counters (%s)" % os.path.basename(f))
        acc = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value") or 0)
            calls[k].add(row.get("Dispatch_Id"))
        for k in sorted(acc):
            print("  %-34s dispatches %5d  " % (k, len(calls[k])) + "  ".join("%s=%.6g" % (c, v) for c, v in sorted(acc[k].items())))
            summary[k]["counters"].update(acc[k])
            summary[k]["dispatches"] = len(calls[k])
if json_out:
    out = dict(summary)
    if config is not None:
        out["_config"] = config  # which bench.py --config the passes ran (bench.py only prices a kernel with traffic of its own config)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pathtracer_amd.capi import kernel_source_hash
    out["_csrc_hash"] = kernel_source_hash()  # the kernel sources the counters were collected from (bench.py: traffic is null when the tree differs)
    json.dump(out, open(json_out, "w"), indent=1, sort_keys=True)
