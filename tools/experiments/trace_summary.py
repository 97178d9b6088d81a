#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV: per (kernel, grid) the launch count and the median / minimum duration.
   usage: tools/trace_summary.py <dir with *_kernel_trace.csv> [name filter]"""
import csv, glob, sys, statistics
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        key = (name[:90], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, wgs), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-90s wgs=%-6d n=%-4d med=%8.1f us min=%8.1f us" % (name, wgs, len(d), statistics.median(d) / 1e3, min(d) / 1e3))
