#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for one kernel (mean per launch)."""
import csv, glob, sys, collections
pat = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "fastfir"
for f in sorted(glob.glob(pat, recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-28s n=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
