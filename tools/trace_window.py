#!/usr/bin/env python3
"""A time window of a rocprofv3 --kernel-trace CSV: kernels whose start lies within [end - back_us, end - back_us + span_us).
   usage: tools/trace_window.py <dir> <back_us> <span_us>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if True:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
tend = max(r[1] for r in rows)
t0 = tend - int(float(sys.argv[2]) * 1e3); t1 = t0 + int(float(sys.argv[3]) * 1e3)
qs = {}
for s, e, name, q, wgs in rows:
    qs.setdefault(q, len(qs))
    if t0 <= s < t1:
        short = name.replace("void csdr::", "").replace("csdr::", "").split("(")[0][:48]
        print("%8.1f %8.1f  (%6.1f) q%-2d %-48s wgs=%d" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qs[q], short, wgs))
