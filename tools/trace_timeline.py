#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace CSV as a timeline: kernels of the last complete step (steps are told apart
by the first downconv launch of each), start / end in us from the step's first launch, with the queue each ran on.
   usage: tools/trace_timeline.py <dir> [steps back from the end, default 2] [anchor kernel substring]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Kernel_Name"] or os.environ.get("TIMELINE_ALL"):    # TIMELINE_ALL=1: the runtime's and torch's kernels too
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
anchor = sys.argv[3] if len(sys.argv) > 3 else "downconv"
# a step starts at an anchor launch that follows a non-anchor launch by a gap
starts = [i for i, r in enumerate(rows) if anchor in r[2] and (i == 0 or rows[i][0] - max(x[1] for x in rows[max(0, i - 12):i]) > 0 and all(anchor not in x[2] for x in rows[max(0, i - 3):i]))]
per = {}
step_starts = []
last = -10**18
for i in starts:
    if rows[i][0] - last > 1_000_000:      # steps are > 1 ms apart
        step_starts.append(i); last = rows[i][0]
i0 = step_starts[-1 - back]; i1 = step_starts[-back] if back > 0 else len(rows)
t0 = rows[i0][0]
qs = {}
for s, e, name, q, wgs in rows[i0:i1]:
    qs.setdefault(q, len(qs))
    short = name.replace("void csdr::", "").replace("csdr::", "").split("(")[0][:48]
    print("%8.1f %8.1f  q%-2d %-48s wgs=%d" % ((s - t0) / 1e3, (e - t0) / 1e3, qs[q], short, wgs))
print("step span %.1f us" % ((max(r[1] for r in rows[i0:i1]) - t0) / 1e3))
