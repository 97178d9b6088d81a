#!/bin/bash
# GPU box, round 6: 3.5 ms of the pipelined C4 chain's kernel trace (what overlaps what across calls)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_pipe; mkdir -p gpurun_out/prof_pipe
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pipe -o t -- python3 tools/bench_c4_pipe.py > gpurun_out/pipe_tl.log 2>&1
python3 - <<'PY' > gpurun_out/r6_pipe_timeline.txt
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_pipe/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
t_end = rows[-1][1]
# the window [t_end - 6 ms, t_end - 2.5 ms]
lo, hi = t_end - 6_000_000, t_end - 2_500_000
qs = {}
first = None
for s, e, name, q, wgs in rows:
    if s < lo or s > hi: continue
    if first is None: first = s
    qs.setdefault(q, len(qs))
    short = name.replace("void csdr::", "").replace("csdr::", "").split("(")[0][:44]
    print("%8.1f %8.1f  q%-2d %-44s wgs=%d" % ((s - first) / 1e3, (e - first) / 1e3, qs[q], short, wgs))
PY
rm -rf gpurun_out/prof_pipe
tail -1 gpurun_out/pipe_tl.log
