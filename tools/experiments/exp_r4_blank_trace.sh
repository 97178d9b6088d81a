#!/bin/bash
# GPU box: kernel timeline of the datagram-fed chain with the fused blanker (last step of tools/bench_chain.py)
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
rm -rf gpurun_out/prof_bl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bl -o t -- python3 tools/bench_chain.py > gpurun_out/bl_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_bl 1 noiseblank > gpurun_out/r4_blank_timeline.txt 2>&1
python3 - <<'PY' >> gpurun_out/r4_blank_timeline.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_bl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-92s n=%4d mean %.1f us min %.1f max %.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
PY
rm -rf gpurun_out/prof_bl
