#!/bin/bash
# kernel + memcpy trace of the host form (drop-in CDemodulator, 256-sample calls): where a 19968-sample pass goes
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/hf_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hf_trace -- python3 $GRAFT_REPO_ROOT/tools/bench_host_form_only.py > $GRAFT_REPO_ROOT/gpurun_out/hf_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/hf_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void csdr::","").split("(")[0][:40]))
for f in glob.glob("gpurun_out/hf_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", r.get("Name",""))))
rows.sort()
n=len(rows)
i0=n*3//4
t0=rows[i0][0]
for s,e,name in rows[i0:i0+60]:
    print("%9.1f %9.1f %7.1f  %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,name))
PY
