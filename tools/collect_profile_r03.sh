#!/bin/bash
# Round-3 evidence in one gpurun call: K1 (kernel-trace stats + FETCH/WRITE/SQ passes -> profiles/r03_*), the in-kernel
# clock of K1 (tools/k1_cycles.py on the -DK1_CYC build -> profiles/r03_k1_cycles.json), K2 per decimator plan as the
# chain launches it (kernel-trace stats of tools/bench_k2_plans.py -> profiles/r03_k2_plans_*), then the chain kernels
# (tools/collect_chain_profile.sh -> profiles/r03_chain_*).   usage: tools/collect_profile_r03.sh
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
bash tools/collect_profile.sh r03 && echo "K1 profile done"
CSDR_LIB_PATH=$PWD/cutesdr_amd/libcutesdr_mi_cyc.so python3 tools/k1_cycles.py > profiles/r03_k1_cycles.json 2> gpurun_out/k1_cycles.err; cat profiles/r03_k1_cycles.json
OUT=gpurun_out/prof_k2plans_r03
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/bench_k2_plans.py 86 > profiles/r03_k2_plans.json 2> gpurun_out/k2plans.err
cat profiles/r03_k2_plans.json
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_k2plans_r03/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "downconv" in r["Name"]:
            rows.append([r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
with open("profiles/r03_k2_plans_kernel_stats.csv", "w", newline="") as fo:
    w = csv.writer(fo); w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]); w.writerows(rows)
PY
bash tools/collect_chain_profile.sh r03 && echo "chain profile done"
cp profiles/r03_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
