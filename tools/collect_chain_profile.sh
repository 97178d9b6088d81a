#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of tools/bench_chain.py (down-converter alone,
# then the whole mixed chain on 256 channels), condensed into profiles/<tag>_chain_*.  Usage:
#   tools/collect_chain_profile.sh r01
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_chain_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/bench_chain.py > $OUT/chain.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, os
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        rows.append([n if len(n) < 100 else n[:97] + "...", r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
with open("profiles/%s_chain_kernel_stats.csv" % tag, "w", newline="") as fo:
    w = csv.writer(fo); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]); w.writerows(rows)
line = None
for l in open(os.path.join(out, "chain.log")):
    if l.startswith("{"): line = json.loads(l)
json.dump({"tag": tag, "command": "rocprofv3 --kernel-trace --stats -- python3 tools/bench_chain.py", "bench_chain_line_under_profiler": line},
          open("profiles/%s_chain_summary.json" % tag, "w"), indent=1)
print(line)
PY
