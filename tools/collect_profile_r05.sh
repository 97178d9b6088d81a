#!/bin/bash
# Round-5 evidence in one gpurun call: K1 (kernel-trace stats + FETCH / WRITE / SQ passes -> profiles/r05_summary.json,
# r05_kernel_stats.csv, traffic_latest.json), the chain kernels on the C4 share (tools/collect_chain_profile.sh ->
# profiles/r05_chain_*), the strict chain's timeline (profiles/r05_chain_timeline.txt), the 4096- and 8192-point filter
# launches (profiles/r05_fastfir_sizes_kernel_stats.csv) and the builder's own full bench.py line
# (profiles/r05_bench_builder_run.json).   usage: tools/collect_profile_r05.sh
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out profiles
bash tools/collect_profile.sh r05 && echo "K1 profile done"
bash tools/collect_chain_profile.sh r05 && echo "chain profile done"
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o t -- python3 tools/bench_c4_strict.py > gpurun_out/tl.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_tl 2 > profiles/r05_chain_timeline.txt 2>&1
rm -rf gpurun_out/prof_tl gpurun_out/prof_ffs
for n in 4096 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ffs/$n -- python3 tools/bench_ff_size.py $n > gpurun_out/ffs_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob
rows = []
for f in sorted(glob.glob("gpurun_out/prof_ffs/*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fastfir" in r["Name"]:
            rows.append(r)
if rows:
    with open("profiles/r05_fastfir_sizes_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
rm -rf gpurun_out/prof_ffs
python3 bench.py --steps 100 --warmup 20 > profiles/r05_bench_builder_run.json 2> gpurun_out/bench_r05.err
cp profiles/r05_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
tail -c 600 profiles/r05_bench_builder_run.json
