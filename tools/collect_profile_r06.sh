#!/bin/bash
# Round-6 evidence in one gpurun call: K1 (kernel-trace stats + FETCH / WRITE / SQ passes -> profiles/r06_summary.json,
# r06_kernel_stats.csv, traffic_latest.json), the chain kernels on the C4 share (tools/collect_chain_profile.sh ->
# profiles/r06_chain_*), the strict chain's timeline (profiles/r06_chain_timeline.txt), the 4096- and 8192-point filter
# launches (profiles/r06_fastfir_sizes_kernel_stats.csv) and the builder's own full bench.py line
# (profiles/r06_bench_builder_run.json), and one strict step workgroup by workgroup with and without the co-run grid
# (profiles/r06_wgtrace_strict*.json).   usage: tools/collect_profile_r06.sh
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out profiles
bash tools/collect_profile.sh r06 && echo "K1 profile done"
bash tools/collect_chain_profile.sh r06 && echo "chain profile done"
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o t -- python3 tools/bench_c4_strict.py > gpurun_out/tl.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_tl 2 > profiles/r06_chain_timeline.txt 2>&1
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
    with open("profiles/r06_fastfir_sizes_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
rm -rf gpurun_out/prof_ffs
python3 bench.py --steps 100 --warmup 20 > profiles/r06_bench_builder_run.json 2> gpurun_out/bench_r06.err
# one strict step drawn workgroup by workgroup (tools/wg_trace.py; the traced copy of the library is built by `python tools/wg_trace.py build`)
if [ -f cutesdr_amd/_var/trace/libcutesdr_mi_trace.so ]; then
  CSDR_LIB_PATH=cutesdr_amd/_var/trace/libcutesdr_mi_trace.so python3 tools/wg_trace.py run strict profiles/r06_wgtrace_strict.json > gpurun_out/wgtrace_r06.log 2>&1
  CSDR_DC_WGS_CORUN=0 CSDR_LIB_PATH=cutesdr_amd/_var/trace/libcutesdr_mi_trace.so python3 tools/wg_trace.py run strict profiles/r06_wgtrace_strict_full_rounds.json > gpurun_out/wgtrace_r06b.log 2>&1
fi
cp profiles/r06_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
tail -c 600 profiles/r06_bench_builder_run.json
