#!/bin/bash
# Round-4 evidence in one gpurun call: K1 (kernel-trace stats + FETCH / WRITE / SQ passes -> profiles/r04_summary.json,
# r04_kernel_stats.csv, traffic_latest.json), the chain kernels on the C4 share (tools/collect_chain_profile.sh ->
# profiles/r04_chain_*), the strict chain's timeline (profiles/r04_chain_timeline.txt) and the builder's own full
# bench.py line (profiles/r04_bench_builder_run.json).   usage: tools/collect_profile_r04.sh
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out profiles
bash tools/collect_profile.sh r04 && echo "K1 profile done"
bash tools/collect_chain_profile.sh r04 && echo "chain profile done"
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o t -- python3 tools/bench_c4_strict.py > gpurun_out/tl.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_tl 2 > profiles/r04_chain_timeline.txt 2>&1
rm -rf gpurun_out/prof_tl
python3 bench.py --steps 100 --warmup 20 > profiles/r04_bench_builder_run.json 2> gpurun_out/bench_r04.err
cp profiles/r04_* profiles/traffic_latest.json gpurun_out/ 2>/dev/null
tail -c 600 profiles/r04_bench_builder_run.json
