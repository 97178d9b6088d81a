#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM traffic and SQ counters for bench.py's headline
# workload, summarised into profiles/<tag>_* (tracked).  Usage: tools/collect_profile.sh r02
# (counters in their own passes, never together with a trace domain)
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
B="python3 bench.py --no-cpu --no-check --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 5 --warmup 1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/write -- $B --steps 5 --warmup 1 > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $B --steps 5 --warmup 1 > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/sq2 -- $B --steps 5 --warmup 1 > $OUT/bench_sq2.log 2>&1
python3 tools/summarise_profile.py $OUT $TAG
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_summary.json profiles/traffic_latest.json gpurun_out/ 2>/dev/null   # gpurun merges only gpurun_out/ back
