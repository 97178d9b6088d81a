#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM traffic counters for bench.py's
# default command, summarised into profiles/ (tracked).  Usage: tools/collect_profile.sh r01
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/write -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_sq.log 2>&1
python3 tools/summarise_profile.py $OUT $TAG
