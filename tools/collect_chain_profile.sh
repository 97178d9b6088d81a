#!/bin/bash
# Run on the GPU box (via gpurun): the other kernels of the path on BASELINE config C4's per-GPU share
# (tools/bench_chain.py: K2 down-converter, K3 spectrum, K6 blanker / unpack, the whole chain) -- kernel-trace
# stats, then HBM traffic counters in their own passes -- condensed into profiles/<tag>_chain_*.
#   tools/collect_chain_profile.sh r02
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_chain_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/bench_chain.py > $OUT/chain.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- python3 tools/bench_chain.py > $OUT/chain_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- python3 tools/bench_chain.py > $OUT/chain_write.log 2>&1
python3 tools/summarise_chain_profile.py $OUT $TAG
