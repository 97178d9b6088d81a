#!/bin/bash
# GPU box, round 6: the lean walk with EIGHT waves (two samples per lane, CSDR_POSTCHAIN_WAVES=8) against four: C2, C5,
# the strict C4 step; then the post-chain and chain parity tests under it
export TMPDIR=/tmp
out=gpurun_out/r6_walk8.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
    for v in 4 ${WALK_WAVES:-8}; do
        a=$(CSDR_POSTCHAIN_WAVES=$v timeout -k 10 300 python3 tools/experiments/bench_c2c5.py 2>&1 | grep '^{' | tail -1)
        b=$(CSDR_POSTCHAIN_WAVES=$v timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "waves=$v rep$rep $a $b" | tee -a $out
    done
done
CSDR_POSTCHAIN_WAVES=${WALK_WAVES:-8} timeout -k 10 900 python -m pytest tests/test_postchain_gpu.py tests/test_chain_parity_gpu.py tests/test_randomized_gpu.py -m gpu -q 2>&1 | tail -4 | tee -a $out
