#!/bin/bash
# GPU box, round 6: the strict schedule with the caller's stream joining a call only behind the next call's launches
# (CSDR_CHAIN_LAZY_JOIN=1) -- back-to-back steps, alternating
export TMPDIR=/tmp
out=gpurun_out/r6_lazy_join.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2 3; do
    for v in 0 1; do
        r=$(CSDR_CHAIN_LAZY_JOIN=$v timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "lazy_join=$v rep$rep $r" | tee -a $out
    done
done
