#!/bin/bash
# GPU box, round 6: stream priorities of the plan groups (CSDR_CHAIN_PRIO, by rank: 0 = highest) -- the strict step on a
# fresh object and on the object made right after bench.py's control_plane (the slow state of tools/experiments/r6_repro_mode3.py)
export TMPDIR=/tmp
out=gpurun_out/r6_chain_prio.txt
mkdir -p gpurun_out
: > $out
for p in 0,1,2 2,1,0 1,2,0 1,1,0 0,0,0 2,2,0 0,2,1; do
    a=$(CSDR_CHAIN_PRIO=$p timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
    b=$(CSDR_CHAIN_PRIO=$p timeout -k 10 300 python3 tools/experiments/r6_repro_mode3.py 2>&1 | grep '"ms"' | tail -1)
    c=$(CSDR_CHAIN_PRIO=$p timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
    echo "prio=$p fresh $a after_control_plane $b fresh_again $c" | tee -a $out
done
