#!/bin/bash
# GPU box, round 6: finer sweep of the co-run grid of the strict C4 step (second group's, later groups' workgroups)
export TMPDIR=/tmp
out=gpurun_out/r6_c4_ab2.txt
mkdir -p gpurun_out
: > $out
run() {   # label, env...
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
run baseline        CSDR_NOP=1
for w in 4000,3264 3840,3072 3584,3072 3328,3072 3072,3072 3072,2816 2816,2816 3072,2560 2560,2560 4096,3072 3072,4096; do
    run corun_$w CSDR_DC_WGS_CORUN=$w
done
