#!/bin/bash
# GPU box, round 6: the chained pipeline (the default form of the pipelined mode) against the three-stage one and the strict step, and its
# co-run grids (second, later groups'; the first group's beside the previous call's walks)
export TMPDIR=/tmp
out=gpurun_out/r6_pipe_chained.txt
mkdir -p gpurun_out
: > $out
run() {   # label, env...
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/bench_c4_pipe.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
a=$(timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1); echo "strict $a" | tee -a $out
run three_stage          CSDR_PIPE_KIND=3
run chained              CSDR_PIPE_KIND=2
run chained_g3072        CSDR_DC_WGS_CORUN=3072,3072 CSDR_PIPE_FIRST_WGS=3072
run chained_g2816        CSDR_DC_WGS_CORUN=2816,2816 CSDR_PIPE_FIRST_WGS=2816
run chained_g2560        CSDR_DC_WGS_CORUN=2560,2560 CSDR_PIPE_FIRST_WGS=2560
run chained_g3328        CSDR_DC_WGS_CORUN=3328,3328 CSDR_PIPE_FIRST_WGS=3328
run chained_g4096        CSDR_DC_WGS_CORUN=4096,4096 CSDR_PIPE_FIRST_WGS=4096
