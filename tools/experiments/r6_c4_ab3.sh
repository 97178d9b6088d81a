#!/bin/bash
# GPU box, round 6: the strict C4 step with down-converters of TWO (or three) rounds of shorter workgroups, whose slots
# open progressively for the kernels launched beside them
export TMPDIR=/tmp
out=gpurun_out/r6_c4_ab3.txt
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
run corun_6656_6144 CSDR_DC_WGS_CORUN=6656,6144
run corun_8192_8192 CSDR_DC_WGS_CORUN=8192,8192
run all_8192        CSDR_DC_WGS=8192 CSDR_DC_WGS_CORUN=8192,8192
run corun_9984_9216 CSDR_DC_WGS_CORUN=9984,9216
run corun_12288     CSDR_DC_WGS_CORUN=12288,12288
run baseline2       CSDR_NOP=1
