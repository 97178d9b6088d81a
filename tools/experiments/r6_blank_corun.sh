#!/bin/bash
# GPU box, round 6: the datagram chain with the blanker for several co-run grids of its (blanked) down-converters
export TMPDIR=/tmp
out=gpurun_out/r6_blank_corun.txt
mkdir -p gpurun_out
: > $out
run() {   # label, env...
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/experiments/bench_blank_widths.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
run default        CSDR_NOP=1
run g3584_3328     CSDR_DC_WGS_CORUN=3584,3328
run g3072_2816     CSDR_DC_WGS_CORUN=3072,2816
run g3328_3328     CSDR_DC_WGS_CORUN=3328,3328
run g3072_3072     CSDR_DC_WGS_CORUN=3072,3072
run g4096_4096     CSDR_DC_WGS_CORUN=4096,4096
run default2       CSDR_NOP=1
