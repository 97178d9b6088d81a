#!/bin/bash
# GPU box, round 6: the strict C4 step with the co-run down-converters capped per CU (CSDR_DC_CAP_CORUN="second,later",
# enforced with unused LDS) so that the dispatcher spreads them evenly, for several grid sizes (CSDR_DC_WGS_CORUN)
export TMPDIR=/tmp
out=gpurun_out/r6_dc_cap.txt
mkdir -p gpurun_out
: > $out
run() {   # label, env...
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
run baseline            CSDR_NOP=1
run cap13_12            CSDR_DC_CAP_CORUN=13,12
run cap13_13            CSDR_DC_CAP_CORUN=13,13
run cap13_14            CSDR_DC_CAP_CORUN=13,14
run cap14_14            CSDR_DC_CAP_CORUN=14,14
run cap14_14_g14        CSDR_DC_CAP_CORUN=14,14 CSDR_DC_WGS_CORUN=3584,3584
run cap14_14_g14_13     CSDR_DC_CAP_CORUN=14,14 CSDR_DC_WGS_CORUN=3584,3328
run cap15_15_g15        CSDR_DC_CAP_CORUN=15,15 CSDR_DC_WGS_CORUN=3840,3840
run cap16_14_g16_13     CSDR_DC_CAP_CORUN=0,14 CSDR_DC_WGS_CORUN=4096,3328
run baseline2           CSDR_NOP=1
