#!/bin/bash
# GPU box, round 6: A/B of the strict C4 step under the experiment switches of capi_demod.hip (one process per
# variant: the switches are read once).  Output: gpurun_out/r6_c4_ab.txt
export TMPDIR=/tmp
out=gpurun_out/r6_c4_ab.txt
mkdir -p gpurun_out
: > $out
run() {   # label, env...
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
run baseline                    CSDR_NOP=1
run hwq8                        GPU_MAX_HW_QUEUES=8
run corun3400                   CSDR_DC_WGS_CORUN=3400
run corun3072                   CSDR_DC_WGS_CORUN=3072
run corun3584                   CSDR_DC_WGS_CORUN=3584
run cumask48                    CSDR_CHAIN_CUMASK=48
run cumask48_hwq8               CSDR_CHAIN_CUMASK=48 GPU_MAX_HW_QUEUES=8
run cumask48_hwq8_corun3328     CSDR_CHAIN_CUMASK=48 GPU_MAX_HW_QUEUES=8 CSDR_DC_WGS_CORUN=3328
run cumask88_hwq8_corun2688     CSDR_CHAIN_CUMASK=88 GPU_MAX_HW_QUEUES=8 CSDR_DC_WGS_CORUN=2688
run unchained                   CSDR_CHAIN_DC_CHAINED=0
run unchained_hwq8              CSDR_CHAIN_DC_CHAINED=0 GPU_MAX_HW_QUEUES=8
run smside_hwq8                 CSDR_SM_SIDE=1 GPU_MAX_HW_QUEUES=8
