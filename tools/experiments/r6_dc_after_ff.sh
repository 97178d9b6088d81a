#!/bin/bash
# GPU box, round 6: the next group's down-converter behind this group's FILTER (CSDR_CHAIN_DC_AFTER_FF=1) instead of
# behind its down-converter, for a few co-run grids; then one traced step
export TMPDIR=/tmp
out=gpurun_out/r6_dc_after_ff.txt
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
run after_ff        CSDR_CHAIN_DC_AFTER_FF=1
run after_ff_14_13  CSDR_CHAIN_DC_AFTER_FF=1 CSDR_DC_WGS_CORUN=3584,3328
run after_ff_15_14  CSDR_CHAIN_DC_AFTER_FF=1 CSDR_DC_WGS_CORUN=3840,3584
run after_ff_16_16  CSDR_CHAIN_DC_AFTER_FF=1 CSDR_DC_WGS_CORUN=4096,4096
run after_ff_16_13  CSDR_CHAIN_DC_AFTER_FF=1 CSDR_DC_WGS_CORUN=4096,3328
run baseline2       CSDR_NOP=1
CSDR_LIB_PATH=cutesdr_amd/_var/trace/libcutesdr_mi_trace.so CSDR_CHAIN_DC_AFTER_FF=1 timeout -k 10 300 python3 tools/wg_trace.py run strict gpurun_out/wgtrace_after_ff.json > gpurun_out/wgtrace_after_ff.txt 2>&1
