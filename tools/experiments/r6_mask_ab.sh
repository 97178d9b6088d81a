#!/bin/bash
# GPU box, round 6: the integer mask kernel, several libraries alternating (CSDR_LIB_PATH; "-" = the product), average
# launch time inside the datagram-fed C4 chain with the blanker (rocprofv3 kernel stats)
#   tools/experiments/r6_mask_ab.sh cutesdr_amd/_var/pf2/libcutesdr_mi_pf2.so - cutesdr_amd/_var/pf4/libcutesdr_mi_pf4.so
export TMPDIR=/tmp
out=gpurun_out/r6_mask_ab.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2; do
    for lib in "$@"; do
        [ "$lib" = "-" ] && lib=""
        echo "lib=${lib:-product} rep$rep" | tee -a $out
        CSDR_LIB_PATH=$lib bash tools/experiments/k6_mask_time.sh 2>&1 | grep -E "^\{|mask" | tee -a $out
    done
done
