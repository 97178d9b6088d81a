#!/bin/bash
# A/B of several alternative builds of the library (tools/altlib.py NAME ...) in one gpurun call: for every NAME the
# generic kernel (variant 0) and the build's K1 (variant 2) are timed interleaved in ONE process (tools/ab_fastfir.py),
# so the ratio ms_2 / ms_0 compares builds across processes.   usage: tools/ab_many.sh base NAME1 NAME2 ...
mkdir -p gpurun_out
for n in "$@"; do
    lib=cutesdr_amd/libcutesdr_mi_$n.so
    [ "$n" = base ] && lib=cutesdr_amd/libcutesdr_mi.so
    CSDR_LIB_PATH=$PWD/$lib timeout -k 10 240 python tools/ab_fastfir.py 0,2 8 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || echo "FAILED $n"
    echo "$n $(cat gpurun_out/ab_$n.json)"
done
