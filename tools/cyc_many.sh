#!/bin/bash
# tools/k1_cycles.py over several diagnostic builds (tools/altlib.py NAME -DK1_CYC ...) in one gpurun call
mkdir -p gpurun_out
for n in "$@"; do
    echo "$n $(CSDR_LIB_PATH=$PWD/cutesdr_amd/libcutesdr_mi_$n.so timeout -k 10 240 python tools/k1_cycles.py 2> gpurun_out/cyc_$n.err | tee gpurun_out/cyc_$n.json)"
done
