#!/bin/bash
# GPU box: in-kernel cycle shares of the lean walk per mode (a -DPC_PROFILE build: tools/altlib.py pcprof postchain_kernels.hip -DPC_PROFILE)
export TMPDIR=/tmp
for m in FM USB AM; do
  CSDR_LIB_PATH=$PWD/cutesdr_amd/libcutesdr_mi_pcprof.so python3 tools/experiments/bench_walk_inputs.py $m carrier 2>&1 | grep -E "pcprof|ms_per_call" | tail -2
done
