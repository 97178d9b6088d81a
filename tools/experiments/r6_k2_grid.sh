#!/bin/bash
# GPU box, round 6: the down-converter ALONE (86 receivers x 2^21, tools/bench_k2_plans.py) by grid size (one-wave
# workgroups per launch) and by register budget (4 waves per SIMD at 128 VGPRs with 68-84 B of scratch, against 3 waves
# at 151-154 VGPRs, no scratch: cutesdr_amd/_var/dc3)
export TMPDIR=/tmp
out=gpurun_out/r6_k2_grid.txt
: > $out
for lib in cutesdr_amd/libcutesdr_mi.so cutesdr_amd/_var/dc3/libcutesdr_mi_dc3.so; do
  for w in 4096 3584 3072 2560 2048; do
    r=$(CSDR_LIB_PATH=$lib CSDR_DC_WGS=$w timeout -k 10 300 python3 tools/bench_k2_plans.py 86 2>&1 | grep '^{' | tail -1)
    echo "$(basename $lib) wgs=$w $r" | tee -a $out
  done
done
