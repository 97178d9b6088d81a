#!/bin/bash
# GPU box, round 6: the lean walk compiled for 3 / 4 waves per SIMD (168 / 128 registers: 268 / 420 B of scratch per lane) so
# that it fits beside 12 down-converter waves per CU -- the strict C4 step with each (cutesdr_amd/_var/walk3, walk4)
export TMPDIR=/tmp
out=gpurun_out/r6_walk_regs.txt
: > $out
for rep in 1 2; do
  for lib in cutesdr_amd/libcutesdr_mi.so cutesdr_amd/_var/walk3/libcutesdr_mi_walk3.so cutesdr_amd/_var/walk4/libcutesdr_mi_walk4.so; do
    r=$(CSDR_LIB_PATH=$lib timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
    echo "$(basename $lib) rep$rep $r" | tee -a $out
  done
done
