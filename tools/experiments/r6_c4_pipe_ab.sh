#!/bin/bash
# GPU box, round 6: the pipelined C4 step with the groups' down-converters chained and sized for a chip that holds the
# previous call's walks
export TMPDIR=/tmp
out=gpurun_out/r6_c4_pipe_ab.txt
: > $out
run() {
    label=$1; shift
    for rep in 1 2; do
        r=$(env "$@" timeout -k 10 300 python3 tools/bench_c4_pipe.py 2>&1 | grep '^{' | tail -1)
        echo "$label rep$rep $r" | tee -a $out
    done
}
run baseline CSDR_NOP=1
run chained CSDR_PIPE_DC_CHAINED=1
for w in 3584 3328 3072 2816; do
  run chained_wgs$w CSDR_PIPE_DC_CHAINED=1 CSDR_PIPE_DC_WGS=$w
  run unchained_wgs$w CSDR_PIPE_DC_WGS=$w
done
run strict_default CSDR_NOP=1
