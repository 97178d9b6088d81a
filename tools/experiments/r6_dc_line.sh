#!/bin/bash
# GPU box, round 6: the strict C4 step with every group's down-converter in one stream (CSDR_CHAIN_DC_LINE: 1 the last
# group's stream, 2 / 3 / 4 a stream of its own at the highest / normal / lowest priority) against each in its group's
# stream behind the previous group's event (0, the default); alternating, 3 repetitions each
export TMPDIR=/tmp
out=gpurun_out/r6_dc_line.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2 3; do
    for v in 0 1 2 3 4; do
        r=$(CSDR_CHAIN_DC_LINE=$v timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "dc_line=$v rep$rep $r" | tee -a $out
    done
done
