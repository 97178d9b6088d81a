#!/bin/bash
# GPU box, round 6: CSDR_CHAIN_SPLIT_LAST A/B on the strict C4 step (alternating), then what the split changes in the audio
export TMPDIR=/tmp
out=gpurun_out/r6_split_last.txt
mkdir -p gpurun_out
: > $out
for rep in 1 2 3 4; do
    for v in 0 1; do
        r=$(CSDR_CHAIN_SPLIT_LAST=$v timeout -k 10 300 python3 tools/bench_c4_strict.py 2>&1 | grep '^{' | tail -1)
        echo "split_last=$v rep$rep $r" | tee -a $out
    done
done
CSDR_CHAIN_SPLIT_LAST=0 timeout -k 10 300 python3 tools/experiments/r6_split_last.py dump /tmp/split0.npy &&
CSDR_CHAIN_SPLIT_LAST=1 timeout -k 10 300 python3 tools/experiments/r6_split_last.py dump /tmp/split1.npy &&
python3 tools/experiments/r6_split_last.py compare /tmp/split0.npy /tmp/split1.npy | tee -a $out
