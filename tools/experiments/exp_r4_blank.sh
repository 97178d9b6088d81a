#!/bin/bash
# GPU box, round 4: the blanker fused into the down-converter's load (mask mode) against the two-pass form
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for m in 0 1 0 1; do
  echo "fused=$m $(CSDR_BLANK_FUSED=$m python3 tools/bench_chain.py 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if 'packets' in k or 'k6' in k})")" | tee -a gpurun_out/r4_blank.log
done
