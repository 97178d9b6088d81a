#!/bin/bash
# GPU box, round 4: strict chain with the groups' down-converters chained (default) or all started at once
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for m in 1 0 1 0; do
  echo "dc_chained=$m strict $(CSDR_CHAIN_DC_CHAINED=$m python3 tools/bench_c4_strict.py 2>&1 | tail -1)" | tee -a gpurun_out/r4_unchained.log
done
for m in 1 0; do
rm -rf gpurun_out/prof_sm
CSDR_CHAIN_DC_CHAINED=$m rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sm -o t -- python3 tools/bench_c4_strict.py > gpurun_out/sm_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_sm 2 > gpurun_out/r4_unchained_timeline$m.txt 2>&1
rm -rf gpurun_out/prof_sm
done
