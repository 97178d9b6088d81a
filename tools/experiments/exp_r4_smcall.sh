#!/bin/bash
# GPU box, round 4: the S-meter as one whole-call scan per receiver (CSDR_SM_CALL) against the S-meter inside the walk
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for m in 0 1 0 1; do
  echo "sm_call=$m strict $(CSDR_SM_CALL=$m python3 tools/bench_c4_strict.py 2>&1 | tail -1) pipe $(CSDR_SM_CALL=$m python3 tools/bench_c4_pipe.py 2>&1 | tail -1)" | tee -a gpurun_out/r4_smcall.log
done
rm -rf gpurun_out/prof_sm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sm -o t -- python3 tools/bench_c4_strict.py > gpurun_out/sm_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_sm 2 > gpurun_out/r4_smcall_timeline.txt 2>&1
rm -rf gpurun_out/prof_sm
