#!/bin/bash
# GPU box, round 4: whole-call S-meter (CSDR_SM_CALL), lean walk (CSDR_PC_LEAN), S-meter on a side stream (CSDR_SM_SIDE)
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for cfg in "0 0 0" "1 0 0" "1 1 0" "1 1 1" "0 0 0" "1 1 1"; do
  set -- $cfg
  echo "sm_call=$1 lean=$2 side=$3 strict $(CSDR_SM_CALL=$1 CSDR_PC_LEAN=$2 CSDR_SM_SIDE=$3 python3 tools/bench_c4_strict.py 2>&1 | tail -1) pipe $(CSDR_SM_CALL=$1 CSDR_PC_LEAN=$2 CSDR_SM_SIDE=$3 python3 tools/bench_c4_pipe.py 2>&1 | tail -1)" | tee -a gpurun_out/r4_lean.log
done
rm -rf gpurun_out/prof_sm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sm -o t -- python3 tools/bench_c4_strict.py > gpurun_out/sm_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_sm 2 > gpurun_out/r4_lean_timeline.txt 2>&1
rm -rf gpurun_out/prof_sm
