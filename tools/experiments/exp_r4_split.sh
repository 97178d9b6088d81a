#!/bin/bash
# GPU box, round 4: strict-mode C4 chain, walks split per recurrence (CSDR_CHAIN_SPLIT) x phased / interleaved groups
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for cfg in "0 0" "0 1" "0 2" "1 2" "0 0" "0 2"; do
  set -- $cfg
  echo "phased=$1 split=$2 $(CSDR_CHAIN_PHASED=$1 CSDR_CHAIN_SPLIT=$2 python3 tools/bench_c4_strict.py 2>&1 | tail -1)" | tee -a gpurun_out/r4_split.log
done
for cfg in "0 2" "1 2"; do
  set -- $cfg
  rm -rf gpurun_out/prof_sp
  CSDR_CHAIN_PHASED=$1 CSDR_CHAIN_SPLIT=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sp -o t -- python3 tools/bench_c4_strict.py > gpurun_out/sp_trace.log 2>&1
  python3 tools/trace_timeline.py gpurun_out/prof_sp 2 > gpurun_out/r4_split_timeline_$1_$2.txt 2>&1
  rm -rf gpurun_out/prof_sp
done
