#!/bin/bash
# GPU box: kernel timeline of one strict-mode C4 step under the environment given (e.g. CSDR_CHAIN_LAST_SPLIT=2)
#   usage: tools/experiments/trace_c4_strict.sh NAME   -> gpurun_out/timeline_NAME.txt
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
rm -rf gpurun_out/prof_t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -o t -- python3 tools/bench_c4_strict.py > gpurun_out/t_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_t 2 > gpurun_out/timeline_$1.txt 2>&1
rm -rf gpurun_out/prof_t
tail -1 gpurun_out/t_trace.log
