#!/bin/bash
# GPU box, round 6: the strict step's timeline with the caller on a stream of its own (OWN=1) and on the null stream
export TMPDIR=/tmp
for own in 1 0; do
  rm -rf gpurun_out/prof_os; mkdir -p gpurun_out/prof_os
  OWN=$own CSDR_CP_SKIP=parity,plain,retune rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_os -o t -- python3 tools/experiments/r6_own_stream.py > gpurun_out/os_$own.log 2>&1
  grep '^{' gpurun_out/os_$own.log
  python3 tools/trace_timeline.py gpurun_out/prof_os 3
done
rm -rf gpurun_out/prof_os
