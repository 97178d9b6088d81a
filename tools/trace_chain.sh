#!/bin/bash
# GPU box: kernel timeline of the last steps of the chain workload (bench.py --workload c4) into gpurun_out/prof_c4
export TMPDIR=/tmp
rm -rf gpurun_out/prof_c4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -o t -- python3 bench.py --workload c4 --steps 10 --warmup 5 --no-cpu > gpurun_out/c4_trace.log 2>&1
grep "^{" gpurun_out/c4_trace.log | cut -c1-200
