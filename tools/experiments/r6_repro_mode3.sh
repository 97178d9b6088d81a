#!/bin/bash
# GPU box, round 6: the strict object made right after bench.py's control_plane (slow) and a second one (fast): one step's
# timeline of each
export TMPDIR=/tmp
for slow in 1; do
  rm -rf gpurun_out/prof_m3; mkdir -p gpurun_out/prof_m3
  SLOW=$slow rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_m3 -o t -- python3 tools/experiments/r6_repro_mode3.py > gpurun_out/m3_$slow.log 2>&1
  grep '"ms"' gpurun_out/m3_$slow.log
  TIMELINE_ALL=1 python3 tools/trace_timeline.py gpurun_out/prof_m3 3
done
rm -rf gpurun_out/prof_m3
