#!/bin/bash
# GPU box, round 4: strict-mode C4 chain with the plan groups interleaved (0), phased with concurrent down-converters
# (1) and phased with chained down-converters (2); then the kernel timeline of the phased form.
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
for m in 0 1 2 0 1 2; do
  echo "phased=$m $(CSDR_CHAIN_PHASED=$m python3 tools/bench_c4_strict.py 2>&1 | tail -1)" | tee -a gpurun_out/r4_phased.log
done
for m in 1 0; do
rm -rf gpurun_out/prof_ph$m
CSDR_CHAIN_PHASED=$m rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ph$m -o t -- python3 tools/bench_c4_strict.py > gpurun_out/ph${m}_trace.log 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_ph$m 2 > gpurun_out/r4_phased_timeline$m.txt 2>&1
rm -rf gpurun_out/prof_ph$m
done
tail -3 gpurun_out/r4_phased_timeline1.txt
