#!/bin/bash
# GPU box: what keeps the noise blanker busy -- LDS, VALU, waits (SQ counters of the kernel alone)
export TMPDIR=/tmp
OUT=gpurun_out/k6pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/experiments/bench_k6_only.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 tools/experiments/bench_k6_only.py > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    f = glob.glob("gpurun_out/k6pmc/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "noiseblank" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), sum(v) / len(v))
PY
tail -1 $OUT/a.log
