#!/bin/bash
# GPU box: where the down-converter's wave time goes (LDS pipe, instruction fetch, waits) -- SQ counters, two passes.
export TMPDIR=/tmp
OUT=gpurun_out/k2pmc3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/a -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    f = glob.glob("gpurun_out/k2pmc3/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "downconv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), round(sum(v) / len(v)))
PY
