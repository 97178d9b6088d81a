#!/bin/bash
# GPU box: is the down-converter bound by the LDS pipe?  LDS activity / conflicts of the kernel alone (tools/exp_k2_stride.py 0)
export TMPDIR=/tmp
OUT=gpurun_out/k2pmc3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/b -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    f = glob.glob("gpurun_out/k2pmc3/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "downconv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), sum(v) / len(v))
PY
tail -2 $OUT/a.log
