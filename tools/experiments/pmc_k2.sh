#!/bin/bash
# GPU box: dynamic instruction mix of the down-converter alone (tools/exp_k2_stride.py 0) from SQ counters.
export TMPDIR=/tmp
OUT=gpurun_out/k2pmc2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 tools/experiments/exp_k2_stride.py 0 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    f = glob.glob("gpurun_out/k2pmc2/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "downconv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), sum(v) / len(v))
PY
