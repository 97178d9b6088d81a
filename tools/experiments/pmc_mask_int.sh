#!/bin/bash
# GPU box: what keeps the integer mask kernel busy (SQ counters of noiseblank_mask_int_kernel inside the datagram chain
# with the blanker; counter passes of their own, no trace domain beside them).  CSDR_LIB_PATH picks the library.
export TMPDIR=/tmp
OUT=gpurun_out/maskpmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/experiments/bench_blank_widths.py > $OUT/a.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 tools/experiments/bench_blank_widths.py > $OUT/b.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/c -- python3 tools/experiments/bench_blank_widths.py > $OUT/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    f = glob.glob("gpurun_out/maskpmc/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "noiseblank_mask_int" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), sum(v) / len(v))
PY
rm -rf $OUT/a $OUT/b $OUT/c      # (the raw counter tables are tens of MB)
