#!/bin/bash
# GPU box: dynamic instruction mix of the post-chain kernel (tools/bench_postchain.py: 85 receivers of one mode per
# group, AM / FM / USB in this order) from SQ counters, per mode (dispatches in launch order).
export TMPDIR=/tmp
OUT=gpurun_out/k4pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 tools/bench_postchain.py 85 > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/k4pmc/a/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "postchain_kernel" in r["Kernel_Name"]]
disp = collections.OrderedDict()
for r in rows: disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
n = len(ids) // 3
for name, chunk in (("AM", ids[:n]), ("FM", ids[n:2 * n]), ("USB", ids[2 * n:])):
    acc = collections.defaultdict(float)
    for i in chunk[len(chunk) // 2:]:
        for k, v in disp[i].items(): acc[k] += v / (len(chunk) - len(chunk) // 2)
    print(name, len(chunk), {k: round(v) for k, v in sorted(acc.items())})
PY
