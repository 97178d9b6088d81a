#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE (KiB, as reported) of the kernels whose name contains $1, under the command that follows
export TMPDIR=/tmp
PAT=$1; shift
OUT=gpurun_out/pmc_fetch
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/a -- "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/b -- "$@" > $OUT/b.log 2>&1
PAT=$PAT python3 - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for tag in "ab":
    f = glob.glob("gpurun_out/pmc_fetch/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if pat in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, len(v), round(sum(v) / len(v), 1))
PY
