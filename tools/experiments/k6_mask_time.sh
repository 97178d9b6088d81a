#!/bin/bash
# GPU box: average duration of the mask-form blanker and of the fused down-converters inside the datagram-fed C4 chain
#   [ENV=..] tools/k6_mask_time.sh
export TMPDIR=/tmp
OUT=gpurun_out/k6t
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/experiments/bench_blank_widths.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/k6t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "noiseblank" in r["Name"] or "downconv" in r["Name"]:
        print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
