#!/bin/bash
# GPU box: FETCH_SIZE per load width on a known byte count (tools/ubench/fetch_calib.hip) -> profiles/<tag>_fetch_calib.json
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT profiles
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o t -- ./tools/ubench/fetch_calib > $OUT/run.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
bytes_read = float(1 << 30)
res = {"bytes_read_per_launch": bytes_read, "unit": "FETCH_SIZE in KiB as rocprofv3 reports it", "kernels": {}}
for k, v in sorted(acc.items()):
    m = sum(v) / len(v)
    res["kernels"][k] = {"launches": len(v), "FETCH_SIZE_KiB_mean": round(m, 1), "reported_over_true": round(m * 1024 / bytes_read, 4)}
json.dump(res, open("profiles/%s_fetch_calib.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp profiles/${TAG}_fetch_calib.json gpurun_out/
