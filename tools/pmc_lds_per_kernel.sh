#!/bin/bash
# GPU box: LDS pipe cycles per LDS instruction, and LDS / VALU busy, of every csdr kernel under the command that follows
# (a misaligned or conflicted access pattern shows as many cycles per instruction)
#   tools/pmc_lds_per_kernel.sh python3 tools/bench_c4_strict.py
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- "$@" > $OUT/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_lds/a/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "csdr" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].replace("void csdr::", "").replace("csdr::", "").split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    print("%-60s n=%-4d cyc=%9.0f LDS busy %5.1f %%  cycles/LDSinstr %5.1f  conflict %4.1f %%  unaligned %4.1f %%  VALU %5.1f %%" % (
        k, len(c["GRBM_GUI_ACTIVE"]), cyc, 100 * m["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, m["SQ_LDS_IDX_ACTIVE"] / max(1.0, m["SQ_INSTS_LDS"]),
        100 * m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"]), 100 * m["SQ_LDS_UNALIGNED_STALL"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"]),
        100 * m["SQ_INSTS_VALU"] * 4 / 1024 / cyc))
PY
