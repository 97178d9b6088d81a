#!/bin/bash
# GPU box: LDS / VALU / wait counters of the kernels whose name contains $1, under the command that follows
#   tools/pmc_any.sh spectrum_kernel python3 tools/bench_k3.py
export TMPDIR=/tmp
PAT=$1; shift
OUT=gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- "$@" > $OUT/b.log 2>&1
PAT=$PAT python3 - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
tot = {}
for tag in "ab":
    f = glob.glob("gpurun_out/pmc_any/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        tot[k] = sum(v) / len(v); print(k, len(v), tot[k])
cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
print("cycles/launch %.0f  LDS busy %.1f %%  VALU busy %.1f %% (4 clk/instr)  waves %d" % (
    cyc, 100 * tot["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 100 * tot["SQ_INSTS_VALU"] * 4 / 1024 / cyc, tot["SQ_WAVES"]))
PY
