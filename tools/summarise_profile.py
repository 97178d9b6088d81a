#!/usr/bin/env python3
"""Condense a tools/collect_profile.sh run into profiles/<tag>_*.{csv,json} (small, tracked)."""
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
# kernel stats: keep our kernels + a one-line rest
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        short = name if len(name) < 120 else name[:117] + "..."
        rows.append([short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(rows)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "csdr::" in k:
            pmc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in pmc.items():
    summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()}
bench = None
for line in open(os.path.join(out, "bench_trace.log")):
    if line.startswith("{"):
        bench = json.loads(line)
sys.path.insert(0, root)
import bench as benchmod
doc = {"tag": tag, "command": "python3 bench.py --no-cpu --no-check --no-secondary (kernel-trace run); the same with --steps 5 --warmup 1 for every PMC pass",
       "k1_source_sha16": benchmod.k1_source_hash(),
       "bench_line_under_profiler": bench, "pmc_mean_per_launch": summary,
       "notes": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced "
                "stream's bytes (MI355X_MICROARCH.md, HBM) so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."}
for k, cs in summary.items():
    if "fastfir_os2_kernel<14>" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        hbm = (2 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024
        doc["hbm_bytes_per_launch"] = hbm
        doc["hbm_bytes_over_algorithmic"] = hbm / (16.0 * 256 * (1 << 19))
        json.dump({"hbm_bytes_per_launch": hbm, "source": tag, "k1_source_sha16": benchmod.k1_source_hash(),
                   "kernel": k}, open(os.path.join(prof, "traffic_latest.json"), "w"))
        if "GRBM_GUI_ACTIVE" in cs:
            doc["grbm_gui_active_per_launch_sum_over_8_xcds"] = cs["GRBM_GUI_ACTIVE"]["mean"]
    if "fastfir_os2_kernel<14>" in k and "SQ_WAVE_CYCLES" in cs:
        wc = cs["SQ_WAVE_CYCLES"]["mean"]
        doc["sq_shares_of_wave_cycles"] = {c: round(cs[c]["mean"] / wc, 4) for c in cs if c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and c != "SQ_WAVES"}
        if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_ACTIVE_INST_LDS" in cs:
            doc["lds_bank_conflict_over_lds_active"] = round(cs["SQ_LDS_BANK_CONFLICT"]["mean"] / cs["SQ_ACTIVE_INST_LDS"]["mean"], 4)
json.dump(doc, open(os.path.join(prof, tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k != "pmc_mean_per_launch"})[:600])
