#!/usr/bin/env python3
"""Condense a tools/collect_chain_profile.sh run into profiles/<tag>_chain_kernel_stats.csv and
profiles/<tag>_chain_summary.json: per kernel (and grid size: the same kernel runs with different shapes) the
average duration from the kernel trace and the mean FETCH_SIZE / WRITE_SIZE per launch from the PMC passes.
HBM bytes are given both ways -- FETCH_SIZE doubled (the guide's gfx950 correction, calibrated for wide
16-byte-per-lane streaming reads) and as counted -- and, where the workload's byte count is known exactly, the
ratio of each to it: the calibration the guide asks for before trusting an absolute."""
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
C, T = 256, 1 << 21
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        rows.append([n if len(n) < 100 else n[:97] + "...", r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
with open("profiles/%s_chain_kernel_stats.csv" % tag, "w", newline="") as fo:
    w = csv.writer(fo); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]); w.writerows(rows)
# per (kernel, grid) durations from the trace
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr::" in r["Kernel_Name"]:
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])      # the PMC files carry the product
            dur[(r["Kernel_Name"].split("(")[0], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "csdr::" in r["Kernel_Name"]:
                g = int(r["Grid_Size"])
                pmc[(r["Kernel_Name"].split("(")[0], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = None
for l in open(os.path.join(out, "chain.log")):
    if l.startswith("{"): line = json.loads(l)
npk = (T // 240) // 8 * 8
known = {   # kernel substring -> (what, algorithmic bytes per launch of the stand-alone measurement, grid predicate)
    "downconv_kernel": ("K2 alone: 256 ch x 2^21 in, /32 out", C * T * (8 + 8 / 32.0)),
    "spectrum16_kernel": ("K3: 256 ch x 512 frames x 4096, 8 B in + 4 B out per bin (SURVEY 8d); the kernel keeps the running sums in "
                            "registers and writes only the last frame's bels, so its traffic is the 8 B in", C * 512 * 4096 * 12),
    "noiseblank_kernel": ("K6 blanker: 256 ch x 2^21, 8 B in + 8 B out", C * T * 16),
    "unpack_kernel": ("K6 unpack 24 bit: 6 B in + 8 B out per sample", C * (T // 240) * 240 * 14),
}
kern = {}
for key, cs in sorted(pmc.items()):
    name, grid = key
    d = dur.get(key, [])
    e = {"grid": grid, "launches_timed": len(d), "avg_us": round(sum(d) / len(d) / 1e3, 2) if d else None}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in cs: e[c + "_KiB_mean"] = round(sum(cs[c]) / len(cs[c]), 1)
    if "FETCH_SIZE_KiB_mean" in e and "WRITE_SIZE_KiB_mean" in e:
        e["hbm_bytes_fetch_doubled"] = (2 * e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024
        e["hbm_bytes_as_counted"] = (e["FETCH_SIZE_KiB_mean"] + e["WRITE_SIZE_KiB_mean"]) * 1024
    kern["%s grid %d" % (name, grid)] = e
# the stand-alone measurements (all 256 channels in one launch)
for sub, (what, alg) in known.items():
    # the stand-alone launch of a kernel: the shape whose counted traffic is nearest the known byte count
    cands = [(abs(kern["%s grid %d" % k].get("hbm_bytes_fetch_doubled", 0.0) / alg - 1.0) if len([q for q in pmc if sub in q[0]]) > 1 else 0.0, k)
             for k in pmc if sub in k[0]]
    if not cands: continue
    k = min(cands)[1]
    e = kern["%s grid %d" % k]
    e["standalone"] = what
    e["algorithmic_bytes"] = alg
    if e["avg_us"]:
        e["achieved_GBps"] = round(alg / (e["avg_us"] * 1e-6) / 1e9, 1)
        e["frac_of_8TBps"] = round(alg / (e["avg_us"] * 1e-6) / 8e12, 4)
    if "hbm_bytes_fetch_doubled" in e:
        e["traffic_over_algorithmic_fetch_doubled"] = round(e["hbm_bytes_fetch_doubled"] / alg, 3)
        e["traffic_over_algorithmic_as_counted"] = round(e["hbm_bytes_as_counted"] / alg, 3)
        if e["avg_us"]:     # the fraction on the bytes the counters saw move (K3 writes a third of what SURVEY 8d prices)
            e["GBps_on_counted_traffic"] = round(e["hbm_bytes_fetch_doubled"] / (e["avg_us"] * 1e-6) / 1e9, 1)
            e["frac_of_8TBps_on_counted_traffic"] = round(e["hbm_bytes_fetch_doubled"] / (e["avg_us"] * 1e-6) / 8e12, 4)
json.dump({"tag": tag, "command": "tools/collect_chain_profile.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in their own passes, each around python3 tools/bench_chain.py",
           "bench_chain_line_under_profiler": line, "kernels": kern,
           "notes": "FETCH_SIZE / WRITE_SIZE in KiB.  On gfx950 FETCH_SIZE counts half the bytes of a wide (16 B per lane) coalesced streaming "
                    "read (MI355X_MICROARCH.md, HBM); other access widths are uncalibrated, hence both readings and their ratio to the "
                    "exactly known byte count of each stand-alone measurement."},
          open("profiles/%s_chain_summary.json" % tag, "w"), indent=1)
for k, e in kern.items():
    if "standalone" in e: print(k, {x: e[x] for x in e if x in ("avg_us", "achieved_GBps", "frac_of_8TBps", "frac_of_8TBps_on_counted_traffic", "traffic_over_algorithmic_fetch_doubled", "traffic_over_algorithmic_as_counted")})
