import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
# last step: find last 'postchain_kernel' end, go back 3 ms
end = max(r[1] for r in rows if "postchain_kernel" in r[2])
sel = [r for r in rows if r[0] > end - 2_600_000 and r[0] <= end]
# start at first downconv in window after a gap
t0 = None
for i, r in enumerate(sel):
    if "downconv" in r[2] and (i == 0 or r[0] - sel[i-1][1] > 20000):
        t0 = r[0]; sel = sel[i:]
if t0 is None: t0 = sel[0][0]
qs = {}
for s, e, n, q in sel:
    qs.setdefault(q, len(qs))
    print("%8.1f %8.1f q%d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, qs[q], n.replace("void csdr::", "").replace("csdr::", "")[:70]))
