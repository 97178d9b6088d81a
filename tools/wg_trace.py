#!/usr/bin/env python3
"""One strict (or pipelined) step of the C4 chain drawn WORKGROUP BY WORKGROUP: a diagnostic copy of the library
(-DCSDR_WG_TRACE, cutesdr_amd/csrc/wg_trace.hpp) in which every workgroup of the chain's kernels records its start, its
end and the CU it ran on.  rocprofv3's kernel trace has one begin / end per launch; this shows when the workgroups of
a launch really became resident beside the other kernels, and on which CUs.
  step 1 (here, no GPU):  python tools/wg_trace.py build        -> cutesdr_amd/_var/trace/libcutesdr_mi_trace.so
  step 2 (GPU box):       CSDR_LIB_PATH=cutesdr_amd/_var/trace/libcutesdr_mi_trace.so python tools/wg_trace.py run [strict|pipelined] [out.json]
The traced build's own step time is a little longer than the product's (one atomic and four stores per workgroup)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = {1: "downconv", 2: "fastfir", 3: "smeter_call", 4: "agc_peaks", 5: "walk", 6: "sq_maps", 7: "sq_decide", 8: "sq_apply"}


def build():
    from cutesdr_amd import _build
    print(_build.build(variant="trace", variant_flags=["-DCSDR_WG_TRACE"]))


def run(mode="strict", out=None):
    import numpy as np
    import torch
    import cutesdr_amd as ca
    import bench
    ctx = bench.dist_init()
    torch.cuda.set_device(0)
    if os.environ.get("WGTRACE_OWN_STREAM"):           # the caller on a stream of its own instead of the null stream
        torch.cuda.set_stream(torch.cuda.Stream())
    w = bench.C4Workload(torch, ca, ctx, 256)
    w.set_mode(mode == "pipelined")
    if os.environ.get("WGTRACE_PRE") == "control_plane":
        # the object made right after bench.py's control_plane, everything else dropped: the slow kind of
        # tools/experiments/r6_repro_mode3.py (HISTORY round 6: a batch object's step time is bistable)
        import gc
        w.control_plane(ctx)
        w.b.flush(w.stream); torch.cuda.synchronize()
        w.b = None; w.kept = {}; w.mode = None
        gc.collect()
        w.set_mode(mode == "pipelined")
    for _ in range(12):
        w.step()
    torch.cuda.synchronize()
    cap = 1 << 17
    buf = torch.zeros((4 + 4 * cap,), device="cuda", dtype=torch.int64)
    buf[1] = cap
    L = ca.lib()
    L.csdr__wgtrace_set.argtypes = [C.c_void_p]
    torch.cuda.synchronize()
    assert L.csdr__wgtrace_set(C.c_void_p(buf.data_ptr())) == 0
    nsteps = 3
    for _ in range(nsteps):
        w.step()
    if mode == "pipelined":
        w.b.flush(w.stream)
    torch.cuda.synchronize()
    L.csdr__wgtrace_set(None)
    a = buf.cpu().numpy().astype(np.uint64)
    n = int(a[0])
    assert n <= cap, "trace buffer too small: %d records" % n
    r = a[4:4 + 4 * n].reshape(n, 4)
    kind = (r[:, 0] >> np.uint64(56)).astype(int)
    launch = ((r[:, 0] >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(int)
    t0 = r[:, 1].astype(np.int64); t1 = r[:, 2].astype(np.int64)
    blk = (r[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    hw = r[:, 3] & np.uint64(0xFFFFFFFF)
    xcc = (r[:, 3] >> np.uint64(32)).astype(int) & 15
    cu = (xcc * 1024 + (((hw >> np.uint64(13)) & np.uint64(7)).astype(int)) * 32 + (((hw >> np.uint64(12)) & np.uint64(1)).astype(int)) * 16
          + ((hw >> np.uint64(8)) & np.uint64(15)).astype(int))
    # the LAST step: launches are numbered in host issue order; a step issues the same number of launches each time
    per_step = launch.max() // nsteps
    first = launch.max() - per_step + 1
    sel = launch >= first
    base = t0[sel].min()
    us = lambda t: (t - base) / 100.0                         # 100 MHz ticks -> us
    rows = []
    for ln in range(first, launch.max() + 1):
        m = launch == ln
        if not m.any():
            continue
        s, e = us(t0[m]), us(t1[m])
        order = np.argsort(s)
        # "rounds": workgroups that start after the first workgroup of the launch has already ended
        late = int((s > e.min()).sum())
        rows.append({"launch": int(ln - first), "kernel": KINDS.get(int(kind[m][0]), "?"), "wgs": int(m.sum()),
                     "first_start_us": round(float(s.min()), 1), "median_start_us": round(float(np.median(s)), 1),
                     "last_start_us": round(float(s.max()), 1), "first_end_us": round(float(e.min()), 1),
                     "last_end_us": round(float(e.max()), 1), "median_wg_us": round(float(np.median(e - s)), 1),
                     "max_wg_us": round(float((e - s).max()), 1), "wgs_started_after_first_end": late,
                     "cus_used": int(len(set(cu[m].tolist())))})
    span = max(x["last_end_us"] for x in rows)
    res = {"mode": mode, "step_span_us": round(span, 1), "launches": rows}
    # how many down-converter workgroups share a CU with a walk, per down-converter launch
    walks = [(us(t0[i]), us(t1[i]), cu[i]) for i in np.nonzero(sel & (kind == 5))[0]]
    for x in rows:
        if x["kernel"] != "downconv":
            continue
        m = launch == first + x["launch"]
        s, e, c = us(t0[m]), us(t1[m]), cu[m]
        on_walk_cu = np.zeros(len(s), bool)
        for ws, we, wc in walks:
            on_walk_cu |= (c == wc) & (s < we) & (e > ws)
        x["wgs_beside_a_walk_on_its_cu"] = int(on_walk_cu.sum())
        x["wg_us_beside_walk_vs_not"] = [round(float(np.median((e - s)[on_walk_cu])), 1) if on_walk_cu.any() else None,
                                         round(float(np.median((e - s)[~on_walk_cu])), 1) if (~on_walk_cu).any() else None]
        # where the long workgroups are: the spread of the durations, by XCD, by the CU's share of the launch, by segment
        d = e - s
        x["wg_us_percentiles_5_25_50_75_95_100"] = [round(float(v), 1) for v in np.percentile(d, [5, 25, 50, 75, 95, 100])]
        xm = xcc[m]
        x["wg_us_median_by_xcd"] = [round(float(np.median(d[xm == q])), 1) if (xm == q).any() else None for q in range(8)]
        per_cu = {}
        for cc in c.tolist():
            per_cu[cc] = per_cu.get(cc, 0) + 1
        load = np.array([per_cu[cc] for cc in c.tolist()])
        x["wgs_per_cu_min_max"] = [int(load.min()), int(load.max())]
        x["wg_us_median_by_wgs_on_its_cu"] = {str(int(q)): round(float(np.median(d[load == q])), 1) for q in sorted(set(load.tolist()))}
        b = blk[m]
        # a launch is channels x segments, segment = block % nseg; nseg is not known here: the LAST blocks of the grid in
        # issue order against the first ones tells whether late issue matters
        order = np.argsort(b)
        q4 = len(order) // 4
        x["wg_us_median_by_block_quarter"] = [round(float(np.median(d[order[i * q4:(i + 1) * q4]])), 1) for i in range(4)]
        x["end_us_percentiles_50_90_99_100"] = [round(float(v), 1) for v in np.percentile(e, [50, 90, 99, 100])]
        # start-time histogram in 50 us bins
        h, _ = np.histogram(s - s.min(), bins=np.arange(0, max(100.0, (s - s.min()).max() + 50.0), 50.0))
        x["start_hist_50us"] = h.tolist()
    txt = json.dumps(res, indent=1)
    print(txt)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    else:
        run(*(sys.argv[2:4]))
