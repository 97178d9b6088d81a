#!/usr/bin/env python3
"""GPU box, round 6: which history makes the next strict object slow?  SEQ = letters: s = make a strict object and drop it
unused, S = make one and run it, p / P = the same for a pipelined one, c / C = twelve 22 MB torch tensors made and dropped /
kept (bench.py's control_plane clones that much audio between objects); the final strict object is then made and timed."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)

def remake(w, pipelined):
    """drop every batch object the workload holds and make a new one (what C4Workload.set_mode did on every switch until
    round 6)"""
    import gc
    if w.b is not None:
        w.b.flush(w.stream); torch.cuda.synchronize()
    w.b = None; w.kept = {}; w.mode = None
    gc.collect()
    w.set_mode(pipelined)
kept = []
for ch in os.environ.get("SEQ", ""):
    if ch in "cC":                       # c: twelve 22 MB torch tensors made and dropped (into torch's cache); C: kept
        t = [torch.empty((85, 65536), device="cuda", dtype=torch.float32) for _ in range(12)]
        if ch == "C":
            kept.append(t)
        del t
        continue
    remake(w, ch in "pPT")
    if ch == "T":                        # a pipelined object whose FM / USB receivers are set to what they already are
        base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                    FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                    AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
        modes = [None, (ca.DEMOD_FM, dict()),
                 (ca.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]
        for k in range(3):
            if k > 0:
                for c in range(w.C):
                    if c % 3:
                        m, kw = modes[c % 3]
                        w.b.set_freq(c, -(100e3 + 500.0 * (c % 1024)))
                        w.b.set_demod(c, m, ca.DemodInfo(**dict(base, **kw)))
            w.step()
            w.b.flush(w.stream); torch.cuda.synchronize()
    if ch in "SP":
        for _ in range(3):
            w.step()
        w.b.flush(w.stream); torch.cuda.synchronize()
remake(w, False)
print(json.dumps({"seq": os.environ.get("SEQ", ""), "ms": round(bench.gpu_ms(torch, w.step, 8, 30), 3)}))
