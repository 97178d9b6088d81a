#!/usr/bin/env python3
"""GPU box, round 6: after bench.py's old control_plane sequence (objects dropped and remade between 530 MB of torch
clones), is device memory handed out in pieces that stream slowly?  Sixty fresh 24 MB allocations, the time of an
in-place add over each (read + write), before and after the sequence."""
import gc, json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)

def probe(label, n=60, mb=24):
    torch.cuda.empty_cache()
    ts = [torch.zeros((mb << 20) // 4, device="cuda", dtype=torch.float32) for _ in range(n)]
    torch.cuda.synchronize()
    us = []
    for t in ts:
        for _ in range(3):
            t.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            t.add_(1.0)
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 100.0)
    us.sort()
    print(json.dumps({label: {"min_us": round(us[0], 1), "median_us": round(us[len(us) // 2], 1), "max_us": round(us[-1], 1),
                              "slower_than_1.5x_median": sum(1 for u in us if u > 1.5 * us[len(us) // 2])}}))
    del ts
    gc.collect(); torch.cuda.empty_cache()

def remake(w, pipelined):
    if w.b is not None:
        w.b.flush(w.stream); torch.cuda.synchronize()
    w.b = None; w.kept = {}; w.mode = None
    gc.collect()
    w.set_mode(pipelined)

probe("fresh process")
w = bench.C4Workload(torch, ca, ctx, 256)
remake(w, False)
keep = []
for rep in range(2):                     # what control_plane's parity passes did until round 6
    remake(w, False); remake(w, True)
    for k in range(3):
        w.step(); w.b.flush(w.stream); torch.cuda.synchronize()
        keep.append(w.aud[1:256:3, :w.T // 32].clone()); keep.append(w.aud[2:256:3, :w.T // 32].clone())
del keep
remake(w, False)
print(json.dumps({"strict_object_ms": round(bench.gpu_ms(torch, w.step, 8, 30), 3)}))
probe("beside the slow object")
remake(w, False)
print(json.dumps({"next_strict_object_ms": round(bench.gpu_ms(torch, w.step, 8, 30), 3)}))
