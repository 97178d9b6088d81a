#!/usr/bin/env python3
"""A call of the C4 share as P pieces through the PIPELINED object, flushed at the end (what a strict call cut into
pieces inside the library would cost): P = 1, 2, 4."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(True)
out = {}
for P in (1, 2, 4):
    n = w.T // P
    def step():
        for k in range(P):
            w.b.process_ptr(w.x.data_ptr() + 8 * k * n, w.T, n, w.aud.data_ptr(), w.cap, w.stream)
        w.b.flush(w.stream)
        torch.cuda.current_stream().synchronize()
    for _ in range(5): step()
    import time
    t0 = time.perf_counter()
    for _ in range(20): step()
    out["pieces_%d_flushed_ms" % P] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
print(json.dumps(out))
