#!/usr/bin/env python3
"""Strict-mode C4 share with every call cut into P pieces in time (P consecutive calls of T/P samples): does the
post-chain of piece k overlapping the down-converters of piece k+1 beat one call's longer serial tail?"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(False)
out = {}
for P in (1, 2, 4):
    n = w.T // P
    def step():
        for k in range(P):
            w.b.process_ptr(w.x.data_ptr() + 8 * k * n, w.T, n, w.aud.data_ptr(), w.cap, w.stream)
    out["pieces_%d_ms" % P] = round(bench.gpu_ms(torch, step, 10, 20), 4)
print(json.dumps(out))
