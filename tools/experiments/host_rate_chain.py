#!/usr/bin/env python3
"""How long the HOST needs to enqueue one step of the chain workload (bench.py C4Workload), against how long the
GPU needs to run it: if the first is not well below the second the chain is launch-bound."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import bench, cutesdr_amd as ca
ctx = bench.dist_init()
w = bench.C4Workload(torch, ca, ctx, 256)
for _ in range(10): w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): w.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_enqueue_ms_per_step": round((t1 - t0) / 40 * 1e3, 3), "total_ms_per_step": round((t2 - t0) / 40 * 1e3, 3)}))
