#!/usr/bin/env python3
"""Host time of each pipelined csdr_demod_batch call on the C4 share (is the host ahead of the GPU?)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(len(sys.argv) < 2 or sys.argv[1] != "strict")
for _ in range(10):
    w.step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); w.step(); ts.append(round((time.perf_counter() - a) * 1e6))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_us_per_call": ts, "enqueue_ms": round((t1 - t0) * 1e3, 2), "drain_ms": round((t2 - t1) * 1e3, 2)}))
