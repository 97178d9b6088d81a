#!/usr/bin/env python3
"""GPU box, round 6: how long the HOST takes to enqueue one C4 chain step (no waiting), strict and pipelined, against the
step's period -- is the pipelined mode's period the device's or the host's?"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
if os.environ.get("OWN"):                # the caller on a stream of its own instead of the null stream
    torch.cuda.set_stream(torch.cuda.Stream())
out = {}
for pipelined in (False, True):
    w = bench.C4Workload(torch, ca, ctx, 256)
    w.set_mode(pipelined)
    for _ in range(20):
        w.step()
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # the same with a synchronise after every step: the device's own time for one step, nothing queued behind it
    t3 = time.perf_counter()
    for _ in range(20):
        w.step(); torch.cuda.synchronize()
    t4 = time.perf_counter()
    out[w.mode] = {"host_enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 3), "period_ms": round((t2 - t0) / n * 1e3, 3),
                   "synchronous_step_ms": round((t4 - t3) / 20 * 1e3, 3)}
    del w
print(json.dumps(out))
