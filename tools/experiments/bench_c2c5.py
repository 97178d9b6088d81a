#!/usr/bin/env python3
"""BASELINE configs C2 and C5 alone (bench.py's chain_one_receiver, no CPU legs): for A/B runs of post-chain changes."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
out = {}
for name in ("c2", "c5"):
    r = bench.chain_one_receiver(torch, ca, ctx, False, name)
    out[name] = {"ms_per_call": r["ms_per_call"], "x_real_time": r["x_real_time"]}
print(json.dumps(out))
