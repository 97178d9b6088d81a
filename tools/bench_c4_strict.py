#!/usr/bin/env python3
"""The C4 per-GPU share through csdr_demod_batch in strict mode only (one kernel at a time: for kernel traces)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(False)
elapsed, ms = bench.timed_steps(torch, ctx, w.step, 20, 10, prewarm=False)
print(json.dumps({w.mode: round(elapsed / 20 * 1e3, 4)}))
