#!/usr/bin/env python3
"""The C4 per-GPU share through csdr_demod_batch, pipelined and strict, on fresh objects: bench.py's C4Workload
without the rest of the line (for A/B runs of chain experiments: CSDR_LIB_PATH=<alt build> python tools/bench_c4_modes.py)."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
out = {}
for mode in (True, False, True):
    w.set_mode(mode)
    elapsed, ms = bench.timed_steps(torch, ctx, w.step, 30, 15, prewarm=False)
    out.setdefault(w.mode, []).append(round(elapsed / 30 * 1e3, 4))
print(json.dumps(out))
