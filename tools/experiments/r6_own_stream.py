#!/usr/bin/env python3
"""GPU box, round 6: the strict C4 step with the caller's stream a stream of its own instead of the null stream (bench.py
runs on torch's default stream = the null stream), fresh and after bench.py's control_plane sequence"""
import gc, json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
own = os.environ.get("OWN", "1") == "1"
if os.environ.get("NULLFIRST"):          # some work on the default (null) stream before the host's own stream exists
    torch.zeros(1 << 20, device="cuda").add_(1); torch.cuda.synchronize()
if own:
    s = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "0")))
    torch.cuda.set_stream(s)
extra = []
for _ in range(int(os.environ.get("EXTRA", "0"))):    # more streams made (and used once) before the library makes its own
    e = torch.cuda.Stream(priority=int(os.environ.get("EXTRA_PRIO", "0")))
    with torch.cuda.stream(e):
        torch.zeros(8, device="cuda").add_(1)
    extra.append(e)
torch.cuda.synchronize()
w = bench.C4Workload(torch, ca, ctx, 256)
out = {"own_stream": own, "extra": len(extra), "caller_stream": hex(w.stream)}
w.set_mode(False)
out["fresh"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
w.control_plane(ctx)
w.b.flush(w.stream); torch.cuda.synchronize()
w.b = None; w.kept = {}; w.mode = None
gc.collect()
w.set_mode(False)
out["after_control_plane"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
print(json.dumps(out))
