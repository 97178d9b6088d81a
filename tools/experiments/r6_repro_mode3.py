#!/usr/bin/env python3
"""GPU box, round 6: (under rocprofv3 --kernel-trace) the strict object made right after bench.py's control_plane -- 30 steps
at the end of the trace; SLOW=0 makes a second new strict object first (that one is fast)"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)

def remake(w, pipelined):
    """drop every batch object the workload holds and make a new one (what C4Workload.set_mode did on every switch until
    round 6)"""
    import gc
    if w.b is not None:
        w.b.flush(w.stream); torch.cuda.synchronize()
    w.b = None; w.kept = {}; w.mode = None
    gc.collect()
    w.set_mode(pipelined)
remake(w, False)
w.control_plane(ctx)
mode = os.environ.get("VARIANT", "")
if mode == "hold":                       # keep the pipelined object alive while the strict one is made
    keep = w.b; w.b = None; w.mode = None
if mode == "dummy":                      # fill what the pipelined object frees with something else first
    w.b.flush(w.stream); torch.cuda.synchronize(); w.b = None; w.mode = None
    import gc; gc.collect()
    holes = [ca.DeviceBuffer(sz) for sz in (1 << 20, 4 << 20, 16 << 20, 64 << 20, 1 << 20, 4 << 20, 16 << 20, 64 << 20)]
if mode == "empty":                      # give torch's cached blocks (control_plane's clones of the audio rows) back first
    w.b.flush(w.stream); torch.cuda.synchronize(); w.b = None; w.mode = None
    import gc; gc.collect()
    torch.cuda.empty_cache()
remake(w, False)
if os.environ.get("SLOW", "1") == "0":
    remake(w, False)
torch.cuda.synchronize()
ms = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
import ctypes as C
L = ca.lib()
L.csdr__demod_batch_probe.restype = C.c_int
L.csdr__demod_batch_probe.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
us, by = (C.c_double * 8)(), (C.c_double * 8)()
n = L.csdr__demod_batch_probe(w.b.h, us, by, 8)
print(json.dumps({"ms": ms, "group_buffers_copy_GBps": [round(2 * by[i] / us[i] / 1e3, 1) if us[i] else None for i in range(max(n, 0))],
                  "group_buffer_MB": [round(by[i] / 1e6, 1) for i in range(max(n, 0))]}))
