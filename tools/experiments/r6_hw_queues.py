#!/usr/bin/env python3
"""GPU box, round 6: how many streams of one process really run side by side?  K streams, one 1-ms spin kernel of ONE
workgroup each (torch.cuda._sleep), total time / 1 ms = how many queues' worth of serialisation.  By GPU_MAX_HW_QUEUES."""
import json, os, sys, time
import torch
torch.cuda.set_device(0)
cyc = 2_000_000
torch.cuda._sleep(cyc); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "one_ms": round(one * 1e3, 3), "rounds_by_streams": {}}
for prio in (0, -1):
    for K in (1, 2, 3, 4, 5, 6, 8, 12, 16):
        ss = [torch.cuda.Stream(priority=prio) for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in ss:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        out["rounds_by_streams"]["prio%d_K%d" % (prio, K)] = round((time.perf_counter() - t0) / one, 2)
        del ss
print(json.dumps(out))
