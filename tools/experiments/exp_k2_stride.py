#!/usr/bin/env python3
"""K2 alone on 256 ch x 2^21 with the channel rows at a power-of-two pitch and at padded pitches: do all
workgroups marching through rows 16 MiB apart camp on the same HBM channels?  (experiment, GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 21
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
res = {}
for pad in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1040").split(",")]:
    S = T + pad
    x = torch.randn((C, S, 2), device=dev, dtype=torch.float32) * 100.0
    y = torch.empty((C, T // 16, 2), device=dev, dtype=torch.float32)
    dc = ca.DownConvertBatch(C)
    dc.set_data_rate(2e6, 15000.0)
    for c in range(C): dc.set_frequency(-100e3 - 500.0 * c, channel=c)
    f = lambda: dc.process_ptr(x.data_ptr(), S, T, y.data_ptr(), T // 16, stream)
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    res["pad_%d" % pad] = round(e0.elapsed_time(e1) / 50, 4)
    dc.close(); del x, y
print(json.dumps(res))
