#!/usr/bin/env python3
"""K1 across FFT sizes on the C3 data volume (256 channels x 2^19 samples): ms per launch and
algorithmic GB/s (16 B per sample).  Secondary measurement, not the bench.py contract."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 19
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
res = {}
for n in (2048, 4096, 8192, 16384):
    ff = ca.FastFirBatch(C, n)
    ff.setup(-5000, 5000, 0, 62500.0)
    for _ in range(60): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    res[n] = {"ms": round(ms, 4), "GBps": round(C * T * 16 / ms / 1e6, 1)}
print(json.dumps(res))
