#!/usr/bin/env python3
"""K3 across the single-pass sizes: 256 channels x 2^21 samples per launch (8 B per bin read), ms and TB/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 21
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
st = torch.cuda.current_stream().cuda_stream
out = {}
for n in ([int(a) for a in sys.argv[1:]] or (2048, 4096, 8192, 16384)):
    fb = ca.FftBatch(C); fb.set_params(n, False, 0.0, 2e6); fb.set_ave(1)
    f = lambda: fb.put_display_ptr(x.data_ptr(), T, T // n, st)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    out[n] = {"ms": round(ms, 4), "TBps_at_8B_per_bin": round(C * T * 8 / ms / 1e9, 2)}
print(json.dumps(out))
