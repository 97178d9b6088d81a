#!/usr/bin/env python3
"""K3 alone: 4096-pt display spectra, 256 channels x 512 frames per launch (BASELINE config C1's transform)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 21
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
st = torch.cuda.current_stream().cuda_stream
out = {}
for ave in (1, 4):
    fb = ca.FftBatch(C); fb.set_params(4096, False, 0.0, 2e6); fb.set_ave(ave)
    f = lambda: fb.put_display_ptr(x.data_ptr(), T, 512, st)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    out["ave%d_ms" % ave] = round(e0.elapsed_time(e1) / 30, 4)
print(json.dumps(out))
