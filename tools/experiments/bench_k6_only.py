#!/usr/bin/env python3
"""The noise blanker alone on 256 receivers x 2^21 samples (for counter passes: rocprofv3 --pmc ... -- python3 tools/bench_k6_only.py)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 21
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * 1000.0
xb = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
nbk = ca.NoiseProcBatch(C); nbk.setup(True, 50.0, 2.0, 2e6)
f = lambda: nbk.process_ptr(x.data_ptr(), T, T, xb.data_ptr(), T, st)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"k6_ms": round(e0.elapsed_time(e1) / 10, 4)}))
