#!/usr/bin/env python3
"""Can the down-converter's waves share a CU with a post-chain walk?  256 FM receivers in one group (one walk workgroup
per CU), pipelined: the kernel trace shows whether the next call's down-converter starts while the walk still runs."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 1 << 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * 1000.0
b = ca.DemodBatch(C, 2048); b.set_input_rate(2e6)
for c in range(C): b.set_demod(c, 2, ca.fm_defaults())
b.commit()
for c in range(C): b.set_freq(c, -(100e3 + 500.0 * c))
if len(sys.argv) > 2 and sys.argv[2] == "pipe":
    b.set_pipelined(True)
aud = torch.empty((C, T // 16 + 4096), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
step = lambda: b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), T // 16 + 4096, st)
for _ in range(6): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(round(e0.elapsed_time(e1) / 10, 4))
