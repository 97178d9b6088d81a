import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C, T = 85, 1 << 21
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
FS, A = 2e6, 3276.7
t = torch.arange(T, device=dev, dtype=torch.float64) / FS
x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
for c in range(C):
    ph = 2 * torch.pi * (100e3 + 500.0 * c) * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
    x[c, :, 0] += (A * torch.cos(ph)).float(); x[c, :, 1] += (A * torch.sin(ph)).float()
b = ca.DemodBatch(C, 2048); b.set_input_rate(FS)
for c in range(C): b.set_demod(c, 2, ca.fm_defaults())
b.commit()
for c in range(C): b.set_freq(c, -(100e3 + 500.0 * c))
aud = torch.empty((C, T // 16 + 4096), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
step = lambda: b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), T // 16 + 4096, st)
for _ in range(10): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): step()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("CSDR_FM_DEFER", "1"), round(e0.elapsed_time(e1) / 30, 4))
