"""round 5: does the headline kernel's time depend on the DATA (it runs at the chip's power limit)?  Same launch, three inputs:
bench.py's (Gaussian noise at -20 dBFS), SURVEY 8d's C3 recipe (three tones at -20 dBFS + noise at -70 dBFS), zeros."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C, T, n = 256, 1 << 19, 16384
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
def run(x, label):
    y = torch.empty_like(x)
    ff = ca.FastFirBatch(C, n); ff.setup(-5000, 5000, 0, 62500.0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({label: round(e0.elapsed_time(e1) / 100, 4)}), flush=True)
noise = torch.randn((C, T, 2), device=dev) * 3276.7
t = torch.arange(T, device=dev, dtype=torch.float64)
ph = lambda f: 2 * torch.pi * f * t / 62500.0
tones = sum(3276.7 * torch.stack([torch.cos(ph(f)), torch.sin(ph(f))], -1) for f in (1000.0, -2340.0, 12000.0)).float()
sig = tones.unsqueeze(0).expand(C, T, 2).contiguous() + torch.randn((C, T, 2), device=dev) * 10.36
for rep in range(2):
    run(noise, "noise_-20dBFS")
    run(sig, "3tones_-20dBFS_noise_-70dBFS")
    run(torch.zeros_like(noise), "zeros")
