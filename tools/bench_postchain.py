#!/usr/bin/env python3
"""K4 alone: the chain on N receivers of ONE mode (one decimator-plan group, so its launches run one after the
other with the chip to themselves), timed per kernel with HIP events around the whole step and, under
rocprofv3 --kernel-trace, per launch.  usage: bench_postchain.py [channels]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C = int(sys.argv[1]) if len(sys.argv) > 1 else 85
T = 1 << 21
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
FS, A = 2e6, 3276.7
t = torch.arange(T, device=dev, dtype=torch.float64) / FS
out = {}
base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
            FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100, AgcManualGain=30,
            AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
modes = {"AM": (0, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)), "FM": (2, dict()),
         "USB": (3, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))}
for name, (m, kw) in modes.items():
    x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
    for c in range(C):
        fc = 100e3 + 500.0 * c
        if name == "AM":
            ph = 2 * torch.pi * fc * t; amp = A * (1.0 + 0.5 * torch.sin(2 * torch.pi * 1000.0 * t))
            x[c, :, 0] += (amp * torch.cos(ph)).float(); x[c, :, 1] += (amp * torch.sin(ph)).float()
        elif name == "FM":
            ph = 2 * torch.pi * fc * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
            x[c, :, 0] += (A * torch.cos(ph)).float(); x[c, :, 1] += (A * torch.sin(ph)).float()
        else:
            for off in (1200.0, 2340.0):
                ph = 2 * torch.pi * (fc + off) * t
                x[c, :, 0] += (0.5 * A * torch.cos(ph)).float(); x[c, :, 1] += (0.5 * A * torch.sin(ph)).float()
    b = ca.DemodBatch(C, 2048); b.set_input_rate(FS)
    for c in range(C): b.set_demod(c, m, ca.DemodInfo(**dict(base, **kw)))
    b.commit()
    for c in range(C): b.set_freq(c, -(100e3 + 500.0 * c))
    aud = torch.empty((C, T // 16 + 4096), device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    def step(): b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), T // 16 + 4096, st)
    for _ in range(10): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 20, 3)
    del x, b, aud
print(json.dumps({"channels": C, "chain_ms_per_step": out}))
