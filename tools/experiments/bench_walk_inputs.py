#!/usr/bin/env python3
"""Post-chain walk time by mode and by what the receivers hear: a carrier (bench.py's C4 signals) or noise only (an
idle channel: PLL unlocked, AGC riding the noise).  Run under rocprofv3 --kernel-trace for the walk's own duration;
prints the chain's ms per call.   usage: tools/bench_walk_inputs.py <AM|SAM|FM|USB|CWU> <carrier|noise>"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
mode, what = sys.argv[1], sys.argv[2]
C, T, FS, A = 85, 1 << 20, 2e6, 3276.7
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
t = torch.arange(T, device=dev, dtype=torch.float64) / FS
if what == "carrier":
    for c in range(C):
        fc = 100e3 + 500.0 * c
        if mode in ("AM", "SAM"):
            ph = 2 * torch.pi * fc * t; amp = A * (1.0 + 0.5 * torch.sin(2 * torch.pi * 1000.0 * t))
            x[c, :, 0] += (amp * torch.cos(ph)).float(); x[c, :, 1] += (amp * torch.sin(ph)).float()
        elif mode == "FM":
            ph = 2 * torch.pi * fc * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
            x[c, :, 0] += (A * torch.cos(ph)).float(); x[c, :, 1] += (A * torch.sin(ph)).float()
        else:
            for off in (1200.0, 2340.0) if mode == "USB" else (0.0,):
                ph = 2 * torch.pi * (fc + off) * t
                x[c, :, 0] += (0.5 * A * torch.cos(ph)).float(); x[c, :, 1] += (0.5 * A * torch.sin(ph)).float()
base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
            FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
            AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
MODES = {"FM": (2, dict()), "AM": (0, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)),
         "SAM": (1, dict(HiCutmin=100, HiCutmax=10000, LowCutmax=-100, LowCutmin=-10000, Symetric=0)),
         "USB": (3, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0)),
         "CWU": (5, dict(HiCut=500, LowCut=-500, HiCutmin=50, HiCutmax=1000, LowCutmax=-50, LowCutmin=-1000, Offset=700, Symetric=0))}
m, kw = MODES[mode]
b = ca.DemodBatch(C, 2048); b.set_input_rate(FS)
for c in range(C): b.set_demod(c, m, ca.DemodInfo(**dict(base, **kw)))
b.commit()
for c in range(C): b.set_freq(c, -(100e3 + 500.0 * c))
cap = T // 16 + 4096
aud = torch.empty((C, cap), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
step = lambda: b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), cap, st)
for _ in range(6): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"mode": mode, "input": what, "ms_per_call": round(e0.elapsed_time(e1) / 10, 4)}))
