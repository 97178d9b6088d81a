#!/usr/bin/env python3
"""256 mixed AM / FM / USB receivers cut from ONE 2 MSPS stream (csdr_demod_batch_set_input_rows): the C4 share with a
shared input instead of one stream per receiver -- the down-converters read 16 MB instead of 4.3 GB per call.
   usage: tools/bench_c4_shared.py [streams, default 1]"""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import numpy as np
import torch
import cutesdr_amd as ca
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C, T, FS = 256, 1 << 21, 2e6
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.randn((S, T, 2), generator=g, device=dev, dtype=torch.float32) * (32767.0 * 10 ** (-70 / 20))
t = torch.arange(T, device=dev, dtype=torch.float64) / FS
A = 3276.7 / 4
fc = lambda c: -960e3 + 7.5e3 * c
for c in range(C):
    ph = 2 * torch.pi * fc(c) * t
    if c % 3 == 0:
        amp = A * (1.0 + 0.5 * torch.sin(2 * torch.pi * 1000.0 * t))
        x[c % S, :, 0] += (amp * torch.cos(ph)).float(); x[c % S, :, 1] += (amp * torch.sin(ph)).float()
    elif c % 3 == 1:
        ph = ph + 2.0 * torch.sin(2 * torch.pi * 1000.0 * t)
        x[c % S, :, 0] += (A * torch.cos(ph)).float(); x[c % S, :, 1] += (A * torch.sin(ph)).float()
    else:
        ph = 2 * torch.pi * (fc(c) + 1200.0) * t
        x[c % S, :, 0] += (A * torch.cos(ph)).float(); x[c % S, :, 1] += (A * torch.sin(ph)).float()
base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
            FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
            AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
modes = [(ca.DEMOD_AM, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)), (ca.DEMOD_FM, dict()),
         (ca.DEMOD_USB, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0))]
out = {}
for pipelined in (True, False):
    b = ca.DemodBatch(C, 2048); b.set_input_rate(FS)
    for c in range(C):
        m, kw = modes[c % 3]
        b.set_demod(c, m, ca.DemodInfo(**dict(base, **kw)))
    b.set_input_rows(np.arange(C, dtype=np.int32) % S)
    b.commit()
    for c in range(C):
        b.set_freq(c, -fc(c))
    if pipelined:
        b.set_pipelined(True)
    cap = T // 16 + 4096
    aud = torch.zeros((C, cap), device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), cap, st)
    for _ in range(10): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    out["pipelined" if pipelined else "strict"] = round(e0.elapsed_time(e1) / 20, 4)
    sm = b.smeter_all()
    out["smeter_first3"] = [round(float(v), 2) for v in sm[:3]]
    del b
print(json.dumps({"streams": S, "receivers": C, "ms_per_call": out}))
