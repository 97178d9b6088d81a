#!/usr/bin/env python3
"""BASELINE config 2 on the device: ONE channel, 2 MSPS IQ resident in HBM -> CDownConvert ->
16384-point CFastFIR -> FM demodulator + AGC (csdr_demod_batch with one channel).  Secondary
measurement: a single receiver cannot fill the chip, this is its latency-bound rate."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch
import cutesdr_amd as ca
T = 1 << 24
dev = torch.device("cuda", 0)
t = torch.arange(T, device=dev, dtype=torch.float64) / 2e6
ph = 2 * torch.pi * 100e3 * t + 3.0 * torch.sin(2 * torch.pi * 1000.0 * t)
x = torch.stack([(3276.7 * torch.cos(ph)).float(), (3276.7 * torch.sin(ph)).float()], dim=-1).reshape(1, T, 2).contiguous()
x += torch.randn_like(x) * 10.0
b = ca.DemodBatch(1, 16384); b.set_input_rate(2e6); b.set_demod(0, ca.DEMOD_FM, ca.fm_defaults()); b.commit(); b.set_freq(0, -100e3)
aud = torch.empty((1, T // 32 + 16384), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
f = lambda: b.process_ptr(x.data_ptr(), T, T, aud.data_ptr(), aud.shape[1], st)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"config": "C2: 1 channel, 2 MSPS -> downconvert -> 16384-pt FastFIR -> FM + AGC", "samples": T,
                  "ms": round(ms, 3), "input_MSps": round(T / ms / 1e3, 1), "x_realtime_at_2MSPS": round(T / ms / 1e3 / 2.0, 1)}))
