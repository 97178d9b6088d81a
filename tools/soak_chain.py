#!/usr/bin/env python3
"""Soak run for the whole batch chain: fresh objects, the same input, repeated; audio and S-meter must
be bit-identical from run to run (scans, guessed solves and barriers are deterministic)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import cutesdr_amd as ca
import test_postchain_gpu as T
C, N, REPS = 96, 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((C, 2 * N, 2), generator=g, device=dev, dtype=torch.float32) * 30.0
t = torch.arange(2 * N, device=dev, dtype=torch.float64) / 2e6
for c in range(C):
    ph = 2 * torch.pi * (100e3 + 300.0 * c) * t + (3.0 * torch.sin(2 * torch.pi * 1000.0 * t) if c % 3 == 1 else 0.0)
    amp = 3276.7 * (1.0 + 0.5 * torch.sin(2 * torch.pi * 800.0 * t)) if c % 3 == 0 else 3276.7
    x[c, :, 0] += (amp * torch.cos(ph)).float(); x[c, :, 1] += (amp * torch.sin(ph)).float()
names = ["AM", "FM", "USB", "SAM", "LSB", "CWU"]
st = torch.cuda.current_stream().cuda_stream
ref, bad = None, 0
for rep in range(REPS):
    b = ca.DemodBatch(C, 2048); b.set_input_rate(2e6)
    for c in range(C):
        m, kw = T.MODES[names[c % 6]]
        b.set_demod(c, m, T.info(ca, **kw))
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3 - 300.0 * c)
    aud = torch.zeros((C, 2 * (N // 16 + 4096)), device=dev, dtype=torch.float32)
    half = aud.shape[1] // 2
    for call in range(2):
        xin = x[:, call * N:(call + 1) * N]
        b.process_ptr(xin.data_ptr(), 2 * N, N, aud[:, call * half:].data_ptr(), aud.shape[1], st)
    torch.cuda.synchronize()
    sm = [b.smeter_ave(c) for c in (0, 1, C - 1)]
    cur = (aud.clone(), sm)
    if ref is None: ref = cur
    else: bad += int((cur[0].view(torch.int32) != ref[0].view(torch.int32)).any().item() or cur[1] != ref[1])
    del b
print(json.dumps({"reps": REPS, "mismatching_runs": bad, "audio_abs_sum": float(ref[0].abs().sum().item())}))
