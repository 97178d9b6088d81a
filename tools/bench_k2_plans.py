#!/usr/bin/env python3
"""K2 per decimator plan, as the chain runs it: one plan group of C receivers x 2^21 samples at 2 MSPS
(a third of BASELINE config C4's per-GPU share by default).  Prints one JSON line: ms and algorithmic GB/s
(8 B in + 8 B out per decimated sample) for the FM / SSB / AM / CW plans."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C = int(sys.argv[1]) if len(sys.argv) > 1 else 86
T = 1 << 21
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((C, T, 2), generator=g, device=dev, dtype=torch.float32) * 3276.7
y = torch.empty((C, T // 16, 2), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
out = {"channels": C, "samples": T}
for name, bw in (("fm_11_11_15_19_31", 15000.0), ("ssb_11_11_15_23_51", 20000.0), ("am_11_11_11_15_23_51", 10000.0), ("cw_3_3_11_11_11_11_15", 1000.0)):
    dc = ca.DownConvertBatch(C)
    rate = dc.set_data_rate(2e6, bw)
    for c in range(C):
        dc.set_frequency(-100e3 - 500.0 * c, channel=c)
    f = lambda: dc.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T // 16, st)
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    dec = 2e6 / rate
    out[name] = {"ms": round(ms, 4), "alg_GBps": round(C * T * (8 + 8 / dec) / ms / 1e6, 1), "out_rate": rate}
print(json.dumps(out))
