#!/usr/bin/env python3
"""Soak run for the overlap-save kernels: the same input through a reset filter, many times at every
size and in both variants, must give bit-identical outputs (the wide-store hazard of DESIGN.md K1 was
intermittent).  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
C, T, REPS = 256, 1 << 19, int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
st = torch.cuda.current_stream().cuda_stream
res = {}
for n in (2048, 4096, 8192, 16384):
    ff = ca.FastFirBatch(C, n); ff.setup(-5000, 5000, 0, 62500.0)
    y0 = torch.empty_like(x); y = torch.empty_like(x)
    ff.reset(); ff.process_ptr(x.data_ptr(), T, T, y0.data_ptr(), T, st); torch.cuda.synchronize()
    bad = 0
    for _ in range(REPS):
        ff.reset(); ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st); torch.cuda.synchronize()
        bad += int((y.view(torch.int32) != y0.view(torch.int32)).any().item())
    res[str(n)] = {"reps": REPS, "mismatching_runs": bad}
print(json.dumps(res))
