#!/usr/bin/env python3
"""What the box's HBM actually delivers to ordinary kernels (torch): read-only reduction, copy, scale.
Context for the roofline fractions in DESIGN.md (the 8 TB/s peak is the spec figure)."""
import json, torch
dev = torch.device("cuda", 0)
n = 1 << 30                                     # 4 GiB of fp32
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def t(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
ms = t(lambda: torch.sum(x)); out["read_sum_TBps"] = round(4 * n / ms / 1e9, 2)
ms = t(lambda: y.copy_(x)); out["copy_TBps"] = round(8 * n / ms / 1e9, 2)
ms = t(lambda: torch.mul(x, 2.0, out=y)); out["scale_TBps"] = round(8 * n / ms / 1e9, 2)
ms = t(lambda: y.zero_()); out["write_TBps"] = round(4 * n / ms / 1e9, 2)
print(json.dumps(out))
