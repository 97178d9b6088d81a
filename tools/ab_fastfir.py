#!/usr/bin/env python3
"""A/B timing of the K1 builds on the C3 workload in ONE process, interleaved rounds (the methodology
rule for perf deltas): variant 0 = fastfir_os_kernel (generic), 2 = software-pipelined build.
Also checks that every variant reproduces variant 0's output.  usage: ab_fastfir.py [variants] [rounds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutesdr_amd as ca
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,2").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Cn, T = 256, 1 << 19
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((Cn, T, 2), generator=g, device=dev, dtype=torch.float32) * 3276.7
st = torch.cuda.current_stream().cuda_stream
L = ca.lib()
L.csdr__fastfir_set_variant.restype = C.c_int
L.csdr__fastfir_set_variant.argtypes = [C.c_void_p, C.c_int]
objs, outs = {}, {}
for v in variants:
    ff = ca.FastFirBatch(Cn, 16384)
    ff.setup(-5000, 5000, 0, 62500.0)
    assert L.csdr__fastfir_set_variant(ff.h, v) == 0
    objs[v] = ff
    outs[v] = torch.empty_like(x)
res = {"variants": variants}
for v in variants:                       # same stream history for everyone: two calls from reset
    for _ in range(2):
        objs[v].process_ptr(x.data_ptr(), T, T, outs[v].data_ptr(), T, st)
torch.cuda.synchronize()
ref = outs[variants[0]]
for v in variants[1:]:
    res["maxdiff_%d" % v] = float((outs[v] - ref).abs().max())
res["ref_absmax"] = float(ref.abs().max())
# steady clocks first
for _ in range(100):
    for v in variants:
        objs[v].process_ptr(x.data_ptr(), T, T, outs[v].data_ptr(), T, st)
torch.cuda.synchronize()
times = {v: [] for v in variants}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        e0.record()
        for _ in range(40):
            objs[v].process_ptr(x.data_ptr(), T, T, outs[v].data_ptr(), T, st)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 40)
for v in variants:
    ts = sorted(times[v])
    res["ms_%d" % v] = {"median": round(ts[len(ts) // 2], 4), "min": round(ts[0], 4), "max": round(ts[-1], 4),
                        "frac_of_8TBps": round(Cn * T * 16 / (ts[len(ts) // 2] * 1e-3) / 8e12, 4)}
print(json.dumps(res))
