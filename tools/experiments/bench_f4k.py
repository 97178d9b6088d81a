import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C, T = 256, 1 << 19
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
res = {}
for n in (4096,):
    for bpw in (0, 16, 32, 64, 128, 256):
        ff = ca.FastFirBatch(C, n)
        ff.setup(-5000, 5000, 0, 62500.0)
        for _ in range(60): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st, bpw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st, bpw)
        e1.record(); torch.cuda.synchronize()
        res["bpw%d" % bpw] = round(e0.elapsed_time(e1) / 200, 4)
print(json.dumps(res))
