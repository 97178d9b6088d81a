# one FastFIR size at the C3 shape (256 ch x 2^19), 20 launches: a target for tools/pmc_any.sh
#   tools/pmc_any.sh fastfir python3 tools/bench_ff_size.py 4096
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C, T = 256, 1 << 19
dev = torch.device("cuda", 0)
x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
ff = ca.FastFirBatch(C, n)
ff.setup(-5000, 5000, 0, 62500.0)
bpw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for _ in range(5): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st, bpw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ff.process_ptr(x.data_ptr(), T, T, y.data_ptr(), T, st, bpw)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"n": n, "ms": round(e0.elapsed_time(e1) / 20, 4)}))
