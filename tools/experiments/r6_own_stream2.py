import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(False)
out = {}
out["null_1"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
s = torch.cuda.Stream()
null = w.stream
with torch.cuda.stream(s):
    w.stream = s.cuda_stream
    out["own_same_object"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
w.stream = null
out["null_2"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
with torch.cuda.stream(s):
    w.stream = s.cuda_stream
    out["own_again"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
w.stream = null
out["null_3"] = round(bench.gpu_ms(torch, w.step, 8, 30), 3)
print(json.dumps(out))
