import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
w = bench.C4Workload(torch, ca, ctx, 256)

def remake(w, pipelined):
    """drop every batch object the workload holds and make a new one (what C4Workload.set_mode did on every switch until
    round 6)"""
    import gc
    if w.b is not None:
        w.b.flush(w.stream); torch.cuda.synchronize()
    w.b = None; w.kept = {}; w.mode = None
    gc.collect()
    w.set_mode(pipelined)
out = []
def t():
    return round(bench.gpu_ms(torch, w.step, 8, 20), 3)
remake(w, False); out.append(("strict0", t()))
for i in range(4):                      # new strict objects, nothing pipelined in between
    remake(w, False); out.append(("strict_again%d" % i, t()))
for i in range(4):                      # a pipelined object in between each
    remake(w, True); a = t(); remake(w, False); out.append(("pipe%d" % i, a)); out.append(("strict_after%d" % i, t()))
print(json.dumps(out))
