import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
C = 256
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
out = {}
for n in ([int(v) for v in os.environ.get("K3_SIZES", "4096,16384").split(",")]):
    for frames in ([int(v) for v in os.environ.get("K3_FRAMES", "8,32,128").split(",")]):
        T = n * frames
        x = torch.randn((C, T, 2), device=dev, dtype=torch.float32) * 3276.7
        fb = ca.FftBatch(C); fb.set_params(n, False, 0.0, 2e6); fb.set_ave(1)
        f = lambda: fb.put_display_ptr(x.data_ptr(), T, frames, st)
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        out["%d/%d" % (n, frames)] = {"ms": round(ms, 4), "us_per_frame_per_ch": round(ms * 1e3 / frames, 2)}
        del x, fb
print(json.dumps(out))
