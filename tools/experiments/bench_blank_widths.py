import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
w = bench.C4Workload(torch, ca, ctx, 256)
w.set_mode(False)
b, x, T, C = w.b, w.x, w.T, w.C
dev = x.device
PKT = int(os.environ.get("PKT", "1444"))
per = 240 if PKT == 1444 else 256
npk = (T // per) // 8 * 8
Tp = npk * per
pk = torch.zeros((C, npk, PKT), device=dev, dtype=torch.uint8)
for c0 in range(0, C, 32):
    if PKT == 1444:
        v = torch.round(x[c0:c0 + 32, :Tp].reshape(-1, npk, 480) * 256.0).clamp(-(1 << 23), (1 << 23) - 1).to(torch.int32)
        body = torch.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], dim=-1).to(torch.uint8).reshape(-1, npk, 1440)
    else:
        v = torch.round(x[c0:c0 + 32, :Tp].reshape(-1, npk, 512)).clamp(-32768, 32767).to(torch.int32)
        body = torch.stack([v & 255, (v >> 8) & 255], dim=-1).to(torch.uint8).reshape(-1, npk, 1024)
    pk[c0:c0 + 32, :, 4:] = body
    del v, body
aud = w.aud
st = torch.cuda.current_stream().cuda_stream
out = {}
for width in (2.0, 3.0):
    nb = ca.NoiseProcBatch(C); nb.setup(True, 50.0, width, 2e6)
    def run():
        rc = ca.lib().csdr_demod_batch_process_packets(b.h, pk.data_ptr(), npk, PKT, nb.h, aud.data_ptr(), w.cap, st)
        assert rc == 0
    out["width_%g_ms" % width] = round(bench.gpu_ms(torch, run, 10, 30), 3)
def run0():
    rc = ca.lib().csdr_demod_batch_process_packets(b.h, pk.data_ptr(), npk, PKT, None, aud.data_ptr(), w.cap, st)
    assert rc == 0
out["no_blanker_ms"] = round(bench.gpu_ms(torch, run0, 10, 30), 3)
print(json.dumps(out))
