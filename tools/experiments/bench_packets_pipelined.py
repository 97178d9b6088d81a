"""round 5: the datagram-fed C4 share with the blanker, strict against pipelined mode (the mask kernel of call k + 1 beside
the post-chains of call k)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import cutesdr_amd as ca
import bench
ctx = bench.dist_init()
torch.cuda.set_device(0)
c4 = bench.C4Workload(torch, ca, ctx, 256)
C, T, x = c4.C, c4.T, c4.x
npk = (T // 240) // 8 * 8
pk = bench.datagrams_of(torch, x, npk)
res = {}
for pipelined in (False, True, False, True):
    c4.set_mode(pipelined)
    nb = ca.NoiseProcBatch(C, device=0)
    nb.setup(True, 50.0, 2.0, bench.C4_FS)
    for key, blk in (("plain", None), ("blanker", nb)):
        def run():
            rc = ca.lib().csdr_demod_batch_process_packets(c4.b.h, pk.data_ptr(), npk, 1444, blk.h if blk is not None else None,
                                                           c4.aud.data_ptr(), c4.cap, c4.stream)
            assert rc == 0, ca._capi.last_error()
        ms = bench.gpu_ms(torch, run, 8, 30)
        c4.b.flush(c4.stream); torch.cuda.synchronize()
        res.setdefault(("pipelined" if pipelined else "strict") + "_" + key, []).append(round(ms, 4))
    del nb
print(json.dumps(res))
