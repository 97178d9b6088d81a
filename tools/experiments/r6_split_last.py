#!/usr/bin/env python3
"""GPU box, round 6: the strict C4 step with the LAST plan group's call run as two half calls (CSDR_CHAIN_SPLIT_LAST=1).
  dump OUT.npy   three steps of the C4 share, the audio rows and out counts of the last one -> OUT.npy
  compare A B    largest difference between two dumps, per receiver kind"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "dump":
    import torch
    import cutesdr_amd as ca
    import bench
    ctx = bench.dist_init()
    torch.cuda.set_device(0)
    w = bench.C4Workload(torch, ca, ctx, 256)
    w.set_mode(False)
    outs = []
    for _ in range(3):
        w.step()
        torch.cuda.synchronize()
        outs.append(w.aud.cpu().numpy().copy())
    counts = np.array([w.b.out_count(c) for c in range(w.C)])
    np.save(sys.argv[2], np.stack(outs))
    np.save(sys.argv[2] + ".counts.npy", counts)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ca_, cb = np.load(sys.argv[2] + ".counts.npy"), np.load(sys.argv[3] + ".counts.npy")
    print("counts equal:", bool((ca_ == cb).all()), "min/max", ca_.min(), ca_.max())
    for kind, name in ((0, "AM"), (1, "FM"), (2, "USB")):
        rows = np.arange(kind, a.shape[1], 3)
        d = np.abs(a[:, rows].astype(np.float64) - b[:, rows]).max(axis=(1, 2))
        print(name, "max |a| %.3g" % np.abs(a[:, rows]).max(), "max diff per step", d, "identical" if not d.any() else "")
