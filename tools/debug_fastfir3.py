import sys, numpy as np
sys.path.insert(0, ".")
import cutesdr_amd as ca
np.set_printoptions(linewidth=200, precision=3, suppress=True)
for n in (8192,):
    L = n // 2
    for C, hops, bpw in ((1, 1, 1), (1, 2, 1), (1, 2, 2), (3, 5, 1), (3, 5, 5), (2, 3, 0)):
        b = ca.FastFirBatch(C, n)
        b.setup(-5000, 5000, 0, 62500.0)
        H = b.response(0)
        rng = np.random.default_rng(0)
        x = (rng.standard_normal((C, hops * L)) + 1j * rng.standard_normal((C, hops * L))).astype(np.complex64)
        y = b.process(x, blocks_per_wg=bpw)
        errs = np.zeros((C, hops))
        for c in range(C):
            xx = np.concatenate([np.zeros(L), x[c]])
            for blk in range(hops):
                seg = xx[blk * L: blk * L + n]
                ref = np.fft.fft(n * np.fft.ifft(seg) * H)[L:]
                errs[c, blk] = np.abs(y[c, blk * L:(blk + 1) * L] - ref).max()
        print("C=%d hops=%d bpw=%d  err per (ch,block):\n" % (C, hops, bpw), errs)
