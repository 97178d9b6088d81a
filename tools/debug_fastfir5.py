import sys, numpy as np
sys.path.insert(0, ".")
import cutesdr_amd as ca
np.set_printoptions(linewidth=220, precision=3, suppress=True)
for n in (2048, 4096, 8192, 16384):
    L = n // 2; R0 = n // 1024; G = 32 // R0
    b = ca.FastFirBatch(1, n); b.setup(-5000, 5000, 0, 62500.0); H = b.response(0)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, 2 * L)) + 1j * rng.standard_normal((1, 2 * L))).astype(np.complex64)
    y = b.process(x, blocks_per_wg=2)[0]
    xx = np.concatenate([np.zeros(L), x[0]])
    for blk in range(2):
        seg = xx[blk * L: blk * L + n]
        ref = np.fft.fft(n * np.fft.ifft(seg) * H)[L:]
        err = np.abs(y[blk * L:(blk + 1) * L] - ref)
        bad = np.nonzero(err > 1e-4)[0]
        rows, cols = bad // 1024, bad % 1024
        print("N=%d blk=%d bad=%d rows=%s col%%G=%s threads(t=col//G): %s" % (n, blk, len(bad), sorted(set(rows)), sorted(set(cols % G)), sorted(set(cols // G))[:40]))
        if len(bad):
            i = bad[0]
            print("   y=%s ref=%s  y/ref=%s" % (y[blk * L + i], ref[i], y[blk * L + i] / ref[i]))
