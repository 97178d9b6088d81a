import sys, numpy as np
sys.path.insert(0, ".")
import cutesdr_amd as ca
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for n in (2048, 16384):
    L = n // 2
    b = ca.FastFirBatch(1, n)
    b.setup(-5000, 5000, 0, 62500.0)
    H = b.response(0)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, 2 * L)) + 1j * rng.standard_normal((1, 2 * L))).astype(np.complex64)
    y = b.process(x, blocks_per_wg=1)[0]
    xx = np.concatenate([np.zeros(L), x[0]])
    ref = []
    for blk in range(2):
        seg = xx[blk * L: blk * L + n]
        X = n * np.fft.ifft(seg)
        ref.append(np.fft.fft(X * H)[L:])
    ref = np.concatenate(ref)
    err = np.abs(y - ref)
    print("N", n, "max err", err.max(), "ref max", np.abs(ref).max())
    e2 = err.reshape(2, -1, 1024)        # [block][n1-HALF][n2]
    print(" err by block/n1 row (max over n2):\n", e2.max(axis=2))
    print(" err by n2%32 (max):", e2.max(axis=(0, 1)).reshape(-1, 32).max(axis=0))
    print(" err by n2//32 (max):", e2.max(axis=(0, 1)).reshape(-1, 32).max(axis=1))
    # impulse / allpass style probes
    for pos in (0, 1, 5, 1024, L - 1):
        if pos >= 2 * L: continue
        xi = np.zeros((1, 2 * L), dtype=np.complex64); xi[0, pos] = 1.0
        b.reset(); yi = b.process(xi, blocks_per_wg=1)[0]
        taps = np.fft.fft(H)[: L + 1]
        want = np.zeros(2 * L, dtype=complex); m = min(L + 1, 2 * L - pos); want[pos: pos + m] = taps[:m]
        print("  impulse@%d err %.3g (taps max %.3g)" % (pos, np.abs(yi - want).max(), np.abs(taps).max()))
