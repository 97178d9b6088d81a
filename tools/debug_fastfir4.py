import sys, numpy as np
sys.path.insert(0, ".")
import cutesdr_amd as ca
np.set_printoptions(linewidth=220, precision=3, suppress=True)
n = 8192; L = n // 2
b = ca.FastFirBatch(1, n); b.setup(-5000, 5000, 0, 62500.0); H = b.response(0)
rng = np.random.default_rng(0)
x = (rng.standard_normal((1, L)) + 1j * rng.standard_normal((1, L))).astype(np.complex64)
y = b.process(x, blocks_per_wg=1)[0]
seg = np.concatenate([np.zeros(L), x[0]])
full = np.fft.fft(n * np.fft.ifft(seg) * H)       # all n outputs of the circular convolution
ref = full[L:]
err = np.abs(y - ref)
print("bad frac", (err > 1e-3).mean())
e2 = err.reshape(-1, 1024)
print("err by output row (n1-HALF):", e2.max(axis=1))
cols = e2.max(axis=0).reshape(-1, 4)
print("err by column mod 4:", cols.max(axis=0))
for i in (0, 1, 2, 3, 4, 5, 1024, 1025, 2048, 4095):
    j = np.argmin(np.abs(full - y[i]))
    print("y[%d] = %s  ref=%s   matches full[%d] (diff %.2g) -> n1=%d col=%d" % (i, y[i], ref[i], j, abs(full[j] - y[i]), j // 1024, j % 1024))
