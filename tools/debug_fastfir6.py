import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
import cutesdr_amd as ca
from cutesdr_amd._capi import lib
L_ = lib()
L_.csdr__dbg_fastfir_stage.restype = C.c_int
L_.csdr__dbg_fastfir_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
n = 8192; L = n // 2
for twin in (0, 9):
    b = ca.FastFirBatch(1, n); b.setup(-5000, 5000, 0, 62500.0); H = b.response(0)
    dbg = ca.DeviceBuffer(n * 8)
    L_.csdr__dbg_fastfir_stage(b.h, twin, C.c_void_p(dbg.ptr))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, 2 * L)) + 1j * rng.standard_normal((1, 2 * L))).astype(np.complex64)
    y = b.process(x, blocks_per_wg=2)[0]
    xx = np.concatenate([np.zeros(L), x[0]])
    for blk in range(2):
        seg = xx[blk * L: blk * L + n]
        ref = np.fft.fft(n * np.fft.ifft(seg) * H)[L:]
        err = np.abs(y[blk * L:(blk + 1) * L] - ref)
        print("twin=%d blk=%d bad=%d maxerr=%.3g" % (twin, blk, (err > 1e-4).sum(), err.max()))
