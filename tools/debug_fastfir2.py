import sys, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import cutesdr_amd as ca
from cutesdr_amd._capi import lib
from emulate_fastfir import block
np.set_printoptions(linewidth=200, precision=3, suppress=True)
L_ = lib()
L_.csdr__dbg_fastfir_stage.restype = C.c_int
L_.csdr__dbg_fastfir_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for n in (16384,):
    L = n // 2
    b = ca.FastFirBatch(1, n)
    b.setup(-5000, 5000, 0, 62500.0)
    H = b.response(0)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((1, L)) + 1j * rng.standard_normal((1, L))).astype(np.complex64)
    blockin = np.concatenate([np.zeros(L), x[0].astype(complex)])
    st = []
    final = block(blockin, H, n, st)
    dbg = ca.DeviceBuffer(n * 8)
    for s in (1, 2, 3, 4):
        L_.csdr__dbg_fastfir_stage(b.h, s, C.c_void_p(dbg.ptr))
        b.reset()
        b.process(x, blocks_per_wg=1)
        got = dbg.download(np.complex64, n)
        want = st[s - 1]
        err = np.abs(got - want)
        print("N %d stage %d: max err %.3g (max val %.3g)  bad frac %.3f" % (n, s, err.max(), np.abs(want).max(), (err > 1e-3 * np.abs(want).max()).mean()))
        if err.max() > 1e-3 * np.abs(want).max():
            bad = np.nonzero(err > 1e-3 * np.abs(want).max())[0]
            print("   first bad idx", bad[:24], " count", len(bad))
            print("   got ", got[bad[:4]], "\n   want", want[bad[:4]])
            # does got match want at some permuted position?
            for i in bad[:4]:
                j = np.argmin(np.abs(want - got[i]))
                print("   got[%d] equals want[%d]? diff %.3g" % (i, j, abs(want[j] - got[i])))
            break
    L_.csdr__dbg_fastfir_stage(b.h, 0, None)
