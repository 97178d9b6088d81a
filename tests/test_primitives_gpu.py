"""Device-side unit test of the packed-complex primitives (VOP3P op_sel/neg asm forms) and the
register butterflies of csrc/fft_core.hpp against numpy."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bitrev(v, R):
    r, m = 0, 1
    while m < R:
        r = (r << 1) | (v & 1); v >>= 1; m <<= 1
    return r


def test_packed_complex_primitives_and_butterflies():
    from cutesdr_amd._capi import lib, check
    L = lib()
    L.csdr__selftest_fft.restype = C.c_int
    L.csdr__selftest_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(36) + 1j * rng.standard_normal(36)).astype(np.complex64)
    x[1] = np.exp(0.37j)
    out = np.zeros(148, dtype=np.complex64)
    check(L.csdr__selftest_fft(0, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "selftest")
    a, w, u, v = [complex(z) for z in x[:4]]
    want = [a * w, a * np.conj(w), u + 1j * v, u - 1j * v, 1j * (u - v), -1j * (u - v)]
    np.testing.assert_allclose(out[:6], want, rtol=0, atol=2e-6)
    d = x[4:36].astype(np.complex128)
    F = 32 * np.fft.ifft(d)                       # positive-exponent DFT
    got = np.array([out[8 + bitrev(k, 32)] for k in range(32)])
    np.testing.assert_allclose(got, F, atol=2e-5)
    np.testing.assert_allclose(out[40:72], 32 * d, atol=1e-4)      # dit(-1) of dif(+1) = 32 x
    F16 = 16 * np.fft.ifft(d[:16])
    got16 = np.array([out[72 + bitrev(k, 16)] for k in range(16)])
    np.testing.assert_allclose(got16, F16, atol=1e-5)
    np.testing.assert_allclose(out[88:104], np.exp(0.37j * np.arange(16)), atol=3e-6)
    # small radices (outer pass of N = 2048..16384): forward values and forward->inverse round trip
    for R, fo, io in ((8, 104, 112), (4, 120, 124), (2, 128, 130)):
        FR = R * np.fft.ifft(d[:R])
        gotR = np.array([out[fo + bitrev(k, R)] for k in range(R)])
        np.testing.assert_allclose(gotR, FR, atol=1e-5, err_msg="dif%d" % R)
        np.testing.assert_allclose(out[io:io + R], R * d[:R], atol=2e-5, err_msg="dit%d" % R)
    np.testing.assert_allclose(out[132:148], 16 * d[:16], atol=5e-5, err_msg="dit16")


def test_small_accessors_of_the_abi():
    """The entry points nothing else touches: csdr_version, the batch down-converter's NCO frequency (SetFrequency adds the
    CW offset, downconvert.cpp:98-103 -- and SetDataRate re-applies the sum, :169: the offset is added twice, the quirk the
    golden anchors pin), the shard count, and ResetFFT through the batch and the drop-in form (counts back to zero, the
    next frame starts the average again)."""
    import ctypes as C
    import cutesdr_amd as ca
    from util_signals import tones_plus_noise
    L = ca.lib()
    assert L.csdr_version() >= 1
    dc = ca.DownConvertBatch(3)
    dc.set_data_rate(2e6, 1000.0)
    dc.set_cw_offset(700.0, channel=1)
    dc.set_frequency(-100e3)
    L.csdr_downconvert_batch_get_nco_freq.restype = C.c_double
    f = [L.csdr_downconvert_batch_get_nco_freq(dc.h, c) for c in range(3)]
    assert f[0] == f[2] == -100e3 and f[1] == -100e3 + 700.0
    sh = ca.ShardedDemodBatch([0, 0], 6, 2048)
    assert L.csdr_demod_shard_count(sh.h) == 2
    n, fs = 2048, 2e6
    x = tones_plus_noise(3, 3 * n, fs, [200e3])
    fb = ca.FftBatch(1); fb.set_params(n, False, 0.0, fs); fb.set_ave(4)
    fb.put_display(x[None, :])
    assert fb.total_count(0) == 3
    assert L.csdr_fft_batch_reset(fb.h) == 0
    fb.put_display(x[None, :n])
    assert fb.total_count(0) == 1
    g = ca.CFft(); g.SetFFTParams(n, False, 0.0, fs); g.SetFFTAve(4)
    for k in range(3):
        assert g.PutInDisplayFFT(x[k * n:(k + 1) * n]) == k + 1
    g.ResetFFT()
    assert g.PutInDisplayFFT(x[:n]) == 1
    one = ca.CFft(); one.SetFFTParams(n, False, 0.0, fs); one.SetFFTAve(4); one.PutInDisplayFFT(x[:n])
    assert np.array_equal(g.ave_buf(), one.ave_buf()) and np.array_equal(fb.ave_buf(0), one.ave_buf().astype(np.float32))
