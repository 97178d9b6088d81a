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
