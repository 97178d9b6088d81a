"""GPU parity of the batched overlap-save kernel (K1) against the fp64 oracle, through the
C ABI.  Tolerance: |err| <= 2e-5 * max|x| per sample (K1 <= 2e-5 * ||x||inf),
sample counts exact."""
import numpy as np
import pytest
from util_signals import tones_plus_noise

pytestmark = pytest.mark.gpu
TOL = 2e-5


def oracle_filter(oracle, n, x, setup):
    ff = oracle.CFastFIR(n)
    assert ff.SetupParameters(*setup) == 1
    return ff.ProcessData(x)


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
def test_batch_matches_oracle_shared_filter(oracle, n):
    import cutesdr_amd as ca
    C, hops, fs = 3, 5, 62500.0
    T = hops * (n // 2)
    x = np.stack([tones_plus_noise(c, T, fs, [1000.0, -3000.0, 20000.0]) for c in range(C)])
    b = ca.FastFirBatch(C, n)
    assert b.setup(-5000, 5000, 0, fs) == 1
    assert b.setup(-5000, 5000, 0, fs) == 0          # unchanged -> early out like the reference
    y = b.process(x)
    assert y.shape == (C, T)
    for c in range(C):
        ref = oracle_filter(oracle, n, x[c], (-5000, 5000, 0, fs))
        assert len(ref) == T
        err = np.abs(y[c] - ref).max()
        assert err <= TOL * np.abs(x[c]).max(), (c, err)


@pytest.mark.parametrize("n,C,hops", [(2048, 16, 37), (2048, 24, 5), (4096, 16, 21), (4096, 24, 5), (8192, 8, 9), (16384, 8, 5)])
def test_batch_channel_counts_that_are_multiples_of_eight(oracle, n, C, hops):
    """With eight channels or a multiple, the runs of a channel are laid over the workgroup index so that they share
    an XCD (the other branch of the kernels' index mapping); a prime hop count gives ragged runs."""
    import cutesdr_amd as ca
    fs = 62500.0
    T = hops * (n // 2)
    x = np.stack([tones_plus_noise(40 + c, T, fs, [700.0 * (c % 5 + 1), -4100.0, 15000.0]) for c in range(C)])
    b = ca.FastFirBatch(C, n)
    b.setup(-5000, 5000, 0, fs)
    for c in range(0, C, 3):
        b.setup(200 + 10 * c, 3000 + 10 * c, 0, fs, channel=c)          # some channels with a filter of their own
    y = b.process(x)
    for c in range(C):
        cut = (200 + 10 * c, 3000 + 10 * c, 0, fs) if c % 3 == 0 else (-5000, 5000, 0, fs)
        ref = oracle_filter(oracle, n, x[c], cut)
        assert np.abs(y[c] - ref).max() <= TOL * np.abs(x[c]).max(), c


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
def test_batch_distinct_filters_and_response(oracle, n):
    """(both sizes that have a kernel of their own: 16384 the pipelined one, 2048 the 128-thread one)"""
    import cutesdr_amd as ca
    C, fs = 4, 62500.0
    T = 3 * (n // 2) if n == 16384 else 19 * (n // 2)
    x = np.stack([tones_plus_noise(c, T, fs, [500.0 * (c + 1), -12000.0]) for c in range(C)])
    b = ca.FastFirBatch(C, n)
    b.setup(-5000, 5000, 0, fs)
    cuts = [(-5000, 5000, 0), (100, 2800, 0), (-2800, -100, 0), (-250, 250, 700)]
    for c, (lo, hi, off) in enumerate(cuts):
        assert b.setup(lo, hi, off, fs, channel=c) == 1
    y = b.process(x)
    for c, (lo, hi, off) in enumerate(cuts):
        ff = oracle.CFastFIR(n)
        ff.SetupParameters(lo, hi, off, fs)
        np.testing.assert_allclose(b.response(c), ff.coef(), atol=1e-12)
        ref = ff.ProcessData(x[c])
        assert np.abs(y[c] - ref).max() <= TOL * np.abs(x[c]).max()


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
def test_batch_streaming_state_and_run_lengths(oracle, n):
    import cutesdr_amd as ca
    C, fs = 2, 62500.0
    L = n // 2
    T = 7 * L
    x = np.stack([tones_plus_noise(10 + c, T, fs, [2000.0, 9000.0]) for c in range(C)])
    refs = [oracle_filter(oracle, n, x[c], (-5000, 5000, 0, fs)) for c in range(C)]
    for bpw in (1, 2, 3, 7, 0):
        b = ca.FastFirBatch(C, n)
        b.setup(-5000, 5000, 0, fs)
        y = b.process(x, blocks_per_wg=bpw)
        for c in range(C):
            assert np.abs(y[c] - refs[c]).max() <= TOL * np.abs(x[c]).max(), bpw
    # three calls (3+1+3 hops) continue the stream exactly like one long call
    b = ca.FastFirBatch(C, n)
    b.setup(-5000, 5000, 0, fs)
    parts = [b.process(x[:, :3 * L]), b.process(x[:, 3 * L:4 * L]), b.process(x[:, 4 * L:])]
    y = np.concatenate(parts, axis=1)
    for c in range(C):
        assert np.abs(y[c] - refs[c]).max() <= TOL * np.abs(x[c]).max()
    b.reset()
    y2 = b.process(x[:, :3 * L])
    np.testing.assert_array_equal(y2, parts[0])


def test_batch_rejects_bad_arguments():
    import cutesdr_amd as ca
    from cutesdr_amd._capi import CsdrError
    b = ca.FastFirBatch(2, 2048)
    b.setup(-5000, 5000, 0, 62500.0)
    with pytest.raises(CsdrError):
        b.process(np.zeros((2, 1000), dtype=np.complex64))       # not a multiple of the hop
    with pytest.raises(CsdrError):
        b.setup(5000, -5000, 0, 62500.0)                          # reference: parameter error
    with pytest.raises(CsdrError):
        ca.FastFirBatch(2, 1000)


@pytest.mark.parametrize("n,chunk", [(2048, 240), (2048, 256), (16384, 19968), (4096, 1)])
def test_host_cfastfir_ragged_calls(oracle, n, chunk):
    """CFastFIR drop-in semantics: arbitrary InLength per call, outputs appear hop by hop."""
    import cutesdr_amd as ca
    fs = 62500.0
    total = 3 * n + 777 if chunk > 1 else 3 * (n // 2) + 5
    x = tones_plus_noise(7, total, fs, [1500.0, -8000.0])
    ff = ca.CFastFIR(n)
    ref = oracle.CFastFIR(n)
    assert ff.SetupParameters(100, 2800, 0, fs) == 1
    ref.SetupParameters(100, 2800, 0, fs)
    got, want = [], []
    for i in range(0, total, chunk):
        g = ff.ProcessData(x[i:i + chunk])
        w = ref.ProcessData(x[i:i + chunk])
        assert len(g) == len(w)
        got.append(g); want.append(w)
    got, want = np.concatenate(got), np.concatenate(want)
    assert len(got) == (total // (n // 2)) * (n // 2)
    assert np.abs(got - want).max() <= TOL * np.abs(x).max()


def test_full_size_properties():
    """BASELINE C3 shape (256 ch x 2^19, N=16384): size-independent properties --
    an impulse returns the filter taps (delayed), and the map is linear."""
    import cutesdr_amd as ca
    n, C, fs = 16384, 256, 62500.0
    L = n // 2
    T = 1 << 19
    b = ca.FastFirBatch(C, n)
    b.setup(-5000, 5000, 0, fs)
    H = b.response(0)
    taps = np.fft.fft(H)[: L + 1]                 # inverse of the +exponent forward transform
    x = np.zeros((C, T), dtype=np.complex64)
    pos = [(c * 7919) % (T - 2 * n) for c in range(C)]
    for c in range(C):
        x[c, pos[c]] = 1000.0 * (1 + (c % 5)) * np.exp(1j * c)
    y = b.process(x)
    for c in range(0, C, 17):
        seg = y[c, pos[c]: pos[c] + L + 1]
        want = x[c, pos[c]] * taps
        assert np.abs(seg - want).max() <= 1e-5 * np.abs(x[c, pos[c]]) * np.abs(taps).max() * 10
    # linearity: F(a*x1 + x2) = a*F(x1) + F(x2) on two channels' worth of noise
    rng = np.random.default_rng(99)
    x1 = (rng.standard_normal((C, 4 * L)) + 1j * rng.standard_normal((C, 4 * L))).astype(np.complex64)
    x2 = (rng.standard_normal((C, 4 * L)) + 1j * rng.standard_normal((C, 4 * L))).astype(np.complex64)
    b.reset(); y1 = b.process(x1)
    b.reset(); y2 = b.process(x2)
    b.reset(); y3 = b.process((2.0 * x1 + x2).astype(np.complex64))
    assert np.abs(y3 - (2.0 * y1 + y2)).max() <= 2e-5 * 6


def test_generic_kernel_at_16384_in_child_process():
    """N = 16384 normally runs the software-pipelined build (fastfir2_kernels.hip), which walks its blocks
    in pairs; launches with an odd block count fall back to the generic kernel (fastfir_kernels.hip).
    CSDR_FASTFIR_VARIANT=0 forces the generic kernel for every launch: it stays under the same parity cases.
    The switch is read when a batch object is created, so they run in a child interpreter."""
    import os, subprocess, sys
    env = dict(os.environ, CSDR_FASTFIR_VARIANT="0")
    here = os.path.dirname(__file__)
    code = ("import sys; sys.path.insert(0, %r); import pytest; "
            "sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', '-k', '16384', %r]))"
            % (here, os.path.join(here, "test_fastfir_gpu.py") + "::test_batch_matches_oracle_shared_filter"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_pipelined_and_generic_kernels_agree(oracle, n):
    """The pipelined build (FMA-form butterflies; round 5: also at 8192 and 4096 points, outer pass of radix 8 / 4 over
    four / eight columns per thread, 4096 points as two blocks per workgroup) and the generic kernel against the oracle
    and against each other, for even and odd block counts (the pipelined kernel walks pairs of blocks, then a single
    trailing one; at 4096 points an odd number of RUNS leaves a workgroup's second half idle) and across two calls
    (overlap carried)."""
    import ctypes as C
    import cutesdr_amd as ca
    L = ca.lib()
    L.csdr__fastfir_set_variant.restype = C.c_int
    L.csdr__fastfir_set_variant.argtypes = [C.c_void_p, C.c_int]
    Cn = 3
    rng = np.random.default_rng(5)
    for nb in (1, 2, 5, 8):
        x = (3000.0 * (rng.standard_normal((Cn, nb * n // 2)) + 1j * rng.standard_normal((Cn, nb * n // 2)))).astype(np.complex64)
        outs = []
        for v in (0, 2):
            b = ca.FastFirBatch(Cn, n)
            b.setup(-5000, 5000, 0, 62500.0)
            assert L.csdr__fastfir_set_variant(b.h, v) == 0
            outs.append(np.concatenate([b.process(x), b.process(x[:, ::-1].copy())], axis=1))
        tol = TOL * np.abs(x).max()
        assert np.abs(outs[0] - outs[1]).max() <= tol, nb
        for c in range(Cn):
            ff = oracle.CFastFIR(n); ff.SetupParameters(-5000, 5000, 0, 62500.0)
            ref = np.concatenate([ff.ProcessData(x[c].astype(np.complex128)), ff.ProcessData(x[c, ::-1].astype(np.complex128))])
            for o in outs:
                assert np.abs(o[c] - ref).max() <= tol, (nb, c)


def test_pipelined_kernel_words_do_not_depend_on_chunking_or_run_length():
    """One kernel serves every launch at N = 16384, so the output WORDS of a stream are the same however it is cut
    into calls (odd and even hop counts) and into runs of blocks per workgroup (csdr_demod_batch_process relies on it:
    the number of hops per call varies there)."""
    import cutesdr_amd as ca
    n, Cn, L = 16384, 3, 8192
    rng = np.random.default_rng(11)
    x = (3000.0 * (rng.standard_normal((Cn, 9 * L)) + 1j * rng.standard_normal((Cn, 9 * L)))).astype(np.complex64)

    def run(cuts, bpw):
        b = ca.FastFirBatch(Cn, n)
        b.setup(-4000, 6000, 0, 62500.0)
        parts, at = [], 0
        for c in cuts:
            parts.append(b.process(x[:, at * L:(at + c) * L].copy(), blocks_per_wg=bpw))
            at += c
        assert at == 9
        return np.concatenate(parts, axis=1)

    ref = run([9], 0)
    for cuts, bpw in (([9], 1), ([9], 2), ([9], 4), ([9], 5), ([1] * 9, 0), ([2, 3, 4], 0), ([5, 4], 3), ([1, 8], 2)):
        got = run(cuts, bpw)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (cuts, bpw)
