"""CPU checks of the oracle's front-end restatements (noise blanker, wire format, DC estimate)
against independent numpy statements of the same arithmetic.  The reference has no vectors for
these (parity unpinned beyond this cross-check)."""
import numpy as np
import pytest


def test_blanker_equals_window_formulation(oracle):
    fs, thresh, width = 2e6, 40.0, 20.0
    rng = np.random.default_rng(2)
    n = 60000
    x = 200.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x[rng.random(n) < 3e-4] += 30000.0
    nb = oracle.CNoiseProc(); nb.SetupBlanker(True, thresh, width, fs)
    got = np.concatenate([nb.ProcessBlanker(x[:777]), nb.ProcessBlanker(x[777:])])
    # noiseproc.cpp:92-102: widths and ratio
    W = max(1, min(4096, int(width * 1e-6 * fs))); M = int(0.005 * fs); D = W // 2
    ratio = 0.005 * thresh * M
    mag = np.maximum(np.abs(x.real), np.abs(x.imag))
    cs = np.concatenate([[0.0], np.cumsum(mag)])
    idx = np.arange(n)
    S = cs[idx + 1] - cs[np.maximum(idx - M, 0)]              # last M+1 magnitudes
    trig = mag * ratio > S
    last = np.maximum.accumulate(np.where(trig, idx, -10**9))
    blank = (idx - last) < W
    delayed = np.concatenate([np.zeros(D + 1), x])[:n]
    want = np.where(blank, 0.0, delayed)
    # a trigger decided within rounding of the threshold may differ between the running sum and cumsum
    margin = np.abs(mag * ratio - S) < 1e-6 * S
    assert not margin.any()
    assert np.array_equal(got, want)
    assert 0 < blank.sum() < n // 2


def test_unpack_against_numpy(oracle):
    rng = np.random.default_rng(4)
    raw = rng.integers(0, 256, (5, 1028), dtype=np.uint8)
    want = raw[:, 4:].reshape(-1).view("<i2").astype(np.float64)
    got = oracle.unpack_packets(raw, 1028)
    assert np.array_equal(got.view(np.float64), want)
    raw = rng.integers(0, 256, (5, 1444), dtype=np.uint8)
    b = raw[:, 4:].reshape(-1, 3).astype(np.int64)
    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v) / 256.0      # 24-bit two's complement on the 16-bit scale
    assert np.array_equal(oracle.unpack_packets(raw, 1444).view(np.float64), v)


def test_spurcal_closed_form(oracle):
    x = np.full(50000, 3.0 - 4.0j)
    dc = oracle.spurcal([1.0, 1.0], x)
    k = (1 - 1e-5) ** 50000
    assert dc[0] == pytest.approx(k * 1.0 + (1 - k) * 3.0, rel=1e-9)
    assert dc[1] == pytest.approx(k * 1.0 + (1 - k) * -4.0, rel=1e-9)
