"""Seeded differential sweep of the sample-rate stages against the oracle: random AGC settings
(hang on/off, threshold, slope, decay, sample rate), random signal envelopes (steps, fades,
bursts, silence) and random call lengths.  The AGC averagers and the PLLs are solved per tile
under a guess that is verified afterwards; this sweep is there to hit the guesses' failure and
re-entry paths with inputs nobody hand-picked."""
import numpy as np
import pytest
from util_signals import FULL_SCALE

pytestmark = pytest.mark.gpu


def envelope_signal(rng, n, fs):
    t = np.arange(n) / fs
    kind = rng.integers(0, 4)
    if kind == 0:                                            # level steps
        lv = 10 ** rng.uniform(0.5, 4.2, size=8)
        env = lv[(np.arange(n) * 8 // n)]
    elif kind == 1:                                          # slow fade
        env = 10 ** (2.5 + 1.5 * np.sin(2 * np.pi * rng.uniform(0.5, 5.0) * t))
    elif kind == 2:                                          # bursts over a noise floor
        env = np.where(rng.random(n // 512 + 1).repeat(512)[:n] < 0.3, 8000.0, 30.0)
    else:                                                    # speech-like: product of two modulations
        env = 3000.0 * np.abs(np.sin(2 * np.pi * 3.1 * t) * np.sin(2 * np.pi * 41.0 * t)) + 10.0
    f = rng.uniform(-2000.0, 2000.0)
    x = env * np.exp(2j * np.pi * f * t) + rng.uniform(1.0, 20.0) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return np.clip(x.real, -32000, 32000) + 1j * np.clip(x.imag, -32000, 32000)


@pytest.mark.parametrize("seed", range(12))
def test_agc_random_settings_and_envelopes(oracle, seed):
    import cutesdr_amd as ca
    rng = np.random.default_rng(1000 + seed)
    fs = float(rng.choice([15625.0, 31250.0, 62500.0, 78125.0]))
    hang = bool(rng.integers(0, 2))
    thresh, slope, decay = int(rng.integers(-120, -10)), int(rng.integers(0, 11)), int(rng.integers(20, 2000))
    g, r = ca.CAgc(), oracle.CAgc()
    for o in (g, r):
        o.SetParameters(True, hang, thresh, 30, slope, decay, fs)
    n = 40000
    x = envelope_signal(rng, n, fs)
    pos = 0
    while pos < n:
        m = int(rng.choice([1, 37, 512, 1024, 1500, 4096, 7000]))
        part = x[pos:pos + m]
        got, want = g.ProcessData(part), r.ProcessData(part)
        assert np.abs(got - want).max() <= 1e-3 * FULL_SCALE, (seed, pos, m, hang, thresh, slope, decay, fs)
        pos += m


@pytest.mark.parametrize("seed", range(8))
def test_fm_sam_random_offsets_and_levels(oracle, seed):
    """Carriers that drift, jump and fade: the PLL tiles alternate between the solved and the walked path."""
    import cutesdr_amd as ca
    rng = np.random.default_rng(2000 + seed)
    L = 1024
    for kind in ("fm", "sam"):
        fs = 62500.0 if kind == "fm" else 31250.0
        lim = 6000.0 if kind == "fm" else 1000.0
        n = 24 * L
        t = np.arange(n) / fs
        # piecewise frequency: inside the clamp, near its edge, outside, back
        f = np.repeat(rng.choice([0.2, 0.9, 1.02, 1.6, -0.5, -0.97, -1.3], size=6) * lim, n // 6)
        f = f + 0.05 * lim * np.sin(2 * np.pi * 2.0 * t)
        ph = 2 * np.pi * np.cumsum(f) / fs
        amp = 6000.0 * (1.0 + 0.4 * np.sin(2 * np.pi * 300.0 * t)) if kind == "sam" else 6000.0
        x = amp * np.exp(1j * ph) + 3.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        stereo = bool(rng.integers(0, 2))
        if kind == "fm":
            g, r = ca.CFmDemod(fs), oracle.CFmDemod(fs)
            g.SetSquelch(50); r.SetSquelch(50)
            run = lambda o, p: o.ProcessData(p, 5000.0, stereo)
        else:
            g, r = ca.CSamDemod(fs), oracle.CSamDemod(fs)
            run = lambda o, p: o.ProcessData(p, stereo)
        bad = 0
        for i in range(n // L):
            got, want = run(g, x[i * L:(i + 1) * L]), run(r, x[i * L:(i + 1) * L])
            if kind == "fm":
                assert g.squelched() == r.squelched(), (seed, i)
            seg_pos = i % 4                                   # a frequency jump every 4 hops: compare from the 2nd hop after it
            if seg_pos >= 2 and np.abs(got - want).max() > 1e-3 * FULL_SCALE:
                bad += 1
        assert bad == 0, (seed, kind, stereo)
