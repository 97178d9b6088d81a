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
        assert np.abs(got - want).max() <= 2e-5 * FULL_SCALE, (seed, pos, m, hang, thresh, slope, decay, fs,
                                                               np.abs(got - want).max() / FULL_SCALE)
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
        worst = 0.0
        for i in range(n // L):
            got, want = run(g, x[i * L:(i + 1) * L]), run(r, x[i * L:(i + 1) * L])
            if kind == "fm":
                assert g.squelched() == r.squelched(), (seed, i)
            seg_pos = i % 4                                   # a frequency jump every 4 hops: compare from the 2nd hop after it
            if seg_pos >= 2:
                worst = max(worst, np.abs(got - want).max())
        assert worst <= 2e-5 * FULL_SCALE, (seed, kind, stereo, worst / FULL_SCALE)


@pytest.mark.parametrize("seed", range(8))
def test_downconvert_random_rates_and_calls(oracle, seed):
    """Random input rate / bandwidth (hence decimator chain), NCO frequency, CW offset and call
    lengths (multiples of the decimation, large enough for the reference's in-place stages:
    SURVEY App. A.3), several calls per stream."""
    import cutesdr_amd as ca
    rng = np.random.default_rng(3000 + seed)
    in_rate = float(rng.choice([250e3, 1e6, 2e6, 5e6, 10e6]))
    bw = float(rng.choice([500.0, 2000.0, 6000.0, 15000.0, 40000.0, 150000.0]))
    freq = float(rng.uniform(-0.4, 0.4) * in_rate)
    cw = float(rng.choice([0.0, 700.0, -650.0]))
    g, r = ca.CDownConvert(), oracle.CDownConvert()
    for o in (g, r):
        o.SetCwOffset(cw)
    assert g.SetDataRate(in_rate, bw) == r.SetDataRate(in_rate, bw)
    assert g.stages() == r.stages()
    for o in (g, r):
        o.SetFrequency(freq)
    assert g.nco_freq() == r.nco_freq()
    ns = len(g.stages())
    unit = 1 << ns
    tol = 1e-5 * FULL_SCALE
    for call in range(4):
        n = int(rng.integers(40, 200)) * unit * 8            # every stage keeps more input than taps
        n = min(n, 32768 // unit * unit)                     # the reference's half-band scratch holds 32768 samples
        t = np.arange(n) + call * 100000
        x = 8000.0 * np.exp(2j * np.pi * (-(freq + cw) + 0.3 * min(bw, in_rate / 8)) * t / in_rate) \
            + 100.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        got, want = g.ProcessData(x), r.ProcessData(x)
        assert len(got) == len(want) == n >> ns, (seed, call)
        assert np.abs(got - want).max() <= tol, (seed, call, in_rate, bw, g.stages())


@pytest.mark.parametrize("seed", range(6))
def test_fastfir_random_filters_and_calls(oracle, seed):
    import cutesdr_amd as ca
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([2048, 4096, 8192, 16384]))
    fs = float(rng.choice([15625.0, 62500.0, 78125.0]))
    lo = float(rng.uniform(-0.45, 0.2) * fs)
    hi = float(lo + rng.uniform(0.02, 0.25) * fs)
    off = float(rng.choice([0.0, 700.0]))
    g, r = ca.CFastFIR(n), oracle.CFastFIR(n)
    g.SetupParameters(lo, hi, off, fs); r.SetupParameters(lo, hi, off, fs)
    total = 0
    for call in range(6):
        m = int(rng.choice([1, 240, 777, n // 2, n // 2 + 1, 3 * n + 5]))
        x = 5000.0 * (rng.standard_normal(m) + 1j * rng.standard_normal(m))
        got, want = g.ProcessData(x), r.ProcessData(x)
        assert len(got) == len(want), (seed, call)
        if len(want):
            assert np.abs(got - want).max() <= 2e-5 * 5000.0 * 5, (seed, call, n)   # 5 sigma peaks of the noise
        total += m


def test_one_wave_per_channel_path_in_child_process():
    """Launches of more than 1024 channels use one wave per channel (postchain_kernel<1>); the tests
    above run with four.  CSDR_POSTCHAIN_WAVES=1 forces the one-wave kernel for the same sweep."""
    import os, subprocess, sys
    env = dict(os.environ, CSDR_POSTCHAIN_WAVES="1")
    here = os.path.dirname(__file__)
    code = ("import sys; sys.path.insert(0, %r); import pytest; "
            "sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', '-k', 'agc_random or fm_sam_random', %r]))"
            % (here, os.path.join(here, "test_randomized_gpu.py")))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", range(10))
def test_batch_chain_random_receivers(oracle, seed):
    """A batch of receivers nobody hand-picked: random modes, random filter edges inside each mode's limits (so the
    receivers fall into several decimator plan groups), random AGC settings (hang, slope, threshold, decay; a few with
    the AGC off on manual gain), random carrier offsets and levels, two calls of random whole-window lengths -- each
    receiver against its own oracle CDemodulator under the chain rule of test_postchain_gpu.py, burst by burst from
    the first, the S-meters within 0.02 dB."""
    import cutesdr_amd as ca
    import test_postchain_gpu as T
    from util_signals import fm_carrier, am_carrier, tones_plus_noise
    rng = np.random.default_rng(4200 + seed)
    fs, C = 2e6, 10
    names = ["AM", "SAM", "FM", "USB", "LSB", "CWU"]
    cfg = []
    for c in range(C):
        name = names[int(rng.integers(0, len(names)))] if c >= len(names) else names[c]
        m, kw = T.MODES[name]
        kw = dict(kw)
        if name in ("AM", "SAM"):
            hw = int(rng.integers(20, 80)) * 100                        # 2 .. 8 kHz either side
            kw.update(HiCut=hw, LowCut=-hw)
        elif name == "FM":
            hw = int(rng.integers(50, 120)) * 100                       # 5 .. 12 kHz
            kw.update(HiCut=hw, LowCut=-hw)
        elif name == "USB":
            kw.update(HiCut=int(rng.integers(18, 36)) * 100, LowCut=int(rng.integers(1, 3)) * 100)
        elif name == "LSB":
            kw.update(HiCut=-int(rng.integers(1, 3)) * 100, LowCut=-int(rng.integers(18, 36)) * 100)
        else:
            hw = int(rng.integers(2, 9)) * 100
            kw.update(HiCut=hw, LowCut=-hw)
        # (SAM keeps its AGC: with the loop pulling in on a carrier amplified by a fixed 30-50 dB, the start-up difference
        # of its first burst scales with that gain -- 2e-3 ... 2e-2 of full scale -- and says nothing about the chain)
        agc_on = name == "SAM" or rng.random() < 0.8
        kw.update(AgcOn=int(agc_on), AgcHangOn=int(rng.random() < 0.3), AgcSlope=int(rng.integers(0, 11)),
                  AgcThresh=int(rng.integers(-110, -30)), AgcDecay=int(rng.integers(30, 1500)),
                  AgcManualGain=int(rng.integers(0, 31)))        # (a fixed 46-49 dB puts USB audio -- and with it the fp32
                                                                  # rounding both sides differ by -- at 250 x full scale)
        off = 100e3 + 700.0 * c + float(rng.integers(-200, 200))
        dbfs = float(rng.uniform(-45.0, -8.0))
        cfg.append((name, m, kw, off, dbfs))
    calls = [19968 * int(rng.integers(8, 14)), 19968 * int(rng.integers(3, 9))]
    n = sum(calls)
    x = np.empty((C, n), dtype=np.complex64)
    for c, (name, m, kw, off, dbfs) in enumerate(cfg):
        if name == "FM":
            s = fm_carrier(n, fs, off, dbfs=dbfs)
        elif name in ("AM", "SAM"):
            s = am_carrier(n, fs, off, dbfs=dbfs, channel=c)
        else:
            d = {"USB": 1100.0, "LSB": -1100.0, "CWU": 0.0}[name]
            s = tones_plus_noise(9 + c, n, fs, [off + d, off + 1.6 * d + 250.0]) * 10 ** ((dbfs + 20.0) / 20.0)
        x[c] = s.astype(np.complex64)
    b = ca.DemodBatch(C, 2048)
    b.set_input_rate(fs)
    refs = []
    for c, (name, m, kw, off, dbfs) in enumerate(cfg):
        b.set_demod(c, m, T.info(ca, **kw))
        r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, T.info(oracle, **kw)); r.SetDemodFreq(-off)
        refs.append(r)
    b.commit()
    for c, (name, m, kw, off, dbfs) in enumerate(cfg):
        b.set_freq(c, -off)
    assert b.group_count()[0] >= 2                               # the receivers do spread over plan groups
    # the loop-only check beside the end-to-end one: every receiver's stages behind the filter, as oracle objects,
    # fed the GPU chain's OWN filter output (stage tap 2 of the batch) -- no start-up allowance at all, from the first sample
    from test_chain_taps_gpu import _oracle_post_chain
    b.set_taps(2)
    post = [_oracle_post_chain(oracle, name, False, refs[c].GetOutputRate(), kw) for c, (name, m, kw, off, dbfs) in enumerate(cfg)]
    a0, first = 0, [0] * C
    for ncall in calls:
        got = b.process(x[:, a0:a0 + ncall])
        for c, (name, m, kw, off, dbfs) in enumerate(cfg):
            want = refs[c].process_append(x[c, a0:a0 + ncall].astype(np.complex128))
            assert len(got[c]) == len(want), (seed, c, name)        # (a narrow plan may not fill a hop in a short call)
            if not len(want):
                continue
            own = post[c](b.tap(c, 2))[1]
            loop = T.burst_errors(got[c], own)
            gain_allow = 1.0 if kw["AgcOn"] else max(1.0, np.abs(own).max() / T.FULL_SCALE)   # (manual gain: audio beyond full scale)
            assert (loop <= (3e-5 if name == "FM" else 2e-5) * T.FULL_SCALE * gain_allow).all(), \
                (seed, c, name, "post-chain on its own filter output", loop[:6] / T.FULL_SCALE)
            errs = T.burst_errors(got[c], want)
            # (SAM locking onto a carrier up to 200 Hz off its tuning: 3e-3 ... 2e-2 of full scale in the burst of the
            # pull-in -- by seed and by the last bit of the samples in front of the loop: the same seeds moved between
            # 6e-3 and 1.9e-2 when the filter kernel and the oscillator's re-anchor points changed in round 5 --, 3e-4 ...
            # 4e-3 in the next, 2e-6 from the third: the chain rule's steady bound from burst 2 stands)
            T.check_chain_bursts(errs, name, first[c], (seed, c, name, kw),
                                 from_zero=(3e-2 * FULL_SCALE if name == "SAM" else T.FROM_ZERO))
            first[c] += len(want) // 1024
        a0 += ncall
    sm = b.smeter_all()
    for c in range(C):
        assert abs(float(sm[c]) - refs[c].GetSMeterAve()) <= 0.02, (seed, c, cfg[c][0])
