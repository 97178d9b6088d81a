"""GPU parity of the sample-rate stages (K4: S-meter, AGC, demodulators, CFir, CIir) and of the
whole CDemodulator chain against the fp64 oracle, through the C ABI.
Tolerances, the same rule in every test of this file, applied burst by burst (a burst = one
FastFIR hop of audio):
  * a stage on the oracle's own input (the leaf objects): 2e-5 of full scale from the first sample, PLLs from the
    burst after they have locked (measured 1e-6 .. 6e-6);
  * the whole chain, AM / SSB / CW: 5e-4 of full scale from sample 0, 2e-5 from the third burst; SAM mono 2.8e-3 / 5.4e-4
    in its first two bursts, SAM stereo within the audio range there (bistable: startup_bounds.py), then the same;
  * the whole chain, FM: the burst of the pull-in is arbitrary, then 5.6e-2, 9.8e-3, 2e-3, 3.8e-4, 7.4e-5 one to
    five bursts behind it and 3e-5 from the sixth -- each the ORACLE's own spread under an fp32 filter's error floor x 2
    (startup_bounds.py, reproduced on the CPU by test_oracle_independent.py::test_startup_spread_*);
identical squelch decisions, exact sample counts; the leaf filters are compared much tighter (they are linear)."""
import numpy as np
import pytest
from util_signals import tones_plus_noise, fm_carrier, am_carrier, FULL_SCALE

pytestmark = pytest.mark.gpu

import startup_bounds as SB

STEADY = 2e-5 * FULL_SCALE
FROM_ZERO = 5e-4 * FULL_SCALE
# The start-up bounds are DERIVED: tests/startup_bounds.py holds the per-burst spread of the oracle chain against itself under
# an fp32 filter's error floor (tests/test_oracle_independent.py::test_startup_spread_* reproduces every number on the CPU)
# and the factor on top of it; the stages behind the filter are pinned without any allowance by
# tests/test_chain_taps_gpu.py::test_post_chain_on_the_gpus_own_filter_output_from_the_first_sample.
SAM_FIRST = SB.SAM_FIRST * FULL_SCALE          # SAM mono chain, the stream's first burst (2 x 1.4e-3)
SAM_SECOND = SB.SAM_SECOND * FULL_SCALE        # SAM mono chain, the second (2 x 2.7e-4)
FM_STARTUP = [None if b is None else b * FULL_SCALE for b in SB.FM_STARTUP]   # FM chain, k bursts behind the pull-in
FM_SECOND, FM_THIRD = FM_STARTUP[1], FM_STARTUP[2]
FM_LOCKED = 1e-3 * FULL_SCALE          # FM chain after a control call in mid-stream (tests/test_rate_change_gpu.py)
FM_STEADY = 3e-5 * FULL_SCALE          # FM chain, steady state


def burst_errors(got, want, hop=1024):
    """max |got - want| of every burst of `hop` audio samples"""
    assert len(got) == len(want)
    return np.array([np.abs(got[j:j + hop] - want[j:j + hop]).max() for j in range(0, len(want), hop)])


def fm_start_late(first_burst_err):
    """bursts by which the FM bounds start later: 1 when the stream's very first burst differs by more than a fifth of
    full scale (the start-up difference decays by ~5 per burst from whatever that burst left)"""
    return 1 if first_burst_err > 0.2 * FULL_SCALE else 0


def check_chain_bursts(errs, mode, first_burst=0, what="", fm_late=0, from_zero=FROM_ZERO, stereo=False):
    """the chain rule of the module docstring; errs[i] belongs to burst first_burst + i of the stream.  fm_late: bursts
    by which the FM bounds start later (given by the caller, or found from the stream's first burst when errs starts
    there: fm_start_late)"""
    errs = np.asarray(errs, dtype=float)
    idx = first_burst + np.arange(len(errs))
    given_late = fm_late
    start = 0                          # the burst in which the loop pulls in: the stream's first one with audio
    if mode == "FM" and first_burst == 0 and len(errs):
        fm_late = max(fm_late, fm_start_late(errs[0]))
        big = np.nonzero(errs[:3] > 0.2 * FULL_SCALE)[0]
        start = int(big[0]) if len(big) else 0
    if mode == "FM":
        # The FM chain's first burst with audio is the PLL pulling in from zero state on the filter's start-up: the fp32
        # filter's error floor there moves the ORACLE's own burst by 0.8 ... 1.6 of full scale, and whatever that burst
        # left decays by ~5 per burst -- on both sides (startup_bounds.py; a 10 MSPS chain's first burst is silent on both
        # sides and its pull-in is burst 1).  So: every word finite and within the audio range, then burst by burst.
        assert np.isfinite(errs).all() and (errs <= 2.5 * FULL_SCALE).all(), (what, mode, errs[:4] / FULL_SCALE)
        # k bursts behind the pull-in: the oracle's own spread there x 2 (startup_bounds.py), then the steady bound
        for k in range(1, len(FM_STARTUP)):
            assert (errs[idx >= start + k + given_late] <= FM_STARTUP[k]).all(), (what, mode, k, errs[:10] / FULL_SCALE)
        assert (errs[idx >= start + len(FM_STARTUP) + given_late] <= FM_STEADY).all(), (what, mode, errs[:12] / FULL_SCALE)
    else:
        # (SAM: in the stream's first burst the AGC is at full gain on the filter's start-up -- samples of rounding size --
        # and the 100 Hz loop pulls in on a carrier of arbitrary phase over the first two bursts: how far the phase error
        # swings on the way is set by the rounding noise of the filter kernel in front of it, 0.4 ... 1.5e-3 of full scale
        # with the 2048-point kernels the library has had.  4e-3 in the first burst -- the bound the longer filters' AM
        # start-up already has -- 1e-3 in the second, the common bound from the third.)
        if mode == "SAM" and stereo:
            # bistable start (startup_bounds.py): the oracle itself lands 0.48 / 1.70 of full scale apart in bursts 0 and 1
            assert np.isfinite(errs).all() and (errs[idx <= 1] <= SB.SAM_STEREO_BISTABLE * FULL_SCALE).all(), (what, mode, errs[:4] / FULL_SCALE)
        elif mode == "SAM":
            assert (errs[idx == 0] <= max(from_zero, SAM_FIRST)).all() and (errs[idx == 1] <= max(from_zero, SAM_SECOND)).all(), \
                (what, mode, errs[:6] / FULL_SCALE)
        else:
            assert (errs <= from_zero).all(), (what, mode, errs[:6] / FULL_SCALE)
        assert (errs[idx >= 2] <= STEADY).all(), (what, mode, errs[:8] / FULL_SCALE)


def info(oracle_or_ca, **kw):
    base = dict(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000, LowCutmax=-5000,
                FilterClickResolution=100, Offset=0, SquelchValue=0, AgcSlope=0, AgcThresh=-100,
                AgcManualGain=30, AgcDecay=200, AgcOn=1, AgcHangOn=0, Symetric=1)
    base.update(kw)
    return oracle_or_ca.DemodInfo(**base)


MODES = {
    "FM": (2, dict()),
    "AM": (0, dict(HiCutmin=500, HiCutmax=10000, LowCutmax=-500, LowCutmin=-10000)),
    "SAM": (1, dict(HiCutmin=100, HiCutmax=10000, LowCutmax=-100, LowCutmin=-10000, Symetric=0)),
    "USB": (3, dict(HiCut=2800, LowCut=100, HiCutmin=500, HiCutmax=20000, LowCutmax=200, LowCutmin=0, Symetric=0)),
    "LSB": (4, dict(HiCut=-100, LowCut=-2800, HiCutmin=-200, HiCutmax=0, LowCutmax=-500, LowCutmin=-20000, Symetric=0)),
    "CWU": (5, dict(HiCut=500, LowCut=-500, HiCutmin=50, HiCutmax=1000, LowCutmax=-50, LowCutmin=-1000, Offset=700, Symetric=0)),
}


def test_fir_design_and_filtering(oracle):
    import cutesdr_amd as ca
    rng = np.random.default_rng(1)
    x = rng.standard_normal(3000) * 1000
    xc = x + 1j * rng.standard_normal(3000) * 1000
    for kind, args in (("lp", (1.0, 50.0, 5000, 9000, 31250.0)), ("hp", (1.0, 50.0, 5000, 3000, 62500.0)),
                       ("lp", (1.0, 40.0, 4500, 5500, 31250.0))):
        f, r = ca.CFir(), oracle.CFir()
        n1 = (f.InitLPFilter if kind == "lp" else f.InitHPFilter)(*args)
        n2 = (r.InitLPFilter if kind == "lp" else r.InitHPFilter)(*args)
        assert n1 == n2
        if args[1] == 40.0:
            f.GenerateHBFilter(5000.0); r.GenerateHBFilter(5000.0)
        for a, b in zip(f.taps(), r.taps()):
            np.testing.assert_allclose(a, b, atol=1e-14)
        for part in (slice(0, 1000), slice(1000, 3000)):          # state carries over calls
            np.testing.assert_allclose(f.ProcessFilter(x[part]), r.ProcessFilter(x[part]), atol=2e-3)
            got, want = f.ProcessFilter(xc[part]), r.ProcessFilter(xc[part])
            np.testing.assert_allclose(got, want, atol=4e-3)
    f, r = ca.CFir(), oracle.CFir()
    f.InitConstFir([0.25, 0.5, 0.25]); r.InitConstFir([0.25, 0.5, 0.25])
    np.testing.assert_allclose(f.ProcessFilter(x), r.ProcessFilter(x), atol=1e-3)


def test_iir_design_and_filtering(oracle):
    import cutesdr_amd as ca
    rng = np.random.default_rng(2)
    x = rng.standard_normal(4000) * 1000
    xc = x + 1j * rng.standard_normal(4000) * 1000
    for kind, f0, q, fs in (("LP", 3000.0, 1.0, 62500.0), ("HP", 300.0, 0.7, 31250.0), ("BP", 700.0, 5.0, 15625.0),
                            ("BR", 25000, 1000.0, 100000)):
        f, r = ca.CIir(), oracle.CIir()
        f.Init(kind, f0, q, fs); r.Init(kind, f0, q, fs)
        np.testing.assert_allclose(f.coefs(), r.coefs(), rtol=1e-14)
        np.testing.assert_allclose(f.ProcessFilter(x), r.ProcessFilter(x), atol=5e-3)
        np.testing.assert_allclose(f.ProcessFilter(xc), r.ProcessFilter(xc), atol=5e-3)


def level_steps(n, fs, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    env = np.where((t % 0.4) < 0.2, 3000.0, 60.0) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t))
    return env * np.exp(2j * np.pi * 1000 * t) + 5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))


@pytest.mark.parametrize("hang,slope,thresh,decay", [(0, 0, -100, 200), (1, 5, -60, 500), (0, 10, -20, 50)])
def test_agc_complex_and_real(oracle, hang, slope, thresh, decay):
    import cutesdr_amd as ca
    fs = 62500.0
    x = level_steps(62500, fs, 3)
    a, r = ca.CAgc(), oracle.CAgc()
    a.SetParameters(True, bool(hang), thresh, 30, slope, decay, fs)
    r.SetParameters(True, bool(hang), thresh, 30, slope, decay, fs)
    for part in (slice(0, 8192), slice(8192, 62500)):
        got, want = a.ProcessData(x[part]), r.ProcessData(x[part])
        assert np.abs(got - want).max() <= STEADY, np.abs(got - want).max() / FULL_SCALE
    a2, r2 = ca.CAgc(), oracle.CAgc()
    a2.SetParameters(True, bool(hang), thresh, 30, slope, decay, fs)
    r2.SetParameters(True, bool(hang), thresh, 30, slope, decay, fs)
    got, want = a2.ProcessData(x.real.copy()), r2.ProcessData(x.real.copy())
    assert np.abs(got - want).max() <= STEADY, np.abs(got - want).max() / FULL_SCALE
    a2.SetParameters(False, bool(hang), thresh, 45, slope, decay, fs)       # manual gain
    r2.SetParameters(False, bool(hang), thresh, 45, slope, decay, fs)
    np.testing.assert_allclose(a2.ProcessData(x[:1000]), r2.ProcessData(x[:1000]), rtol=1e-6, atol=1e-3)


def test_smeter(oracle):
    import cutesdr_amd as ca
    fs = 62500.0
    x = level_steps(40000, fs, 4)
    s, r = ca.CSMeter(), oracle.CSMeter()
    for part in (slice(0, 8192), slice(8192, 40000)):
        s.ProcessData(x[part], fs); r.ProcessData(x[part], fs)
        assert s.GetAve() == pytest.approx(r.GetAve(), abs=0.01)
    big = 40000.0 * np.exp(2j * np.pi * 0.01 * np.arange(500))      # > full scale: peak above 0 dB is held
    s.ProcessData(big, fs); r.ProcessData(big, fs)
    assert s.GetPeak() == pytest.approx(r.GetPeak(), abs=0.01)
    assert s.GetPeak() == pytest.approx(r.GetPeak(), abs=0.01) == pytest.approx(5.0)   # reset on read


def test_am_sam_fm_demod_leaves(oracle):
    import cutesdr_amd as ca
    L = 1024
    fs = 31250.0
    x = am_carrier(8 * L, fs, 150.0, fmod=800.0, depth=0.6, dbfs=-12.0)
    for stereo in (False, True):
        d, r = ca.CAmDemod(fs), oracle.CAmDemod(fs)
        d.SetBandwidth(4000.0); r.SetBandwidth(4000.0)
        for i in range(8):
            got, want = d.ProcessData(x[i * L:(i + 1) * L], stereo), r.ProcessData(x[i * L:(i + 1) * L], stereo)
            assert np.abs(got - want).max() <= STEADY, ("am", i, np.abs(got - want).max() / FULL_SCALE)
        d, r = ca.CSamDemod(fs), oracle.CSamDemod(fs)
        for i in range(8):
            got, want = d.ProcessData(x[i * L:(i + 1) * L], stereo), r.ProcessData(x[i * L:(i + 1) * L], stereo)
            # acquisition included: the walk of an unlocked tile follows the oracle sample by sample
            assert np.abs(got - want).max() <= STEADY, ("sam", i, np.abs(got - want).max() / FULL_SCALE)
    fs = 62500.0
    x = fm_carrier(16 * L, fs, 300.0, fmod=1000.0, dev=3000.0, dbfs=-6.0, noise_dbfs=-60.0)
    for stereo in (False, True):
        d, r = ca.CFmDemod(fs), oracle.CFmDemod(fs)
        d.SetSquelch(50); r.SetSquelch(50)
        for i in range(16):
            got, want = d.ProcessData(x[i * L:(i + 1) * L], 5000.0, stereo), r.ProcessData(x[i * L:(i + 1) * L], 5000.0, stereo)
            assert d.squelched() == r.squelched(), i
            assert np.abs(got - want).max() <= STEADY, ("fm", i, np.abs(got - want).max() / FULL_SCALE)
        assert not d.squelched()                            # a clean carrier opens the squelch
    np.testing.assert_array_equal(ca.ssb_demod(x[:100]), oracle.ssb_demod(x[:100]))
    np.testing.assert_array_equal(ca.ssb_demod(x[:100], True), oracle.ssb_demod(x[:100], True))


def make_input(mode, n, fs):
    if mode == "FM":
        return fm_carrier(n, fs, 100e3, dbfs=-20.0)
    if mode in ("AM", "SAM"):
        return am_carrier(n, fs, 100e3, dbfs=-20.0)
    off = {"USB": 1200.0, "LSB": -1200.0, "CWU": 0.0}[mode]
    return tones_plus_noise(9, n, fs, [100e3 + off, 100e3 + off * 1.7 + 300.0])


@pytest.mark.parametrize("mode", ["FM", "AM", "SAM", "USB", "LSB", "CWU"])
def test_cdemodulator_chain_reference_call_pattern(oracle, mode):
    """CDemodulator drop-in: 2 MSPS, 256-sample host calls (netiobase.cpp:59-60), reference filter
    size 2048; counts, audio, S-meter and the overwrite-at-out[0] semantics match."""
    import cutesdr_amd as ca
    m, kw = MODES[mode]
    fs = 2e6
    d, r = ca.CDemodulator(2048), oracle.CDemodulator(2048)
    for obj, mod in ((d, ca), (r, oracle)):
        obj.SetInputSampleRate(fs)
        obj.SetDemod(m, info(mod, **kw))
        obj.SetDemodFreq(-100e3)
    assert d.GetOutputRate() == r.GetOutputRate()
    assert d.buf_limit() == r.buf_limit()
    n = 19968 * (64 if mode == "CWU" else 24)              # CW decimates by 128: seven bursts take longer
    x = make_input(mode, n, fs)
    total_g = total_r = 0
    errs = []
    for i in range(0, n, 256 * 13):                        # uneven relation to the 19968 window
        kg, og = d.ProcessData(x[i:i + 256 * 13])
        kr, orr = r.ProcessData(x[i:i + 256 * 13])
        assert kg == kr
        total_g += kg; total_r += kr
        if kr:                                              # every burst the host gets to see, from the first one
            assert kr % 1024 == 0
            errs.append(np.abs(og[:1024] - orr[:1024]).max())
            if mode == "FM":
                assert (not og[:1024].any()) == (not orr[:1024].any())       # same squelch decision
    assert total_g == total_r > 0 and len(errs) >= 7
    check_chain_bursts(errs, mode, what="256-sample host calls")
    assert d.GetSMeterAve() == pytest.approx(r.GetSMeterAve(), abs=0.02)


@pytest.mark.parametrize("stereo", [False, True], ids=["mono", "stereo"])
def test_config_c2_16384_filter_every_burst(oracle, stereo):
    """BASELINE config C2: one receiver, 2 MSPS -> CDownConvert -> 16384-pt CFastFIR -> AGC -> FM, both overloads of
    ProcessData, fed window by window (m_InBufLimit samples per call); every 8192-sample burst from the first one
    under the chain rule, identical squelch decisions; then the append form of the batch API on the same stream."""
    import cutesdr_amd as ca
    fs = 2e6
    d, r = ca.CDemodulator(16384), oracle.CDemodulator(16384)
    for obj, mod in ((d, ca), (r, oracle)):
        obj.SetInputSampleRate(fs); obj.SetDemod(2, info(mod)); obj.SetDemodFreq(-100e3)
    lim = d.buf_limit()
    assert lim == r.buf_limit()
    n = lim * 132                                           # 82368 decimated samples: ten bursts of 8192
    x = make_input("FM", n, fs)
    errs = []
    for i in range(0, n, lim):
        kg, og = d.ProcessData(x[i:i + lim], stereo)
        kr, orr = r.ProcessData(x[i:i + lim], stereo)
        assert kg == kr and kr in (0, 8192)
        if kr:
            errs.append(np.abs(og[:kr] - orr[:kr]).max())
            assert (not og[:kr].any()) == (not orr[:kr].any())
            assert np.abs(orr[:kr]).max() > 100.0 or len(errs) == 1
    assert len(errs) == 10
    check_chain_bursts(errs, "FM", what="C2")
    assert d.GetSMeterAve() == pytest.approx(r.GetSMeterAve(), abs=0.02)
    if not stereo:
        d2, r2 = ca.CDemodulator(16384), oracle.CDemodulator(16384)
        for obj, mod in ((d2, ca), (r2, oracle)):
            obj.SetInputSampleRate(fs); obj.SetDemod(2, info(mod)); obj.SetDemodFreq(-100e3)
        got, want = d2.process_append(x), r2.process_append(x)
        assert len(got) == len(want) == 10 * 8192
        check_chain_bursts(burst_errors(got, want, 8192), "FM", what="C2 append form")


@pytest.mark.parametrize("nfft", [2048, 4096, 8192])
def test_demod_batch_mixed_modes(oracle, nfft):
    """(4096 / 8192: the batch chain on the other filter kernels -- hop nfft / 2, bursts of that length)"""
    import cutesdr_amd as ca
    fs, C = 2e6, 6
    names = ["AM", "FM", "USB", "FM", "SAM", "LSB"]
    hop = nfft // 2
    b = ca.DemodBatch(C, nfft)
    b.set_input_rate(fs)
    refs = []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        b.set_demod(c, m, info(ca, **kw))
        r = oracle.CDemodulator(nfft)
        r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-100e3 - 1000.0 * c)
        refs.append(r)
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3 - 1000.0 * c)
        assert b.output_rate(c) == refs[c].GetOutputRate()
    n = refs[0].buf_limit() * 16 * (nfft // 2048)
    x = np.stack([make_input(names[c], 2 * n, fs) * np.exp(2j * np.pi * 1000.0 * c * np.arange(2 * n) / fs) for c in range(C)])
    first = [0] * C
    for part in (x[:, :n], x[:, n:]):
        got = b.process(part)
        for c in range(C):
            want = refs[c].process_append(part[c])
            assert len(got[c]) == len(want) > 0 and len(want) % hop == 0, (c, names[c])
            # (the first burst of a longer filter holds more of its start-up -- the AGC at full gain on samples of
            # rounding size: AM 1.6e-3 / 1.8e-3 of full scale at 4096 / 8192 points, 2e-6 from the second burst)
            check_chain_bursts(burst_errors(got[c], want, hop), names[c] if names[c] in ("FM", "SAM") else "other", first[c], (c, names[c]),
                               from_zero=FROM_ZERO if nfft == 2048 else 4e-3 * FULL_SCALE)
            first[c] += len(want) // hop
    for c in range(C):
        assert b.smeter_ave(c) == pytest.approx(refs[c].GetSMeterAve(), abs=0.02)


@pytest.mark.parametrize("pipeline", ["0", "1"])
def test_batch_long_calls_fused_and_pipelined(oracle, pipeline, monkeypatch):
    """Long calls (>= 16 FastFIR hops) through the fused launch and through the optional stage
    pipeline (S-meter | AGC | demodulator as concurrent launches over burst groups, capi_demod.hip
    ChainCore::post, CSDR_CHAIN_PIPELINE=1): same results as the oracle, every burst of both calls under
    the chain rule.  The switch is read once per process, so the
    pipelined case runs in a child interpreter."""
    if pipeline == "1":
        import subprocess, sys, os
        env = dict(os.environ, CSDR_CHAIN_PIPELINE="1")
        code = ("import sys; sys.path.insert(0, %r); import pytest; "
                "sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', "
                "%r + '::test_batch_long_calls_fused_and_pipelined[0]']))" % (os.path.dirname(__file__), __file__))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return
    import cutesdr_amd as ca
    fs, C = 2e6, 4
    names = ["FM", "USB", "SAM", "AM"]
    b = ca.DemodBatch(C, 2048)
    b.set_input_rate(fs)
    refs = []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        b.set_demod(c, m, info(ca, **kw))
        r = oracle.CDemodulator(2048)
        r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-100e3 - 1000.0 * c)
        refs.append(r)
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3 - 1000.0 * c)
    n = 19968 * 64                                            # FM/USB: 39 hops per call, AM/SAM: 19
    x = np.stack([make_input(names[c], 2 * n, fs) * np.exp(2j * np.pi * 1000.0 * c * np.arange(2 * n) / fs) for c in range(C)])
    for call, part in enumerate((x[:, :n], x[:, n:])):
        got = b.process(part)
        for c in range(C):
            want = refs[c].process_append(part[c])
            assert len(got[c]) == len(want) >= 16 * 1024, (c, names[c])
            check_chain_bursts(burst_errors(got[c], want), "FM" if names[c] == "FM" else "other",
                               call * (len(want) // 1024), (c, names[c], call))
    for c in range(C):
        assert b.smeter_ave(c) == pytest.approx(refs[c].GetSMeterAve(), abs=0.02)


def test_config_c5_10msps_fm_to_resampler(oracle):
    """BASELINE config 5: one channel at 10 MSPS -> CDownConvert -> CFastFIR -> CFmDemod ->
    CFractResampler to 48 kHz.  Decimated rate 78125 (chain 3,11,11,11,11,15,27), resampler rate
    78125/48000; every burst of the audio and of the resampled stream under the chain rule, sample counts exact at
    every step."""
    import cutesdr_amd as ca
    fs = 10e6
    d, r = ca.CDemodulator(2048), oracle.CDemodulator(2048)
    for obj, mod in ((d, ca), (r, oracle)):
        obj.SetInputSampleRate(fs); obj.SetDemod(2, info(mod)); obj.SetDemodFreq(-1.2e6)
    assert d.GetOutputRate() == r.GetOutputRate() == 78125.0
    assert d.buf_limit() == r.buf_limit()
    rg, rr = ca.CFractResampler(), oracle.CFractResampler()
    rg.Init(8192); rr.Init(8192)
    rate = 78125.0 / 48000.0
    n = d.buf_limit() * 24                                    # 18 bursts of 1024 at 78 125 S/s
    x = fm_carrier(n, fs, 1.2e6, fmod=1000.0, dev=3000.0, dbfs=-20.0)
    outs_g, outs_r, audio_errs = [], [], []
    for k in range(0, n, d.buf_limit() * 4):
        a_g, a_r = d.process_append(x[k:k + d.buf_limit() * 4]), r.process_append(x[k:k + r.buf_limit() * 4])
        assert len(a_g) == len(a_r)
        for j in range(0, len(a_g), 1024):                    # the sound sink feeds the resampler hop by hop
            audio_errs.append(np.abs(a_g[j:j + 1024] - a_r[j:j + 1024]).max())
            outs_g.append(rg.Resample(a_g[j:j + 1024], rate)); outs_r.append(rr.Resample(a_r[j:j + 1024], rate))
            assert len(outs_g[-1]) == len(outs_r[-1])
    assert len(outs_r) >= 14 and sum(len(o) for o in outs_r) > 8000
    check_chain_bursts(audio_errs, "FM", what="C5 audio")
    # the resampled stream, burst by burst (one resampler call each), under the same rule: a 28-tap interpolator with
    # unit gain neither amplifies the difference nor moves it by more than its 14-sample delay
    check_chain_bursts([np.abs(g - w).max() for g, w in zip(outs_g, outs_r)], "FM", what="C5 resampled")
    assert np.abs(np.concatenate(outs_r[6:])).max() > 100.0   # there is audio to compare


def test_pll_unlockable_carrier_and_relock(oracle):
    """The PLL tiles are solved as a linear system under the guess 'locked, clamp idle' and walked
    sample by sample when the guess fails its check.  A carrier outside the loop's frequency clamp
    (FM +-6 kHz, SAM +-1 kHz) never satisfies it, a carrier inside does: both paths, and the switch
    between them inside one stream, match the oracle."""
    import cutesdr_amd as ca
    L = 1024
    fs = 62500.0
    n = 12 * L
    t = np.arange(3 * n) / fs
    seg = lambda f, k: 8000.0 * np.exp(2j * np.pi * f * t[k * n:(k + 1) * n])
    for stereo in (False, True):
        x = np.concatenate([seg(9000.0, 0), seg(800.0, 1), seg(-11000.0, 2)])      # out, in, out of the clamp
        d, r = ca.CFmDemod(fs), oracle.CFmDemod(fs)
        d.SetSquelch(50); r.SetSquelch(50)
        for i in range(3 * n // L):
            got, want = d.ProcessData(x[i * L:(i + 1) * L], 5000.0, stereo), r.ProcessData(x[i * L:(i + 1) * L], 5000.0, stereo)
            assert d.squelched() == r.squelched(), i
            if i % 12 >= 2:                                 # two hops after each frequency jump
                assert np.abs(got - want).max() <= STEADY, (stereo, i, np.abs(got - want).max() / FULL_SCALE)
    fs = 31250.0
    t = np.arange(3 * n) / fs
    for stereo in (False, True):
        x = np.concatenate([seg(3000.0, 0), seg(200.0, 1), seg(-2500.0, 2)]) * (1.0 + 0.3 * np.sin(2 * np.pi * 700.0 * t))
        d, r = ca.CSamDemod(fs), oracle.CSamDemod(fs)
        for i in range(3 * n // L):
            got, want = d.ProcessData(x[i * L:(i + 1) * L], stereo), r.ProcessData(x[i * L:(i + 1) * L], stereo)
            if i % 12 >= 4:
                assert np.abs(got - want).max() <= STEADY, (stereo, i, np.abs(got - want).max() / FULL_SCALE)


def test_leaf_objects_ragged_call_lengths(oracle):
    """Leaf objects take any call length: 1-sample calls, lengths that leave partly filled lanes in
    the scans (17, 1000), lengths that span several 1024-sample tiles (1025, 2500, 4099)."""
    import cutesdr_amd as ca
    cuts = np.cumsum([0, 1, 17, 1000, 1025, 2500, 1, 4099, 333, 2048])
    fs = 31250.0
    x = am_carrier(int(cuts[-1]), fs, 120.0, fmod=600.0, depth=0.5, dbfs=-10.0)
    g, r = ca.CAgc(), oracle.CAgc()
    for o in (g, r):
        o.SetParameters(True, False, -80, 0, 2, 300, fs)
    ga, ra = ca.CAmDemod(fs), oracle.CAmDemod(fs)
    gs, rs = ca.CSamDemod(fs), oracle.CSamDemod(fs)
    gm, rm = ca.CSMeter(), oracle.CSMeter()
    for k in range(len(cuts) - 1):
        part = x[cuts[k]:cuts[k + 1]]
        got, want = g.ProcessData(part), r.ProcessData(part)
        assert len(got) == len(want) == len(part)
        assert np.abs(got - want).max() <= STEADY, ("agc", k, np.abs(got - want).max() / FULL_SCALE)
        assert np.abs(ga.ProcessData(part) - ra.ProcessData(part)).max() <= STEADY, ("am", k)
        sg, sr = gs.ProcessData(part), rs.ProcessData(part)
        assert np.abs(sg - sr).max() <= STEADY, ("sam", k, np.abs(sg - sr).max() / FULL_SCALE)
        gm.ProcessData(part, fs); rm.ProcessData(part, fs)
        assert gm.GetAve() == pytest.approx(rm.GetAve(), abs=0.02), ("smeter", k)
    fs = 62500.0
    x = fm_carrier(int(cuts[-1]), fs, 300.0, fmod=1000.0, dev=3000.0, dbfs=-6.0, noise_dbfs=-60.0)
    gf, rf = ca.CFmDemod(fs), oracle.CFmDemod(fs)
    gf.SetSquelch(50); rf.SetSquelch(50)
    for k in range(len(cuts) - 1):
        part = x[cuts[k]:cuts[k + 1]]
        got, want = gf.ProcessData(part, 5000.0), rf.ProcessData(part, 5000.0)
        assert gf.squelched() == rf.squelched(), k
        assert np.abs(got - want).max() <= STEADY, ("fm", k, np.abs(got - want).max() / FULL_SCALE)


def test_batch_above_1024_channels_takes_the_one_wave_kernel(oracle):
    """More than 1024 channels in one launch: one wave per channel (postchain_kernel<1>), and grids
    beyond the four-wave case everywhere else.  Every channel gets the same stream and tuning, three
    of them are compared with one oracle run."""
    import cutesdr_amd as ca
    fs, C = 2e6, 1100
    m, kw = MODES["FM"]
    b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
    for c in range(C):
        b.set_demod(c, m, info(ca, **kw))
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3)
    r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-100e3)
    n = 19968 * 10
    x1 = make_input("FM", 2 * n, fs).astype(np.complex64)
    for call in range(2):
        part = np.broadcast_to(x1[call * n:(call + 1) * n], (C, n))
        got = b.process(part)
        want = r.process_append(x1[call * n:(call + 1) * n].astype(np.complex128))
        for c in (0, 517, C - 1):
            assert len(got[c]) == len(want)
            check_chain_bursts(burst_errors(got[c], want), "FM", call * (len(want) // 1024), (c, call))
    assert b.smeter_ave(C - 1) == pytest.approx(r.GetSMeterAve(), abs=0.02)


def _fm_batch_words(path, fastfir_n):
    """FM receivers with open, closed and toggling squelch through csdr_demod_batch, three calls; audio words -> path"""
    import cutesdr_amd as ca
    fs, C = 2e6, 6
    n = 19968 * (16 if fastfir_n == 2048 else 72)
    b = ca.DemodBatch(C, fastfir_n); b.set_input_rate(fs)
    for c in range(C):
        b.set_demod(c, 2, info(ca, SquelchValue=[0, 40, 99, 70, 0, 55][c]))
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3 - 800.0 * c)
    rng = np.random.default_rng(77)
    x = np.stack([make_input("FM", 3 * n, fs) * np.exp(2j * np.pi * 800.0 * c * np.arange(3 * n) / fs) for c in range(C)])
    x[4] = 300.0 * (rng.standard_normal(3 * n) + 1j * rng.standard_normal(3 * n))      # no carrier at all
    x[5, n:2 * n] = 3000.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))    # the carrier drowns in noise, then returns
    outs = []
    for k in range(3):
        outs.append(b.process(x[:, k * n:(k + 1) * n]))
    words = np.concatenate([np.concatenate([o[c] for o in outs]) for c in range(C)])
    np.save(path, words)
    return [sum(len(o[c]) for o in outs) for c in range(C)], [np.concatenate([o[c] for o in outs]) for c in range(C)]


@pytest.mark.parametrize("fastfir_n", [2048, 16384])
def test_fm_squelch_deferred_out_of_the_walk_gives_the_same_words(tmp_path, fastfir_n):
    """The squelch half of CFmDemod (high-pass, average, once-per-burst decision, low-pass: fmdemod.cpp:113-152) runs
    as burst-parallel launches behind the post-chain walk (PC_FM_DEFER) when a call holds several bursts.  Same scans,
    same arithmetic: every audio word equals what the walk itself produces (CSDR_FM_DEFER=0, in a child process: the
    switch is read once) -- receivers whose squelch is open, shut, and opening / closing mid-stream, one-tile bursts
    (2048-point filter) and eight-tile bursts (16384-point filter)."""
    import subprocess, sys, os
    lens, got = _fm_batch_words(str(tmp_path / "deferred.npy"), fastfir_n)
    assert min(lens) > 4 * 1024
    assert np.abs(got[0]).max() > 100.0 and not got[2].any()          # open squelch / threshold 0: forced mute (:129-132)
    opened = [bool(got[5][j:j + 1024].any()) for j in range(0, len(got[5]) - 1023, 1024)]
    assert any(opened) and not all(opened)                             # the fading receiver's squelch moves
    code = ("import sys; sys.path.insert(0, %r); import test_postchain_gpu as T; T._fm_batch_words(%r, %d)"
            % (os.path.dirname(__file__), str(tmp_path / "walk.npy"), fastfir_n))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSDR_FM_DEFER="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, w = np.load(tmp_path / "deferred.npy"), np.load(tmp_path / "walk.npy")
    assert a.shape == w.shape and np.array_equal(a.view(np.uint32), w.view(np.uint32))


def _agc_batch_words(path, fastfir_n):
    """Receivers of every mode and AGC setting through csdr_demod_batch, three calls; audio words + S-meters -> path"""
    import cutesdr_amd as ca
    fs = 2e6
    cases = [("USB", dict()), ("USB", dict(AgcHangOn=1, AgcDecay=500, AgcThresh=-60)), ("AM", dict(AgcSlope=6, AgcDecay=50)),
             ("SAM", dict(AgcThresh=-40)), ("CWU", dict(AgcDecay=1000)), ("FM", dict()), ("LSB", dict(AgcOn=0, AgcManualGain=40)),
             ("AM", dict(AgcDecay=2000, AgcThresh=-120))]
    C = len(cases)
    n = 19968 * (16 if fastfir_n == 2048 else 72)
    b = ca.DemodBatch(C, fastfir_n); b.set_input_rate(fs)
    for c, (mode, kw) in enumerate(cases):
        m, base = MODES[mode]
        b.set_demod(c, m, info(ca, **dict(base, **kw)))
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3)
    rng = np.random.default_rng(78)
    x = np.stack([make_input(mode, 3 * n, fs) for mode, _ in cases])
    x[1, n // 2: n] *= 0.01                                        # a fade: the hang timer holds, then the decay runs
    x[2, n: 2 * n] += 8000.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * (rng.random(n) < 1e-3)   # impulses
    outs = []
    for k in range(3):
        outs.append(b.process(x[:, k * n:(k + 1) * n]))
    per = [np.concatenate([np.asarray(o[c]).ravel() for o in outs]) for c in range(C)]
    np.save(path, np.concatenate(per))
    return per


@pytest.mark.parametrize("fastfir_n", [2048, 16384])
def test_agc_peaks_ahead_of_the_walk_give_the_same_words(tmp_path, fastfir_n):
    """CAgc's log magnitudes and their sliding maximum (agc.cpp:196-231) need nothing from the loop, so a call of several
    bursts computes them burst-parallel in front of the walk (PC_AGC_PRE).  Same fp32 operations: every audio word equals
    the walk's own (CSDR_AGC_PRE=0, in a child process: the switch is read once) -- every mode, hang timer on and off,
    a fade, impulses, AGC off beside AGC on, one-tile and eight-tile bursts, state carried over three calls."""
    import subprocess, sys, os
    got = _agc_batch_words(str(tmp_path / "pre.npy"), fastfir_n)
    assert all(len(g) > 1024 and np.abs(g).max() > 10.0 for g in got)
    code = ("import sys; sys.path.insert(0, %r); import test_postchain_gpu as T; T._agc_batch_words(%r, %d)"
            % (os.path.dirname(__file__), str(tmp_path / "walk.npy"), fastfir_n))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSDR_AGC_PRE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, w = np.load(tmp_path / "pre.npy"), np.load(tmp_path / "walk.npy")
    assert a.shape == w.shape and np.array_equal(a.view(np.uint32), w.view(np.uint32))


def _fm_unlocked_words(path, fastfir_n, fs=2e6):
    """FM receivers that hear noise only, a carrier at the edge of lock, and a clean carrier; squelch open -> path"""
    import cutesdr_amd as ca
    C = 6
    n = 19968 * (16 if fastfir_n == 2048 else 72) * (4 if fs > 5e6 else 1)
    b = ca.DemodBatch(C, fastfir_n); b.set_input_rate(fs)
    for c in range(C):
        b.set_demod(c, 2, info(ca, SquelchValue=-160))            # threshold far up: the squelch stays open
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3)
    rng = np.random.default_rng(79)
    noise = lambda a: a * (rng.standard_normal(2 * n) + 1j * rng.standard_normal(2 * n))
    car = make_input("FM", 2 * n, fs)
    x = np.stack([noise(300.0), noise(3000.0), car + noise(1500.0), car + noise(4000.0), car, 0.02 * car + noise(30.0)])
    outs = [b.process(x[:, k * n:(k + 1) * n]) for k in range(2)]
    per = [np.concatenate([o[c] for o in outs]) for c in range(C)]
    np.save(path, np.concatenate(per))
    return per


@pytest.mark.parametrize("fastfir_n,fs", [(2048, 2e6), (16384, 2e6), (2048, 10e6)], ids=["2048-62k5", "16384-62k5", "2048-78k125"])
def test_fm_unlocked_pll_overlapped_walks_equal_the_sequential_walk(tmp_path, fastfir_n, fs):
    """A tile of CFmDemod's PLL (fmdemod.cpp:166-177) that is not locked -- an idle channel, a carrier in the noise --
    cannot be solved as one linear system; it used to fall to one thread walking 1024 samples.  pll_overlap runs the
    exact recurrence on every thread over its own samples, started early from a zero state, and checks that neighbours
    meet to 1e-9 turns.  Its audio must be the sequential walk's (CSDR_PLL_OVERLAP=0, child process) to 1e-6 of full
    scale on every sample: noise only, weak and strong carriers, two calls; at the two output rates of the BASELINE
    configs (62.5 kS/s: spectral radius 0.47, 48 samples of warm-up; 78.125 kS/s from a 10 MSPS radio: 0.56, 62)."""
    import subprocess, sys, os
    got = _fm_unlocked_words(str(tmp_path / "overlap.npy"), fastfir_n, fs)
    assert all(np.abs(g).max() > 100.0 for g in got)                 # squelch open everywhere: there is audio to compare
    code = ("import sys; sys.path.insert(0, %r); import test_postchain_gpu as T; T._fm_unlocked_words(%r, %d, %r)"
            % (os.path.dirname(__file__), str(tmp_path / "seq.npy"), fastfir_n, fs))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSDR_PLL_OVERLAP="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, w = np.load(tmp_path / "overlap.npy"), np.load(tmp_path / "seq.npy")
    assert a.shape == w.shape
    assert np.abs(a - w).max() <= 1e-6 * FULL_SCALE, np.abs(a - w).max()


@pytest.mark.parametrize("waves", [1, 8])
def test_chain_kernels_with_one_and_eight_waves_per_receiver(tmp_path, waves):
    """postchain_kernel<1> (launches of more than 1024 receivers) and <8> (CSDR_POSTCHAIN_WAVES) walk the same chain as
    the four-wave kernel the other tests run -- AGC peaks taken from the pre-pass, squelch deferred, unlocked FM tiles by
    overlapped walks.  Same words up to the rounding of their differently shaped scans (1e-5 of full scale); on
    receivers whose PLL is not locked a last-bit difference may send the two walks apart for a few samples, so
    there the bound holds for 99.9 % of the samples."""
    import subprocess, sys, os
    ref_a = np.concatenate(_agc_batch_words(str(tmp_path / "a4.npy"), 2048))
    ref_u = np.concatenate(_fm_unlocked_words(str(tmp_path / "u4.npy"), 2048))
    code = ("import sys; sys.path.insert(0, %r); import test_postchain_gpu as T; T._agc_batch_words(%r, 2048); T._fm_unlocked_words(%r, 2048)"
            % (os.path.dirname(__file__), str(tmp_path / "a.npy"), str(tmp_path / "u.npy")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSDR_POSTCHAIN_WAVES=str(waves)), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, u = np.load(tmp_path / "a.npy"), np.load(tmp_path / "u.npy")
    assert a.shape == ref_a.shape and u.shape == ref_u.shape
    assert np.abs(a - ref_a).max() <= 1e-5 * FULL_SCALE, np.abs(a - ref_a).max()
    assert (np.abs(u - ref_u) <= 1e-5 * FULL_SCALE).mean() >= 0.999
