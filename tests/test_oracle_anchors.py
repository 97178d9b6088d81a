"""Pin the fp64 CPU oracle against the known-answer anchors that SURVEY.md section 8c /
App. A.9 recorded from a run of the reference (the reference itself has no tests and
cannot be built here), plus independent numpy restatements of the published algorithms."""
import json
import os
import numpy as np
import pytest

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_anchors.json")))


def test_fft_sign_and_scale(oracle):
    # FwdFFT has the POSITIVE exponent, tone at +5 lands in bin N-5; Rev(Fwd(x)) = N x
    n = 1024
    t = np.arange(n)
    x = np.exp(2j * np.pi * 5 * t / n)
    X = oracle.fft(x, +1)
    assert np.argmax(np.abs(X)) == n - 5
    rng = np.random.default_rng(1)
    y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    np.testing.assert_allclose(oracle.fft(oracle.fft(y, +1), -1), n * y, rtol=0, atol=1e-9)
    # forward == N * ifft (numpy's ifft has the positive exponent and the 1/N)
    np.testing.assert_allclose(oracle.fft(y, +1), n * np.fft.ifft(y), atol=1e-9)
    np.testing.assert_allclose(oracle.fft(y, -1), np.fft.fft(y), atol=1e-9)


@pytest.mark.parametrize("n", [512, 2048, 16384, 65536])
def test_fft_sizes_vs_numpy(oracle, n):
    rng = np.random.default_rng(n)
    y = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    np.testing.assert_allclose(oracle.fft(y, -1), np.fft.fft(y), atol=1e-8 * np.sqrt(n))


def test_nco_envelope_anchor(oracle):
    # |osc| = 1.0, 0.95, ..., 0.983504 @ n=10, -> sqrt(0.95)=0.974679
    dc = oracle.CDownConvert()
    dc.SetDataRate(2e6, 1e9)           # bandwidth so large that no decimation stage is built
    assert dc.stages() == []
    dc.SetFrequency(12345.0)
    y = dc.ProcessData(np.ones(2000, dtype=np.complex128))
    mag = np.abs(y)
    g = GOLDEN["nco_envelope"]["expect"]
    assert mag[0] == pytest.approx(g["mag_0"], abs=1e-12)
    assert mag[1] == pytest.approx(g["mag_1"], abs=1e-12)
    assert mag[10] == pytest.approx(g["mag_10"], abs=g["tol_10"])
    assert mag[-1] == pytest.approx(np.sqrt(0.95), abs=1e-9)
    assert mag[-1] == pytest.approx(g["mag_inf"], abs=g["tol_10"])
    # first sample is already rotated by one increment, phase continuous afterwards
    inc = 2 * np.pi * 12345.0 / 2e6
    np.testing.assert_allclose(np.unwrap(np.angle(y)), inc * (np.arange(2000) + 1), atol=1e-9)


def test_cw_offset_double_add_anchor(oracle):
    # section 8c: SetCwOffset(700); SetFrequency(1000) -> 1700; SetDataRate(2e6,15000) -> 2400,
    # 5 stages, 62500 out
    dc = oracle.CDownConvert()
    dc.SetCwOffset(700)
    dc.SetFrequency(1000)
    assert dc.nco_freq() == 1700
    rate = dc.SetDataRate(2e6, 15000)
    assert dc.nco_freq() == 2400
    assert rate == 62500
    assert dc.stages() == [11, 11, 15, 19, 31]


@pytest.mark.parametrize("in_rate,bw,chain,out", [
    (c["in_rate"], c["max_bw"], c["stages"], c["out_rate"]) for c in GOLDEN["decimator_chains"]["cases"]
])   # FM, AM/SAM, USB/LSB, CW, 10 MSPS FM
def test_decimator_chains_anchor(oracle, in_rate, bw, chain, out):
    # SURVEY App. A.3 chains (computed from downconvert.cpp:127-166 + filtercoef.h:17-28)
    dc = oracle.CDownConvert()
    assert dc.SetDataRate(in_rate, bw) == out
    assert dc.stages() == chain


def test_hb11_equals_direct_formula_anchor(oracle):
    # App. A.9: HB11 unrolled == y[j] = sum_k h[k] xext[2j+k] over 5 consecutive 256-sample calls
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "csdr_hb_taps.h")).read()
    ev = [float(v) for v in re.search(r"/\* HB11 \*/ \{([^}]*)\}", hdr).group(1).split(",")][:3]
    h = np.zeros(11); h[0:5:2] = ev; h[5] = 0.5; h[6:11:2] = ev[::-1]
    dc = oracle.CDownConvert()
    dc.SetDataRate(100000.0, 100000.0 * 0.025 / 2 * 1.01)   # exactly one HB11 stage
    assert dc.stages()[0] == 11
    n_stage = len(dc.stages())
    dc.SetFrequency(0.0)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(5 * 256) + 1j * rng.standard_normal(5 * 256)
    got = np.concatenate([dc.ProcessData(x[i * 256:(i + 1) * 256]) for i in range(5)])
    # undo the NCO amplitude (freq 0 => pure real gain sequence), then compare stage 1 only
    if n_stage == 1:
        a = np.empty(len(x)); g = 1.0
        for i in range(len(x)):
            a[i] = g; g = g * (1.95 - g * g)
        xe = np.concatenate([np.zeros(10), x * a])
        ref = np.array([np.dot(h, xe[2 * j:2 * j + 11]) for j in range(len(x) // 2)])
        np.testing.assert_allclose(got, ref, atol=2e-15 * 4)


def test_resampler_anchor(oracle):
    # App. A.9: Rate=1.0 -> out[i] = in[i-14] to 2.2e-16; 4096 in @ 1.6276 -> 2517 out
    r = oracle.CFractResampler(); r.Init(8192)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(4096)
    y = r.Resample(x, 1.0)
    assert len(y) == 4096
    np.testing.assert_allclose(y[14:], x[:-14], atol=1e-15)
    r2 = oracle.CFractResampler(); r2.Init(8192)
    assert len(r2.Resample(x, 1.6276)) == 2517


def test_display_fft_anchor(oracle):
    # section 8c C1 anchor: -20 dBFS complex tone at +250 kHz, Fs 2 MHz, N 4096, ave 1, dBcomp 0
    # peaks at display index 2560 with -1.3982 bels; screen call returns pixel 2560 as min y
    n, fs = 4096, 2e6
    f = oracle.CFft()
    f.SetFFTParams(n, False, 0.0, fs)
    f.SetFFTAve(1)
    t = np.arange(n)
    x = 3276.7 * np.exp(2j * np.pi * 250e3 * t / fs)
    f.PutInDisplayFFT(x)
    ave = f.ave_buf()
    g = GOLDEN["display_fft_c1"]["expect"]
    assert np.argmax(ave) == g["peak_index"] == 2560
    assert ave[2560] == pytest.approx(g["peak_bels"], abs=g["tol_bels"])
    ov, pix = f.GetScreenIntegerFFTData(1 << 20, n - 1, 0.0, -220.0, -1000000, 1000000)
    assert not ov
    assert np.argmin(pix) == 2560


def test_waterfall_palette_and_line(oracle):
    """CPlotter's palette as its constructor writes it (gui/plotter.cpp:67-83: the six ramps' end points) and the
    waterfall line of CPlotter::draw (:425-441) = palette[255 - GetScreenIntegerFFTData(255, w, ...)]"""
    t = oracle.plotter_color_table()
    rgb = lambda v: ((int(v) >> 16) & 255, (int(v) >> 8) & 255, int(v) & 255)
    assert rgb(t[0]) == (0, 0, 0) and rgb(t[42]) == (0, 0, 255 * 42 // 43)
    assert rgb(t[43]) == (0, 0, 255) and rgb(t[86]) == (0, 255 * 43 // 43, 255)
    assert rgb(t[87]) == (0, 255, 255) and rgb(t[119]) == (0, 255, 0)
    assert rgb(t[120]) == (0, 255, 0) and rgb(t[153]) == (255, 255, 0)
    assert rgb(t[154]) == (255, 255, 0) and rgb(t[216]) == (255, 0, 0)
    assert rgb(t[217]) == (255, 0, 0) and rgb(t[255]) == (255, 0, 128)
    assert (t >> 24 == 0xff).all()
    n, fs = 4096, 2e6
    f = oracle.CFft()
    f.SetFFTParams(n, False, 0.0, fs)
    f.SetFFTAve(1)
    f.PutInDisplayFFT(3276.7 * np.exp(2j * np.pi * 250e3 * np.arange(n) / fs))
    for (w, lo, hi) in ((800, -1000000, 1000000), (3000, 200000, 300000)):
        _, lev = f.GetScreenIntegerFFTData(255, w, 0.0, -160.0, lo, hi)
        _, line = f.WaterfallLine(w, 0.0, -160.0, lo, hi)
        assert np.array_equal(line, t[255 - lev])
        assert rgb(line[np.argmin(lev)])[0] == 255 or rgb(line[np.argmin(lev)])[1] == 255   # the carrier is the hottest pixel


def test_fm_chain_rate_and_smeter_anchor(oracle):
    # section 8c: 2 MSPS / FM -> output rate 62500; 10000-amplitude carrier -> S-meter -5.55 dB
    d = oracle.CDemodulator(2048)
    d.SetInputSampleRate(2e6)
    d.SetDemod(oracle.DEMOD_FM, oracle.fm_defaults())
    d.SetDemodFreq(-100e3)
    assert d.GetOutputRate() == 62500
    assert d.buf_limit() == 19968
    n = 1 << 20
    t = np.arange(n)
    x = 10000.0 * np.exp(2j * np.pi * 100e3 * t / 2e6)
    for i in range(0, n, 1 << 16):
        d.ProcessData(x[i:i + (1 << 16)])
    # 20log10(10000*0.974679/32767) + 5 = -5.53; the survey quotes -5.55 (filter ripple)
    g = GOLDEN["fm_chain"]["expect"]
    assert d.GetSMeterAve() == pytest.approx(g["smeter_ave_db"], abs=g["tol_db"])


def test_fastfir_16384_delay_anchor(oracle):
    # section 6 / A.9: 16384/8193 FastFIR: output = input delayed by (P-1)/2 = 4096 samples,
    # stop-band tone removed to ~7e-5 of pass-band amplitude
    n = 16384
    ff = oracle.CFastFIR(n)
    fs = 62500.0
    assert ff.SetupParameters(-5000, 5000, 0, fs) == 1
    t = np.arange(n * 6)
    x = np.exp(2j * np.pi * 1000.0 * t / fs) + np.exp(2j * np.pi * 20000.0 * t / fs)
    y = ff.ProcessData(x)
    assert len(y) == (len(x) // (n // 2)) * (n // 2)
    want = np.exp(2j * np.pi * 1000.0 * (t[:len(y)] - 4096) / fs)
    err = np.abs(y[n:] - want[n:]).max()
    assert err < 5e-4


def test_oracle_has_not_drifted():
    """tests/golden/oracle_regression.json holds digests of the oracle's own outputs on seeded inputs
    (written by tests/golden/make_oracle_regression.py): the checker must not change silently."""
    import importlib.util
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_oracle_regression", os.path.join(here, "golden", "make_oracle_regression.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "oracle_regression.json")))
    got = mod.build()
    assert got.keys() == want.keys()
    for k in want:
        for f in want[k]:
            assert got[k][f] == pytest.approx(want[k][f], rel=1e-9, abs=1e-6), (k, f)
