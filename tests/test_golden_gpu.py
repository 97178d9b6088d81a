"""The reference-derived known answers of tests/golden/survey_anchors.json (SURVEY.md section 8c /
App. A.9) replayed directly against the HIP path through the C ABI -- no oracle in between.
Tolerances are the fixture's own, widened only where fp32 device arithmetic needs it (stated)."""
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_anchors.json")))


def test_nco_envelope():
    import cutesdr_amd as ca
    g = GOLDEN["nco_envelope"]
    dc = ca.CDownConvert()
    dc.SetDataRate(g["input"]["in_rate"], g["input"]["max_bw"])
    assert dc.stages() == []
    dc.SetFrequency(g["input"]["nco_hz"])
    mag = np.abs(dc.ProcessData(np.ones(2000, dtype=np.complex128)))
    e = g["expect"]
    tol = 2e-6                                               # fp32 phasor and amplitude table
    assert mag[0] == pytest.approx(e["mag_0"], abs=tol)
    assert mag[1] == pytest.approx(e["mag_1"], abs=tol)
    assert mag[10] == pytest.approx(e["mag_10"], abs=tol)
    assert mag[-1] == pytest.approx(e["mag_inf"], abs=tol)


def test_cw_offset_double_add_and_chains():
    import cutesdr_amd as ca
    g = GOLDEN["cw_offset_double_add"]
    dc = ca.CDownConvert()
    dc.SetCwOffset(g["input"]["cw_offset"])
    dc.SetFrequency(g["input"]["frequency"])
    assert dc.nco_freq() == g["expect"]["nco_after_set_frequency"]
    assert dc.SetDataRate(g["input"]["in_rate"], g["input"]["max_bw"]) == g["expect"]["out_rate"]
    assert dc.nco_freq() == g["expect"]["nco_after_set_data_rate"]
    assert dc.stages() == g["expect"]["stages"]
    for c in GOLDEN["decimator_chains"]["cases"]:
        d = ca.CDownConvert()
        assert d.SetDataRate(c["in_rate"], c["max_bw"]) == c["out_rate"]
        assert d.stages() == c["stages"]


def test_resampler_delay_and_count():
    import cutesdr_amd as ca
    e = GOLDEN["resampler"]["expect"]
    rng = np.random.default_rng(5)
    x = rng.standard_normal(e["in_count"])
    r = ca.CFractResampler(); r.Init(8192)
    y = r.Resample(x, 1.0)
    assert len(y) == e["in_count"]
    d = e["unity_rate_delay"]
    np.testing.assert_allclose(y[d:], x[:-d], atol=1e-5 * np.abs(x).max())    # fp32 table and samples
    r2 = ca.CFractResampler(); r2.Init(8192)
    assert len(r2.Resample(x, e["rate"])) == e["out_count"]


def test_display_peak():
    import cutesdr_amd as ca
    g = GOLDEN["display_fft_c1"]
    i, e = g["input"], g["expect"]
    f = ca.CFft(); f.SetFFTParams(i["n"], False, i["db_comp"], i["fs"]); f.SetFFTAve(i["ave"])
    f.PutInDisplayFFT(i["tone_amplitude"] * np.exp(2j * np.pi * i["tone_hz"] * np.arange(i["n"]) / i["fs"]))
    a = f.ave_buf()
    assert np.argmax(a) == e["peak_index"]
    assert a[e["peak_index"]] == pytest.approx(e["peak_bels"], abs=e["tol_bels"])


def test_fm_chain_rate_limit_and_smeter():
    import cutesdr_amd as ca
    g = GOLDEN["fm_chain"]
    i, e = g["input"], g["expect"]
    d = ca.CDemodulator(i["fastfir_n"])
    d.SetInputSampleRate(i["in_rate"])
    d.SetDemod(ca.DEMOD_FM, ca.fm_defaults())
    d.SetDemodFreq(i["demod_freq"])
    assert d.GetOutputRate() == e["out_rate"]
    assert d.buf_limit() == e["in_buf_limit"]
    n = i["samples"]
    x = i["carrier_amplitude"] * np.exp(2j * np.pi * i["carrier_hz"] * np.arange(n) / i["in_rate"])
    for k in range(0, n, 1 << 16):
        d.ProcessData(x[k:k + (1 << 16)])
    assert d.GetSMeterAve() == pytest.approx(e["smeter_ave_db"], abs=e["tol_db"])


def test_fastfir_16384_delay():
    import cutesdr_amd as ca
    g = GOLDEN["fastfir_16384"]
    i, e = g["input"], g["expect"]
    n, fs = i["n"], i["fs"]
    ff = ca.CFastFIR(n)
    ff.SetupParameters(i["lo_cut"], i["hi_cut"], 0, fs)
    t = np.arange(n * 6)
    x = np.exp(2j * np.pi * i["pass_tone_hz"] * t / fs) + np.exp(2j * np.pi * i["stop_tone_hz"] * t / fs)
    y = ff.ProcessData(x)
    assert len(y) == (len(x) // (n // 2)) * (n // 2)
    want = np.exp(2j * np.pi * i["pass_tone_hz"] * (t[:len(y)] - e["delay_samples"]) / fs)
    assert np.abs(y[n:] - want[n:]).max() < e["max_err"]
