"""The chain's test points PROFILE_1..4 (dsp/demodulator.cpp:175,180,187,208: what the reference hands to
g_pTestBench->DisplayData after the down-converter, the filter, the AGC and the demodulator of every pass) on the HIP
path against the oracle's own taps, pass by pass: a difference in the audio now NAMES ITS STAGE -- tap 1 under the
down-converter's tolerance (K2: 1e-5 of full scale), tap 2 under the filter's (K1: 2e-5 of the largest filter input),
tap 3 under the post-chain's (K4, the chain rule of test_postchain_gpu.py), tap 4 = the audio the other tests compare.
Switching the taps on must not change a single word of the audio."""
import numpy as np
import pytest
from util_signals import FULL_SCALE
from test_postchain_gpu import MODES, info, make_input, burst_errors, check_chain_bursts, STEADY, FROM_ZERO
from test_chain_parity_gpu import pair, chain_input

pytestmark = pytest.mark.gpu

K2_TOL = 1e-5 * FULL_SCALE              # down-converter
K1_REL = 2e-5                           # filter: of the largest sample that has entered it
AGC_FIRST = 6e-4 * FULL_SCALE           # tap 3, the stream's first burst: the AGC at full gain (x 7e4 below the knee) on the
                                        # filter's start-up, whose fp32 error floor is absolute: 2 x the 3e-4 the oracle does to
                                        # itself there (tests/startup_bounds.py: LINEAR_SPREAD; measured 1.3e-7 ... 1.4e-4)
AGC_SECOND = STEADY


@pytest.mark.parametrize("stereo", [False, True], ids=["mono", "stereo"])
@pytest.mark.parametrize("mode", ["AM", "SAM", "FM", "USB", "CWU"])
def test_every_stage_of_the_chain_against_the_oracles_taps(oracle, mode, stereo):
    import cutesdr_amd as ca
    d, r = pair(ca, oracle, mode)
    plain, _ = pair(ca, oracle, mode)                   # the same chain with the taps off
    d.enable_taps(15)
    r.enable_taps(True)
    lim = d.buf_limit()
    x = chain_input(mode, lim * (64 if mode == "CWU" else 24), 2e6)
    hop = 1024
    seen_in = 0.0
    e1, e2, e3, e4, in_max = [], [], [], [], []
    for i in range(0, len(x), lim):
        r.clear_taps()
        kg, og = d.ProcessData(x[i:i + lim], stereo)
        kr, orr = r.ProcessData(x[i:i + lim], stereo)
        kp, op = plain.ProcessData(x[i:i + lim], stereo)
        assert kg == kr == kp
        assert np.array_equal(og[:kg], op[:kp]), "switching the taps on changed the audio"
        g1, g2, g3, g4 = d.tap(1), d.tap(2), d.tap(3), d.tap(4)
        w1, w2, w3, w4 = r.tap(1), r.tap(2), r.tap(3), r.tap(4)
        assert len(g1) == len(w1) > 0 and len(g2) == len(w2) == kr and len(g3) == len(w3) == kr
        if stereo:
            g4, w4 = g4.view(np.complex128), w4.view(np.complex128)
        assert len(g4) == len(w4) == kr
        if kr:
            assert np.array_equal(g4, og[:kg].astype(g4.dtype))        # tap 4 IS the audio
        e1.append(np.abs(g1 - w1).max())
        seen_in = max(seen_in, np.abs(w1.real).max(), np.abs(w1.imag).max())
        for j in range(0, kr, hop):
            e2.append(np.abs(g2[j:j + hop] - w2[j:j + hop]).max()); in_max.append(seen_in)
            e3.append(np.abs(g3[j:j + hop] - w3[j:j + hop]).max())
            e4.append(np.abs(g4[j:j + hop] - w4[j:j + hop]).max())
    e1, e2, e3, e4, in_max = map(np.array, (e1, e2, e3, e4, in_max))
    assert len(e2) >= 6
    what = (mode, "stereo" if stereo else "mono")
    print("taps", what, "tap1 %.2e" % (e1.max() / FULL_SCALE), "tap2/in %s" % np.array2string((e2 / in_max)[:4], precision=2),
          "tap3 %s" % np.array2string(e3[:5] / FULL_SCALE, precision=2), "tap4 %s" % np.array2string(e4[:5] / FULL_SCALE, precision=2))
    assert (e1 <= K2_TOL).all(), (what, "down-converter", e1.max() / FULL_SCALE)
    assert (e2 <= K1_REL * in_max).all(), (what, "filter", (e2 / in_max).max())
    # the AGC: its first two bursts as the chain's SAM bounds (the same mechanism), then the steady bound
    assert e3[0] <= AGC_FIRST and (len(e3) < 2 or e3[1] <= AGC_SECOND), (what, "AGC start-up", e3[:4] / FULL_SCALE)
    assert (e3[2:] <= STEADY).all(), (what, "AGC", e3[:8] / FULL_SCALE)
    check_chain_bursts(e4, mode if mode in ("FM", "SAM") else "other", 0, what, stereo=stereo)


def test_tap_callback_is_called_per_pass_in_the_references_order(oracle):
    """the callback form (what a host with a test bench registers): PROFILE_1..4 in that order for every pass, n = 0 for
    taps 2..4 while the filter is still filling, the rate the reference passes (m_OutputRate)"""
    import cutesdr_amd as ca
    d, r = pair(ca, oracle, "USB")
    calls = []
    d.enable_taps(15, lambda profile, n, data, cpx, rate: calls.append((profile, n, cpx, rate, data.copy())))
    r.enable_taps(True)
    lim = d.buf_limit()
    x = chain_input("USB", lim * 8, 2e6)
    passes = 0
    for i in range(0, len(x), lim):
        r.clear_taps()
        kg, og = d.ProcessData(x[i:i + lim])
        kr, orr = r.ProcessData(x[i:i + lim])
        assert kg == kr
        mine = calls[4 * passes:4 * passes + 4]
        assert [c[0] for c in mine] == [1, 2, 3, 4]
        assert [c[1] for c in mine] == [len(r.tap(1)), kr, kr, kr]
        assert [c[2] for c in mine] == [True, True, True, False]
        assert all(c[3] == r.GetOutputRate() for c in mine)
        if kr:
            assert np.abs(mine[3][4] - orr[:kr]).max() <= FROM_ZERO
        passes += 1
    assert len(calls) == 4 * passes
    d.enable_taps(0)
    d.ProcessData(x[:lim])
    assert len(calls) == 4 * passes                        # off: nothing is called, nothing accumulates


def test_batch_taps_equal_the_single_receivers(oracle):
    """the same test points per receiver of a batch (csdr_demod_batch_set_taps): a mixed batch, three calls, every
    receiver's taps 1..3 of the LAST call against an oracle chain fed that receiver's stream"""
    import cutesdr_amd as ca
    fs, lim, calls = 2e6, 19968, 3
    names = ["FM", "AM", "USB", "SAM", "FM", "AM"]
    C = len(names)
    n = lim * 8
    x = np.stack([make_input(m, n * calls, fs) * (1.0 + 0.1 * c) for c, m in enumerate(names)])
    b = ca.DemodBatch(C, 2048)
    b.set_input_rate(fs)
    for c, name in enumerate(names):
        m, kw = MODES[name]
        b.set_demod(c, m, info(ca, **kw))
    b.commit()
    for c in range(C):
        b.set_freq(c, -100e3)
    b.set_taps(7)
    refs = []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        r = oracle.CDemodulator(2048)
        r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-100e3)
        r.enable_taps(True)
        refs.append(r)
    for k in range(calls):
        out = b.process(x[:, k * n:(k + 1) * n].astype(np.complex64))
        for c in range(C):
            r = refs[c]
            r.clear_taps()
            want = r.process_append(x[c, k * n:(k + 1) * n].astype(np.complex64).astype(np.complex128))
            assert len(out[c]) == len(want)
            g1, g2, g3 = b.tap(c, 1), b.tap(c, 2), b.tap(c, 3)
            w1, w2, w3 = r.tap(1), r.tap(2), r.tap(3)
            assert len(g1) == len(w1) and len(g2) == len(w2) == len(want) and len(g3) == len(w3)
            assert np.abs(g1 - w1).max() <= K2_TOL, (k, c, names[c])
            assert np.abs(g2 - w2).max() <= K1_REL * np.abs(w1).max() * 1.5, (k, c, names[c])
            if k == calls - 1:                             # steady state
                assert np.abs(g3 - w3).max() <= STEADY, (k, c, names[c], np.abs(g3 - w3).max() / FULL_SCALE)


def _oracle_post_chain(oracle, mode, stereo, fs_out, kw=None):
    """the reference's stages behind the filter as separate oracle objects, configured as CDemodulator::SetDemod does
    (dsp/demodulator.cpp:107-157): returns f(filter output of one pass) -> (AGC output, audio).  kw: the receiver's
    tDemodInfo fields (default: the mode's)"""
    if mode in ("LSB", "CWU", "CWL"):
        kw = kw if kw is not None else MODES.get(mode, MODES["USB"])[1]
    elif kw is None:
        kw = MODES[mode][1]
    di = info(oracle, **kw)
    agc = oracle.CAgc()
    agc.SetParameters(di.AgcOn, di.AgcHangOn, di.AgcThresh, di.AgcManualGain, di.AgcSlope, di.AgcDecay, fs_out)
    dem = None
    if mode == "AM":
        dem = oracle.CAmDemod(fs_out); dem.SetBandwidth((di.HiCut - di.LowCut) / 2.0)
    elif mode == "SAM":
        dem = oracle.CSamDemod(fs_out)
    elif mode == "FM":
        dem = oracle.CFmDemod(fs_out); dem.SetSquelch(di.SquelchValue)

    def run(z):
        a = agc.ProcessData(z)
        if mode == "FM":
            return a, dem.ProcessData(a, float(di.HiCut), stereo)
        if dem is not None:
            return a, dem.ProcessData(a, stereo)
        return a, oracle.ssb_demod(a, stereo)
    return run


@pytest.mark.parametrize("stereo", [False, True], ids=["mono", "stereo"])
@pytest.mark.parametrize("mode", ["AM", "SAM", "FM", "USB"])
def test_post_chain_on_the_gpus_own_filter_output_from_the_first_sample(oracle, mode, stereo):
    """The loop-only check: the start-up of a chain is where an fp32 filter and an fp64 one hand DIFFERENT
    samples to the stages behind them -- the AGC is at full gain on the filter's start-up, where the fp64 output is 1e-12
    and the fp32 output is its own rounding noise (multiples of 2^-15), and a PLL takes the PHASE of that noise: SAM stereo
    differs from the end-to-end oracle by 0.5 / 1.7 of full scale in its first two bursts on this very input
    (tools/experiments/r6_sam_zero_probe.py) although nothing is wrong.  So the stages behind the filter are checked on
    what the GPU's filter REALLY handed them: tap 2 of the GPU chain, pass by pass, through the oracle's AGC and
    demodulator objects, against taps 3 and 4 of the GPU chain -- tight, and from the first sample: no start-up allowance.
    Together with the filter's own bound (tap 2 against the oracle's, test above) this pins the whole chain."""
    import cutesdr_amd as ca
    d, r = pair(ca, oracle, mode)
    d.enable_taps(15)
    lim = d.buf_limit()
    x = chain_input(mode, lim * 24, 2e6)
    post = _oracle_post_chain(oracle, mode, stereo, r.GetOutputRate())
    e3, e4 = [], []
    for i in range(0, len(x), lim):
        kg, og = d.ProcessData(x[i:i + lim], stereo)
        g2, g3, g4 = d.tap(2), d.tap(3), d.tap(4)
        d.tap(1)
        if not kg:
            continue
        if stereo:
            g4 = g4.view(np.complex128)
        w3, w4 = post(g2)
        for j in range(0, kg, 1024):
            e3.append(np.abs(g3[j:j + 1024] - w3[j:j + 1024]).max())
            e4.append(np.abs(g4[j:j + 1024] - w4[j:j + 1024]).max())
            if mode == "FM":
                assert (not g4[j:j + 1024].any()) == (not w4[j:j + 1024].any())      # identical squelch decisions
    e3, e4 = np.array(e3), np.array(e4)
    what = (mode, "stereo" if stereo else "mono")
    print("post-chain on own filter output", what, "agc %s" % np.array2string(e3[:5] / FULL_SCALE, precision=2),
          "audio %s" % np.array2string(e4[:6] / FULL_SCALE, precision=2))
    assert (e3 <= STEADY).all(), (what, "AGC", e3[:6] / FULL_SCALE)
    if mode == "FM":
        # the overlapped walks of an unlocked tile meet to 1e-9 turns, not bit for bit (include/cutesdr_mi.h): 1e-4 while the
        # loop is pulling in on noise, the steady FM bound once it has locked
        assert (e4[:3] <= 1e-4 * FULL_SCALE).all() and (e4[3:] <= 3e-5 * FULL_SCALE).all(), (what, e4[:8] / FULL_SCALE)
    else:
        assert (e4 <= STEADY).all(), (what, "audio", e4[:6] / FULL_SCALE)
