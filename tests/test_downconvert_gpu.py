"""GPU parity of the NCO + decimator cascade kernel (K2) against the fp64 oracle, through the
C ABI.  Tolerance: |err| <= 1e-5 * full scale (32767) per output sample (K2),
sample counts exact."""
import numpy as np
import pytest
from util_signals import tones_plus_noise, FULL_SCALE

pytestmark = pytest.mark.gpu
TOL = 1e-5 * FULL_SCALE


def run_oracle(oracle, in_rate, bw, freq, chunks, cw=0.0):
    dc = oracle.CDownConvert()
    dc.SetCwOffset(cw)
    dc.SetDataRate(in_rate, bw)
    dc.SetFrequency(freq)
    return dc, [dc.ProcessData(c) for c in chunks]


@pytest.mark.parametrize("in_rate,bw,chain,out_rate", [
    (2e6, 15000, [11, 11, 15, 19, 31], 62500),
    (2e6, 10000, [11, 11, 11, 15, 23, 51], 31250),
    (2e6, 20000, [11, 11, 15, 23, 51], 62500),
    (2e6, 1000, [3, 3, 11, 11, 11, 11, 15], 15625),
    (10e6, 15000, [3, 11, 11, 11, 11, 15, 27], 78125),
])
def test_host_cdownconvert_matches_oracle(oracle, in_rate, bw, chain, out_rate):
    import cutesdr_amd as ca
    dc = ca.CDownConvert()
    assert dc.SetDataRate(in_rate, bw) == out_rate
    assert dc.stages() == chain
    dc.SetFrequency(-100e3)
    n_call = 19968 if in_rate == 2e6 else 99840          # m_InBufLimit windows (demodulator.cpp:145-146)
    x = tones_plus_noise(3, 5 * n_call, in_rate, [100e3 + 1000.0, 100e3 - 4000.0, 380e3])
    chunks = [x[i * n_call:(i + 1) * n_call] for i in range(5)]
    _, refs = run_oracle(oracle, in_rate, bw, -100e3, chunks)
    for c, ref in zip(chunks, refs):
        got = dc.ProcessData(c)
        assert len(got) == len(ref) == n_call >> len(chain)
        assert np.abs(got - ref).max() <= TOL


def test_nco_startup_envelope_and_cw_double_add(oracle):
    import cutesdr_amd as ca
    dc = ca.CDownConvert()
    dc.SetCwOffset(700)
    dc.SetFrequency(1000)
    assert dc.nco_freq() == 1700
    assert dc.SetDataRate(2e6, 1e9) == 2e6          # no stage fits: pure mixer
    assert dc.stages() == []
    assert dc.nco_freq() == 2400                    # SetDataRate re-adds the CW offset (App. A.2)
    ref = oracle.CDownConvert()
    ref.SetCwOffset(700); ref.SetFrequency(1000); ref.SetDataRate(2e6, 1e9)
    x = np.full(4096, 20000.0 + 0j)
    for _ in range(3):                              # phase and amplitude continue across calls
        got, want = dc.ProcessData(x), ref.ProcessData(x)
        assert len(got) == 4096
        assert np.abs(got - want).max() <= 20000.0 * 4e-6
    # retune keeps the phasor (phase continuous), only the increment changes
    dc.SetFrequency(-250e3); ref.SetFrequency(-250e3)
    got, want = dc.ProcessData(x), ref.ProcessData(x)
    assert np.abs(got - want).max() <= 20000.0 * 4e-6


def test_small_calls_are_chunking_independent(oracle):
    """Calls shorter than the cascade's warm-up length (and shorter than the reference can take:
    below 2*(taps-1) samples at a stage its in-place history copy reads overwritten data,
    downconvert.cpp:314-317, and below `taps` it skips the stage, :291-292 -- the host never goes
    there, it feeds 19968 samples per call).  Here the stream semantics simply continue: the
    output must not depend on how the input is chunked, and must match the oracle fed in
    host-sized chunks."""
    import cutesdr_amd as ca
    x = tones_plus_noise(5, 19968 * 2, 2e6, [-50e3 + 2000.0, 300e3])
    ref = oracle.CDownConvert(); ref.SetDataRate(2e6, 15000); ref.SetFrequency(50e3)
    want = np.concatenate([ref.ProcessData(x[:19968]), ref.ProcessData(x[19968:])])
    for chunk in (256, 512, 4992):
        dc = ca.CDownConvert(); dc.SetDataRate(2e6, 15000); dc.SetFrequency(50e3)
        got = np.concatenate([dc.ProcessData(x[i:i + chunk]) for i in range(0, len(x), chunk)])
        assert len(got) == len(want)
        assert np.abs(got - want).max() <= TOL, chunk


def test_batch_mixed_chains_segments_and_state(oracle):
    import cutesdr_amd as ca
    C, T = 6, 1 << 19                                # long enough to be cut into segments
    setups = [(15000, -100e3), (10000, 40e3), (20000, -333e3), (1000, 7e3), (15000, 0.0), (10000, -1e3)]
    b = ca.DownConvertBatch(C)
    refs = []
    for c, (bw, f) in enumerate(setups):
        assert b.set_data_rate(2e6, bw, channel=c) > 0
        b.set_frequency(f, channel=c)
        r = oracle.CDownConvert(); r.SetDataRate(2e6, bw); r.SetFrequency(f)
        refs.append(r)
    x = np.stack([tones_plus_noise(20 + c, 2 * T, 2e6, [-setups[c][1] + 700.0, 450e3]) for c in range(C)])
    for part in (x[:, :T], x[:, T:]):
        got = b.process(part)
        for c in range(C):
            # the oracle, like the reference, has a 32768-sample half-band scratch buffer
            # (MAX_HALF_BAND_BUFSIZE, downconvert.cpp:54): feed it in 65536-sample pieces
            want = np.concatenate([refs[c].ProcessData(part[c][i:i + 65536]) for i in range(0, T, 65536)])
            assert len(got[c]) == len(want)
            assert np.abs(got[c] - want).max() <= TOL, c


def test_precompiled_plans_reproduce_the_runtime_plan_kernel(oracle):
    """The kernel compiled for a stage sequence (DC_PLANS in cutesdr_amd/_build.py, the reference's radio rates x
    demodulator bandwidths: tools/list_dc_plans.py) against the same kernel with the plan taken at run time: the
    same words out, ragged call lengths included (short tiles take the generic stage code in both), and both within
    the tolerance of the oracle."""
    import ctypes as C
    import cutesdr_amd as ca
    from cutesdr_amd import _build
    L = ca.lib()
    L.csdr__downconv_force_dynamic.restype = C.c_int
    L.csdr__downconv_force_dynamic.argtypes = [C.c_int]
    assert L.csdr__downconv_force_dynamic(-1) == len(_build.DC_PLANS)
    cases = [(2e6, 15000, [11, 11, 15, 19, 31]), (2e6, 1000, [3, 3, 11, 11, 11, 11, 15]), (62500, 10000, [51]),
             (250000, 20000, [23, 51]), (625000, 15000, [11, 15, 27]), (80e6 / 130, 10000, [11, 15, 19, 35])]
    try:
        for in_rate, bw, chain in cases:
            assert tuple(chain) in _build.DC_PLANS
            # (every call long enough that no stage sees fewer samples than its taps: the reference's early return for
            # such calls is a documented deviation)
            # and none longer than the reference's half-band scratch buffer
            calls = [16384, 8192 + 640, 4096 + (3 << len(chain)), 20000 - 20000 % (1 << len(chain))]
            x = tones_plus_noise(11, sum(calls), in_rate, [in_rate * 0.05 + 300.0, in_rate * 0.05 - 900.0, in_rate * 0.19])
            outs = []
            for dyn in (0, 1):
                L.csdr__downconv_force_dynamic(dyn)
                dc = ca.CDownConvert()
                dc.SetDataRate(in_rate, bw)
                assert dc.stages() == chain
                dc.SetFrequency(-in_rate * 0.05)
                pos, got = 0, []
                for n in calls:
                    got.append(dc.ProcessData(x[pos:pos + n])); pos += n
                outs.append(np.concatenate(got))
            assert np.array_equal(outs[0], outs[1])
            _, refs = run_oracle(oracle, in_rate, bw, -in_rate * 0.05, np.split(x, np.cumsum(calls)[:-1]))
            assert np.abs(outs[0] - np.concatenate(refs)).max() <= TOL
    finally:
        L.csdr__downconv_force_dynamic(0)


def test_every_precompiled_plan_runs_and_matches_the_runtime_plan_kernel():
    """All of DC_PLANS -- the plans behind the usual front-end rates by default, every stage sequence SetDataRate can
    produce in a CSDR_ALL_DC_PLANS=1 build -- each at a (rate, bandwidth) pair that selects it: the plan-compiled
    kernel and the run-time-plan kernel give the same words on a three-call stream with a ragged middle call."""
    import ctypes as C
    import cutesdr_amd as ca
    from cutesdr_amd import _build
    pairs = _build.DC_PLAN_PAIRS
    L = ca.lib()
    L.csdr__downconv_force_dynamic.restype = C.c_int
    L.csdr__downconv_force_dynamic.argtypes = [C.c_int]
    assert L.csdr__downconv_force_dynamic(-1) == len(pairs) == len(_build.DC_PLANS)
    try:
        for plan in _build.DC_PLANS:
            in_rate, bw = pairs[plan]
            unit = 1 << len(plan)
            calls = [8192, 2048 + 3 * unit, 4096]
            x = tones_plus_noise(5, sum(calls), in_rate, [in_rate * 0.07 + 200.0, in_rate * 0.07 - 700.0, in_rate * 0.21])
            outs = []
            for dyn in (0, 1):
                L.csdr__downconv_force_dynamic(dyn)
                dc = ca.CDownConvert()
                dc.SetDataRate(in_rate, bw)
                assert tuple(dc.stages()) == plan, (in_rate, bw)
                dc.SetFrequency(-in_rate * 0.07)
                pos, got = 0, []
                for n in calls:
                    got.append(dc.ProcessData(x[pos:pos + n])); pos += n
                outs.append(np.concatenate(got))
            assert len(outs[0]) == sum(calls) >> len(plan)
            assert np.array_equal(outs[0], outs[1]), plan
    finally:
        L.csdr__downconv_force_dynamic(0)
