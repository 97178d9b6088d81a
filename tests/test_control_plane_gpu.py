"""The control plane without device-wide synchronisation (VERDICT r5 task 6).

The reference's setters are a mutex and a few stores (dsp/demodulator.cpp:107-157, demodulator.h:68-69: the GUI calls
SetDemodFreq on every mouse move) and take effect with the next ProcessData.  Here set_freq, a same-mode set_demod
(filter edges, AGC constants, squelch, AM bandwidth) and the spectrum's readers queue patches / wait on the object's
own stream (cutesdr_amd/csrc/patch_queue.hpp): they must (1) give the oracle's audio when they arrive in the middle of
a PIPELINED stream that is never flushed, (2) not wait for other work on the device, (3) leave every word alone when
they set what is already set."""
import time
import numpy as np
import pytest
from util_signals import tones_plus_noise, fm_carrier, am_carrier, FULL_SCALE
from test_postchain_gpu import MODES, info, burst_errors, check_chain_bursts

pytestmark = pytest.mark.gpu


def _streams(names, n, fs, f1, f2):
    xs = []
    for c, name in enumerate(names):
        if name == "FM":
            x = fm_carrier(n, fs, f1[c], dbfs=-20.0, channel=c) + fm_carrier(n, fs, f2[c], fmod=700.0, dbfs=-26.0, noise_dbfs=-200.0, channel=c + 10)
        elif name in ("AM", "SAM"):
            x = am_carrier(n, fs, f1[c], dbfs=-20.0, channel=c) + am_carrier(n, fs, f2[c], fmod=600.0, dbfs=-26.0, noise_dbfs=-200.0, channel=c + 10)
        else:
            x = tones_plus_noise(9, n, fs, [f1[c] + 1200.0, f1[c] + 2340.0, f2[c] + 900.0, f2[c] + 1710.0])
        xs.append(x.astype(np.complex64))
    return np.stack(xs)


@pytest.mark.parametrize("pipelined", [True, False], ids=["pipelined", "strict"])
def test_retune_and_parameter_changes_in_the_middle_of_a_stream_that_is_never_flushed(oracle, pipelined):
    """Six calls enqueued back to back on one stream, nothing waited for in between; between the calls: retunes, new
    filter edges + AGC decay (same mode), a new squelch value, an AM bandwidth change, a retune back.  Every receiver's
    whole audio stream against an oracle CDemodulator given the same calls at the same window boundaries."""
    import cutesdr_amd as ca
    fs, lim = 2e6, 19968
    names = ["AM", "FM", "USB", "SAM", "FM", "USB"]
    C, calls, n = len(names), 6, lim * 8
    f1 = [100e3 + 1e3 * c for c in range(C)]
    f2 = [-300e3 - 1e3 * c for c in range(C)]
    xs = _streams(names, n * calls, fs, f1, f2)
    b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
    refs = []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        b.set_demod(c, m, info(ca, **kw))
        r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-f1[c])
        refs.append(r)
    b.commit()
    for c in range(C):
        b.set_freq(c, -f1[c])
    if pipelined:
        b.set_pipelined(True)

    def control(call):
        """the control calls in front of `call`, on both sides"""
        def demod(c, **over):
            m, kw = MODES[names[c]]
            kw = dict(kw, **over)
            b.set_demod(c, m, info(ca, **kw)); refs[c].SetDemod(m, info(oracle, **kw))
        if call == 1:
            for c in (0, 1, 2):
                b.set_freq(c, -f2[c]); refs[c].SetDemodFreq(-f2[c])
        if call == 2:
            demod(1, HiCut=3500, LowCut=-3500, AgcDecay=500)          # FM: filter edges (the squelch high-pass follows HiCut)
            demod(2, HiCut=2400, LowCut=300, AgcThresh=-80)            # USB: edges + AGC knee
            demod(3, HiCut=4000, LowCut=-4000)                         # SAM: edges
            demod(4, SquelchValue=20)                                  # FM: squelch threshold only
        if call == 3:
            demod(0, HiCut=3000, LowCut=-3000)                         # AM: edges = a new audio low-pass with a cleared delay line
            for c in (3, 4, 5):
                b.set_freq(c, -f2[c]); refs[c].SetDemodFreq(-f2[c])
        if call == 4:
            for c in range(C):
                b.set_freq(c, -f1[c]); refs[c].SetDemodFreq(-f1[c])
            demod(1, HiCut=5000, LowCut=-5000, AgcDecay=200)
    din = ca.DeviceBuffer(xs.nbytes)
    din.upload(xs)
    cap = n // 32 + 2048 + 4096
    douts = [ca.DeviceBuffer(C * cap * 4) for _ in range(calls)]
    counts, want = [], [[] for _ in range(C)]
    t0 = time.perf_counter()
    for k in range(calls):
        control(k)
        b.process_ptr(din.ptr + k * n * 8, n * calls, n, douts[k].ptr, cap)       # row stride = the whole stream
        counts.append([b.out_count(c) for c in range(C)])
    host_s = time.perf_counter() - t0
    b.flush()
    ca.sync()
    # the oracle side: fresh chains, the same control calls in front of the same windows
    refs2 = []
    for c, name in enumerate(names):
        m, kw = MODES[name]
        r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, info(oracle, **kw)); r.SetDemodFreq(-f1[c])
        refs2.append(r)
    refs[:] = refs2
    b2 = b
    class _Null:                                                     # the control() helper drives both sides: mute the GPU side now
        def set_demod(self, *a): pass
        def set_freq(self, *a): pass
    b = _Null()
    got = [[] for _ in range(C)]
    since = [0] * C                                                  # bursts since the receiver's last control call
    for k in range(calls):
        touched = {1: (0, 1, 2), 2: (1, 2, 3, 4), 3: (0, 3, 4, 5), 4: tuple(range(C))}.get(k, ())
        control(k)
        out = douts[k].download(np.float32, C * cap).reshape(C, cap)
        for c, name in enumerate(names):
            w = refs[c].process_append(xs[c, k * n:(k + 1) * n].astype(np.complex128))
            g = out[c, :counts[k][c]].astype(np.float64)
            assert len(g) == len(w) > 0, (k, c, name)
            errs = burst_errors(g, w)
            what = (k, c, name, "pipelined" if pipelined else "strict")
            if k == 0:
                check_chain_bursts(errs, name if name in ("FM", "SAM") else "other", 0, what)
            else:
                if c in touched:
                    since[c] = 0
                # behind a control call both sides ring down and settle again from the SAME states: the bounds of
                # test_retune_between_calls_in_the_batch_chain (eight bursts of the start-up bound, then steady)
                early = (1e-3 if name == "FM" else 5e-4) * FULL_SCALE
                steady = (3e-5 if name == "FM" else 2e-5) * FULL_SCALE
                idx = since[c] + np.arange(len(errs))
                assert (errs[idx < 8] <= early).all(), (what, errs[:10] / FULL_SCALE)
                assert (errs[idx >= 8] <= steady).all(), (what, errs[:16] / FULL_SCALE)
            since[c] += len(errs)
    assert host_s < 5.0
    del b2


def test_setters_do_not_wait_for_other_work_on_the_device():
    """A long queue of unrelated launches (300 x the 256-channel 16384-point filter: > 150 ms) is in flight on ANOTHER
    stream; 64 retunes + 64 same-mode SetDemod calls (new filter edges, AGC constants, squelch) on a committed batch and
    the next process call's enqueue must return while that queue is still running -- with a hipDeviceSynchronize anywhere
    on the way they would take as long as the queue.  (Streams through the HIP runtime the library itself is linked to:
    a second runtime in the process -- torch's -- cannot open the device behind it.)"""
    import ctypes as C
    import cutesdr_amd as ca
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamQuery.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    Cn, T = 256, 1 << 19
    rng = np.random.default_rng(3)
    x = ca.DeviceBuffer(Cn * T * 8); y = ca.DeviceBuffer(Cn * T * 8)
    row = (rng.standard_normal(2 * T) * 3000.0).astype(np.float32)
    for c in range(Cn):
        x.upload(row, c * T * 8)
    ff = ca.FastFirBatch(Cn, 16384); ff.setup(-5000, 5000, 0, 62500.0)
    C_, n = 64, 19968 * 4
    names = ["AM", "FM", "USB", "FM"]
    b = ca.DemodBatch(C_, 2048); b.set_input_rate(2e6)
    for c in range(C_):
        m, kw = MODES[names[c % 4]]
        b.set_demod(c, m, info(ca, **kw))
    b.commit()
    xin = ca.DeviceBuffer(C_ * n * 8)
    xrow = (rng.standard_normal(2 * n) * 3000.0).astype(np.float32)
    for c in range(C_):
        xin.upload(xrow, c * n * 8)
    cap = n // 16 + 4096
    aud = ca.DeviceBuffer(C_ * cap * 4)
    side, main = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(side), 1) == 0 and hip.hipStreamCreateWithFlags(C.byref(main), 1) == 0
    try:
        b.process_ptr(xin.ptr, n, n, aud.ptr, cap, main.value)            # warm: allocations, first launches
        ff.process_ptr(x.ptr, T, T, y.ptr, T, side.value)
        ca.sync()
        for _ in range(300):
            ff.process_ptr(x.ptr, T, T, y.ptr, T, side.value)
        t0 = time.perf_counter()
        for c in range(C_):
            b.set_freq(c, -100e3 - 100.0 * c)
            m, kw = MODES[names[c % 4]]
            kw = dict(kw, HiCut=kw.get("HiCut", 5000) - 200, AgcDecay=300, SquelchValue=10)
            b.set_demod(c, m, info(ca, **kw))
        b.process_ptr(xin.ptr, n, n, aud.ptr, cap, main.value)
        host_ms = (time.perf_counter() - t0) * 1e3
        still_running = hip.hipStreamQuery(side) != 0                     # hipErrorNotReady: the unrelated queue is not done
        ca.sync()
    finally:
        ca.sync()
        hip.hipStreamDestroy(side); hip.hipStreamDestroy(main)
    assert still_running, "the side queue had already drained: the measurement says nothing (host %.1f ms)" % host_ms
    assert host_ms < 80.0, host_ms
    out = aud.download(np.float32, C_ * cap)
    assert np.isfinite(out).all()


def test_setting_what_is_already_set_changes_no_word(oracle):
    """set_freq to the frequency a receiver already has and a same-mode SetDemod with the same parameters, between
    calls: the patches carry the host mirror's phase and age of the oscillator and the parameter words -- if the mirror
    were off by one sample anywhere, the audio would show it.  FM and USB receivers (an AM SetDemod clears its low-pass,
    as the reference's does: not a no-op); word for word against a batch nobody touches."""
    import cutesdr_amd as ca
    fs, lim = 2e6, 19968
    names = ["FM", "USB", "FM", "USB"]
    C, calls, n = len(names), 4, lim * 8
    xs = _streams(names, n * calls, fs, [100e3 + 1e3 * c for c in range(C)], [-300e3] * C)
    outs = []
    for touch in (False, True):
        b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
        for c, name in enumerate(names):
            m, kw = MODES[name]
            b.set_demod(c, m, info(ca, **kw))
        b.commit()
        for c in range(C):
            b.set_freq(c, -100e3 - 1e3 * c)
        b.set_pipelined(True)
        res = []
        for k in range(calls):
            if touch and k > 0:
                for c, name in enumerate(names):
                    b.set_freq(c, -100e3 - 1e3 * c)
                    m, kw = MODES[name]
                    b.set_demod(c, m, info(ca, **kw))
            res.append(b.process(xs[:, k * n:(k + 1) * n]))
        outs.append(res)
        del b
    for k in range(calls):
        for c in range(C):
            assert np.array_equal(outs[0][k][c], outs[1][k][c]), (k, c, names[c])
