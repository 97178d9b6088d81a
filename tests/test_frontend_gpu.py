"""GPU parity of the input-rate stages in front of the down-converter
against the fp64 oracle, through the C ABI: CNoiseProc blanker (sample-exact: zeros and delayed
copies of fp32 inputs, identical trigger decisions), wire-format unpack (bit-exact: every 16/24-bit
value is exactly representable in fp32) and the NCO-spur DC estimate (1e-9 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def impulsive(seed, n, fs, rate=2e-4):
    rng = np.random.default_rng(seed)
    x = 300.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x += 3000.0 * np.exp(2j * np.pi * 123e3 * np.arange(n) / fs)
    hits = rng.random(n) < rate
    x[hits] += 25000.0 * np.exp(2j * np.pi * rng.random(hits.sum()))
    return x.astype(np.complex64).astype(np.complex128)     # fp32-representable: both sides see the same values


@pytest.mark.parametrize("fs,thresh,width", [(2e6, 50.0, 2.0), (2e6, 20.0, 100.0), (500e3, 80.0, 3000.0), (6e6, 35.0, 10.0),
                                             (2e6, 40.0, 2040.0)])   # (4080 samples of blanking: two tiles of warm-up per segment)
def test_blanker_matches_oracle_across_calls(oracle, fs, thresh, width):
    import cutesdr_amd as ca
    g, r = ca.CNoiseProc(), oracle.CNoiseProc()
    g.SetupBlanker(True, thresh, width, fs); r.SetupBlanker(True, thresh, width, fs)
    x = impulsive(int(fs) % 1000 + int(width), 500000, fs)
    blanked = 0
    # ragged calls, history across calls; the last one long enough for several segments (each rebuilds its window sum --
    # and, where the 5 ms window fits it, the LDS ring of magnitudes -- from the samples in front of it)
    for a, b in ((0, 240), (240, 5000), (5000, 70000), (70000, 70001), (70001, 200000), (200000, 500000)):
        got, want = g.ProcessBlanker(x[a:b]), r.ProcessBlanker(x[a:b])
        assert np.array_equal(got, want), (a, b, np.nonzero(got != want)[0][:5])
        blanked += int((want == 0).sum())
    assert 0 < blanked < len(x)                               # the case actually blanks, and not everything


def test_blanker_off_passes_data_and_setup_quirk(oracle):
    import cutesdr_amd as ca
    g, r = ca.CNoiseProc(), oracle.CNoiseProc()
    x = impulsive(3, 5000, 2e6)
    assert np.array_equal(g.ProcessBlanker(x), x)             # constructed off (noiseproc.cpp:64)
    for o in (g, r):
        o.SetupBlanker(True, 50.0, 2.0, 2e6)
    y1g, y1r = g.ProcessBlanker(x), r.ProcessBlanker(x)
    assert np.array_equal(y1g, y1r)
    for o in (g, r):
        o.SetupBlanker(True, 50.0, 2.0, 1e6)                  # rate-only change: ignored (noiseproc.cpp:80-86), state kept
    assert np.array_equal(g.ProcessBlanker(x), r.ProcessBlanker(x))
    from cutesdr_amd._capi import CsdrError
    with pytest.raises(CsdrError):
        g.SetupBlanker(True, 51.0, 2.0, 10e6)                 # 5 ms window > 32768 entries: the reference overruns


def test_blanker_batch_channels_independent(oracle):
    import cutesdr_amd as ca
    C, n, fs = 5, 40000, 2e6
    b = ca.NoiseProcBatch(C)
    b.setup(True, 50.0, 2.0, fs)
    b.setup(True, 25.0, 50.0, fs, channel=3)
    b.setup(False, 50.0, 2.0, fs, channel=4)
    x = np.stack([impulsive(10 + c, 2 * n, fs) for c in range(C)])
    refs = [oracle.CNoiseProc() for _ in range(C)]
    for c, r in enumerate(refs):
        r.SetupBlanker(c != 4, 25.0 if c == 3 else 50.0, 50.0 if c == 3 else 2.0, fs)
    for part in (x[:, :n], x[:, n:]):
        got = b.process(part)
        for c in range(C):
            assert np.array_equal(got[c].astype(np.complex128), refs[c].ProcessBlanker(part[c])), c


@pytest.mark.parametrize("pkt_len", [1028, 1444])
def test_unpack_bit_exact(oracle, pkt_len):
    import cutesdr_amd as ca
    rng = np.random.default_rng(pkt_len)
    raw = rng.integers(0, 256, (7, pkt_len), dtype=np.uint8)
    raw[0, 4:10] = [0x00, 0x80, 0xFF, 0x7F, 0x00, 0x00]      # extreme values at the front
    want = oracle.unpack_packets(raw, pkt_len)
    got = ca.unpack_packets(raw, pkt_len)
    assert len(got) == len(want) == 7 * (240 if pkt_len == 1444 else 256)
    assert np.array_equal(got, want)
    # batch form with DC offsets (sdrinterface.cpp:889-894)
    raw3 = rng.integers(0, 256, (3, 4, pkt_len), dtype=np.uint8)
    dc = np.array([[1.5, -2.25], [0.0, 0.0], [-100.0, 37.0]])
    got3 = ca.unpack_packets_batch(raw3, pkt_len, dc)
    for c in range(3):
        w = oracle.unpack_packets(raw3[c], pkt_len) - (dc[c, 0] + 1j * dc[c, 1])
        assert np.abs(got3[c] - w).max() <= 4e-3               # fp32 result of an fp64 subtraction near +-32768
    from cutesdr_amd._capi import CsdrError
    with pytest.raises(CsdrError):
        ca.unpack_packets(np.zeros((1, 1000), dtype=np.uint8), 1000)


def test_spurcal_matches_oracle(oracle):
    import cutesdr_amd as ca
    rng = np.random.default_rng(9)
    x = (50.0 * (rng.standard_normal(300000) + 1j * rng.standard_normal(300000)) + (12.5 - 7.75j))
    x = x.astype(np.complex64).astype(np.complex128)
    dg, dr = np.zeros(2), np.zeros(2)
    for a, b in ((0, 1000), (1000, 120000), (120000, 300000)):
        dg = ca.spurcal(dg, x[a:b]); dr = oracle.spurcal(dr, x[a:b])
        assert np.abs(dg - dr).max() <= 1e-9 * 50.0
    assert dg[0] == pytest.approx(12.5 * (1 - (1 - 1e-5) ** 300000), abs=0.6)     # + the noise the mean has not averaged out


def _pack16(x):
    """16-bit wire format: 4 header bytes, then 256 little-endian I,Q pairs per datagram"""
    iq = np.empty(2 * len(x), dtype=np.int16)
    iq[0::2] = np.clip(np.round(x.real), -32768, 32767); iq[1::2] = np.clip(np.round(x.imag), -32768, 32767)
    body = iq.view(np.uint8).reshape(-1, 1024)
    return np.concatenate([np.zeros((body.shape[0], 4), dtype=np.uint8), body], axis=1)


def _pack24(x):
    """24-bit wire format: 4 header bytes, then 240 I,Q pairs of 3 bytes each (value * 256 on the 16-bit scale)"""
    v = np.empty(2 * len(x), dtype=np.int64)
    v[0::2] = np.clip(np.round(x.real * 256.0), -(1 << 23), (1 << 23) - 1)
    v[1::2] = np.clip(np.round(x.imag * 256.0), -(1 << 23), (1 << 23) - 1)
    u = (v & 0xFFFFFF).astype(np.uint32)
    b3 = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).reshape(-1, 1440)
    return np.concatenate([np.zeros((b3.shape[0], 4), dtype=np.uint8), b3], axis=1)


@pytest.mark.parametrize("blanker", [10.0, 11.0, 0.0], ids=["blanker-odd-delay", "blanker-even-delay", "no-blanker"])
@pytest.mark.parametrize("pkt_len", [1028, 1444])
def test_packets_to_audio_whole_front_end(oracle, pkt_len, blanker):
    """Datagrams -> [blanker ->] down-converter -> FastFIR -> AGC -> demodulator on the device
    (csdr_demod_batch_process_packets) against the oracle's composition of the same steps.  No unpack pass
    runs: the first input-rate kernel decodes the datagrams in its loads (wire_format.hpp) -- the blanker when it
    is on, the down-converter otherwise."""
    import cutesdr_amd as ca
    from util_signals import fm_carrier, am_carrier, FULL_SCALE
    import test_postchain_gpu as T
    width = blanker or 10.0                                   # 20 / 22 samples at 2 MS/s: the delayed sample the blanked
                                                              # down-converter fetches is 11 (odd) / 12 (even) behind
    per = 240 if pkt_len == 1444 else 256
    fs, C = 2e6, 3
    npk = 19968 * 32 // per if per == 256 else 19968 * 30 // per          # whole windows of the chain per call
    n = npk * per
    from util_signals import tones_plus_noise
    # (the third receiver is CW: its plan starts with CIC-3 stages -- the blanked down-converter's other first stage)
    sig = [fm_carrier(2 * n, fs, 100e3, dbfs=-20.0), am_carrier(2 * n, fs, 101e3, dbfs=-20.0, channel=1),
           tones_plus_noise(12, 2 * n, fs, [102e3, 102e3 + 300.0])]
    rng = np.random.default_rng(5)
    for x in sig:                                             # impulses for the blanker to remove
        hits = rng.random(2 * n) < 5e-5
        x[hits] += 30000.0
    raw = np.stack([(_pack24 if pkt_len == 1444 else _pack16)(x) for x in sig])     # [C, 2*npk, pkt_len]
    b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
    nb = None
    if blanker:
        nb = ca.NoiseProcBatch(C); nb.setup(True, 30.0, width, fs)
    refs, rnb = [], []
    for c, (name, f) in enumerate((("FM", -100e3), ("AM", -101e3), ("CWU", -102e3))):
        m, kw = T.MODES[name]
        b.set_demod(c, m, T.info(ca, **kw))
        r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, T.info(oracle, **kw)); r.SetDemodFreq(f)
        refs.append(r)
        q = oracle.CNoiseProc(); q.SetupBlanker(bool(blanker), 30.0, width, fs); rnb.append(q)
    b.commit()
    b.set_freq(0, -100e3); b.set_freq(1, -101e3); b.set_freq(2, -102e3)
    first = [0, 0, 0]
    for call in range(2):
        part = raw[:, call * npk:(call + 1) * npk]
        got = b.process_packets(part, pkt_len, nb)
        for c in range(C):
            xs = oracle.unpack_packets(part[c], pkt_len)
            want = refs[c].process_append(rnb[c].ProcessBlanker(xs) if blanker else xs)
            assert len(got[c]) == len(want) and len(want) % 1024 == 0 and (len(want) > 0 or c == 2), c
            # every burst from the first one under the chain rule (test_postchain_gpu.py)
            # (with the blanker the FM receiver's first burst -- impulses into an empty delay line -- differs by all of
            # full scale, 15 x the usual start-up difference: its bounds start one burst later)
            if not len(want):
                continue
            T.check_chain_bursts(T.burst_errors(got[c], want), ("FM", "AM", "CWU")[c], first[c], (c, call),
                                 fm_late=1 if blanker else 0)
            first[c] += len(want) // 1024


@pytest.mark.parametrize("width", [10.0, 11.0], ids=["odd-delay", "even-delay"])
def test_float_rows_with_fused_blanker(oracle, width):
    """csdr_demod_batch_process_blanked: fp32 rows, the blanker's mask pass and the down-converter that zeroes the
    delayed sample under it -- against the oracle's blanker followed by its chain, FM / AM / CW receivers (three plan
    groups), two calls, every burst under the chain rule."""
    import cutesdr_amd as ca
    from util_signals import fm_carrier, am_carrier, tones_plus_noise
    import test_postchain_gpu as T
    fs, C = 2e6, 3
    n = 19968 * 30
    sig = [fm_carrier(2 * n, fs, 100e3, dbfs=-20.0), am_carrier(2 * n, fs, 101e3, dbfs=-20.0, channel=1),
           tones_plus_noise(12, 2 * n, fs, [102e3, 102e3 + 300.0])]
    rng = np.random.default_rng(6)
    for x in sig:
        x[rng.random(2 * n) < 5e-5] += 30000.0
    xs = np.stack([x.astype(np.complex64) for x in sig])
    b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
    nb = ca.NoiseProcBatch(C); nb.setup(True, 30.0, width, fs)
    refs, rnb = [], []
    for c, (name, f) in enumerate((("FM", -100e3), ("AM", -101e3), ("CWU", -102e3))):
        m, kw = T.MODES[name]
        b.set_demod(c, m, T.info(ca, **kw))
        r = oracle.CDemodulator(2048); r.SetInputSampleRate(fs); r.SetDemod(m, T.info(oracle, **kw)); r.SetDemodFreq(f)
        refs.append(r)
        q = oracle.CNoiseProc(); q.SetupBlanker(True, 30.0, width, fs); rnb.append(q)
    b.commit()
    b.set_freq(0, -100e3); b.set_freq(1, -101e3); b.set_freq(2, -102e3)
    first = [0, 0, 0]
    for call in range(2):
        part = xs[:, call * n:(call + 1) * n]
        got = b.process_blanked(part, nb)
        for c in range(C):
            want = refs[c].process_append(rnb[c].ProcessBlanker(part[c].astype(np.complex128)))
            assert len(got[c]) == len(want) and len(want) % 1024 == 0 and len(want) > 0, c
            T.check_chain_bursts(T.burst_errors(got[c], want), ("FM", "AM", "CWU")[c], first[c], (c, call), fm_late=1)
            first[c] += len(want) // 1024


@pytest.mark.parametrize("fs,width", [(2e6, 20.0), (2000200.0, 21.0)], ids=["odd-lags", "even-lags"])
@pytest.mark.parametrize("pkt_len", [1028, 1444])
def test_blanker_reading_datagrams_is_sample_exact(oracle, pkt_len, fs, width):
    """The blanker kernel fed with datagrams (three streams per sample: new, leaving the window, delayed; a tile inside
    24-bit datagrams takes them as sample pairs, so the parity of the two lags picks the decode -- 10 001 / 21 samples
    at 2 MS/s, 10 002 / 22 at 2.0002 MS/s with a 21 us width) against the oracle blanker on the oracle's unpacked
    samples: outputs are zeros or delayed copies of exactly representable inputs, so every word must match, ragged
    calls included."""
    import ctypes as C_
    import cutesdr_amd as ca
    L = ca.lib()
    L.csdr__noiseproc_batch_process_packets.restype = C_.c_int
    L.csdr__noiseproc_batch_process_packets.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_int, C_.c_int, C_.c_void_p,
                                                        C_.c_longlong, C_.c_void_p]
    per = 240 if pkt_len == 1444 else 256
    Cn = 3
    rng = np.random.default_rng(11)
    counts = [3, 130, 1, 300, 77]                              # datagrams per call
    tot = sum(counts) * per
    x = [1500.0 * (rng.standard_normal(tot) + 1j * rng.standard_normal(tot)) for _ in range(Cn)]
    for xc in x:
        xc[rng.random(tot) < 2e-4] += 25000.0
    raw = np.stack([(_pack24 if pkt_len == 1444 else _pack16)(xc) for xc in x])
    nb = ca.NoiseProcBatch(Cn); nb.setup(True, 25.0, width, fs)
    refs = []
    for c in range(Cn):
        q = oracle.CNoiseProc(); q.SetupBlanker(True, 25.0, width, fs); refs.append(q)
    k0 = 0
    for npk in counts:
        part = np.ascontiguousarray(raw[:, k0:k0 + npk])
        k0 += npk
        dp, do = ca.DeviceBuffer(part.nbytes), ca.DeviceBuffer(Cn * npk * per * 8)
        dp.upload(part)
        assert L.csdr__noiseproc_batch_process_packets(nb.h, C_.c_void_p(dp.ptr), npk, pkt_len, C_.c_void_p(do.ptr), npk * per, None) == 0
        ca.sync()
        got = do.download(np.complex64, Cn * npk * per).reshape(Cn, npk * per)
        for c in range(Cn):
            want = refs[c].ProcessBlanker(oracle.unpack_packets(part[c], pkt_len))
            assert np.array_equal(got[c], want.astype(np.complex64)), (npk, c)


@pytest.mark.parametrize("fs,width", [(2e6, 20.0), (2000200.0, 21.0), (500e3, 100.0), (6e6, 10.0), (2e6, 2040.0)],
                         ids=["ring-odd-lag", "ring-even-lag", "window-below-a-tile", "window-beyond-the-ring", "ring-widest-blank"])
@pytest.mark.parametrize("src", ["rows", 1028, 1444])
def test_blank_mask_is_bit_exact(oracle, src, fs, width):
    """The blanker's MASK form (what csdr_demod_batch_process_packets runs in front of the fused down-converter): one
    bit per sample, no samples out.  Applied to the input the way the down-converter applies it -- out[i] = 0 under the
    mask, else x[i - delay_n - 1] -- the mask must give the oracle blanker's output word for word.  Float rows and
    both datagram formats; sample rates whose 5 ms window takes the LDS ring of magnitudes (10 001 / 10 002 samples:
    both parities of the aligned ring reads) and rates that fall back to the two-stream form (2 501: shorter than a
    tile; 30 001: longer than the ring); ragged calls, one of them long enough for several segments per channel (later
    segments rebuild the window sum and the ring from the samples in front of them)."""
    import ctypes as C_
    import cutesdr_amd as ca
    L = ca.lib()
    L.csdr__noiseproc_batch_mask.restype = C_.c_int
    L.csdr__noiseproc_batch_mask.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_longlong, C_.c_void_p, C_.c_int, C_.c_int, C_.c_int,
                                             C_.c_void_p, C_.c_longlong, C_.POINTER(C_.c_void_p), C_.POINTER(C_.c_void_p),
                                             C_.c_void_p]
    Cn = 3
    per = 240 if src == 1444 else 256
    counts = [3, 130, 1, 1200, 77]                             # datagrams (or 256-sample rows) per call
    tot = sum(counts) * per
    rng = np.random.default_rng(17)
    x = [1500.0 * (rng.standard_normal(tot) + 1j * rng.standard_normal(tot)) for _ in range(Cn)]
    for xc in x:
        xc[rng.random(tot) < 2e-4] += 25000.0
    if src == "rows":
        xs = [xc.astype(np.complex64) for xc in x]
    else:
        raw = np.stack([(_pack24 if src == 1444 else _pack16)(xc) for xc in x])
        xs = [oracle.unpack_packets(raw[c], src).astype(np.complex64) for c in range(Cn)]
    nb = ca.NoiseProcBatch(Cn); nb.setup(True, 25.0, width, fs)
    refs = []
    for c in range(Cn):
        q = oracle.CNoiseProc(); q.SetupBlanker(True, 25.0, width, fs); refs.append(q)
    delay1 = int(width * 1e-6 * fs) // 2 + 1                   # delay_n + 1 (noiseproc.cpp:92-99)
    k0, blanked = 0, 0
    for npk in counts:
        n = npk * per
        a0 = k0 * per
        words = (n + 31) // 32 + 64
        dm = ca.DeviceBuffer(Cn * words * 4)
        st, hi = C_.c_void_p(), C_.c_void_p()
        if src == "rows":
            part = np.ascontiguousarray(np.stack([xc[a0:a0 + n] for xc in xs]))
            dp = ca.DeviceBuffer(part.nbytes); dp.upload(part)
            rc = L.csdr__noiseproc_batch_mask(nb.h, C_.c_void_p(dp.ptr), n, None, 0, 0, n, C_.c_void_p(dm.ptr), words,
                                              C_.byref(st), C_.byref(hi), None)
        else:
            part = np.ascontiguousarray(raw[:, k0:k0 + npk])
            dp = ca.DeviceBuffer(part.nbytes); dp.upload(part)
            rc = L.csdr__noiseproc_batch_mask(nb.h, None, 0, C_.c_void_p(dp.ptr), npk, src, n, C_.c_void_p(dm.ptr), words,
                                              C_.byref(st), C_.byref(hi), None)
        assert rc == 0
        ca.sync()
        m = dm.download(np.uint32, Cn * words).reshape(Cn, words)
        for c in range(Cn):
            bits = ((m[c, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n].astype(bool)
            idx = np.arange(a0, a0 + n) - delay1
            delayed = np.where(idx >= 0, xs[c][np.maximum(idx, 0)], 0).astype(np.complex64)
            got = np.where(bits, np.complex64(0), delayed)
            want = refs[c].ProcessBlanker(xs[c][a0:a0 + n].astype(np.complex128)).astype(np.complex64)
            assert np.array_equal(got, want), (npk, c, np.nonzero(got != want)[0][:5])
            blanked += int(bits.sum())
        k0 += npk
    assert 0 < blanked < Cn * tot                              # the case blanks, and not everything


@pytest.mark.parametrize("seed", range(8))
def test_blanker_random_rates_around_the_ring_limits(oracle, seed):
    """Both forms of the blanker at random sample rates, thresholds, widths and call lengths -- rates drawn around the
    limits of the LDS-ring form (a 5 ms window of 2048 / 4096 samples at its lower end, 12288 / 14336 at its upper one:
    0.41, 0.82, 2.46 and 2.87 MS/s) so that neighbouring windows take different kernels -- against the oracle, word for
    word: the sample form's output, and the mask form's bits applied to the input."""
    import ctypes as C_
    import cutesdr_amd as ca
    L = ca.lib()
    L.csdr__noiseproc_batch_mask.restype = C_.c_int
    L.csdr__noiseproc_batch_mask.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_longlong, C_.c_void_p, C_.c_int, C_.c_int, C_.c_int,
                                             C_.c_void_p, C_.c_longlong, C_.POINTER(C_.c_void_p), C_.POINTER(C_.c_void_p),
                                             C_.c_void_p]
    rng = np.random.default_rng(1000 + seed)
    edge = [2048, 4096, 12288, 14336][seed % 4]
    win = int(edge + rng.integers(-3, 4))                      # mag_n + 1 within three samples of a limit
    fs = (win - 1 + 0.5) / 0.005                                # mag_n = (int)(0.005 fs)
    thresh = float(rng.uniform(10.0, 80.0))
    width = float(rng.uniform(1.0, 60.0))                      # microseconds
    Cn = 2
    calls = [int(v) for v in rng.integers(1, 60000, size=4)] + [300000]
    tot = sum(calls)
    xs = []
    for c in range(Cn):
        x = impulsive(seed * 10 + c, tot, fs, rate=3e-4)
        xs.append(x.astype(np.complex64))
    g = ca.NoiseProcBatch(Cn); g.setup(True, thresh, width, fs)
    m = ca.NoiseProcBatch(Cn); m.setup(True, thresh, width, fs)
    refs = []
    for c in range(Cn):
        q = oracle.CNoiseProc(); q.SetupBlanker(True, thresh, width, fs); refs.append(q)
    delay1 = max(1, min(int(width * 1e-6 * fs), 4096)) // 2 + 1
    a0 = 0
    for n in calls:
        part = np.ascontiguousarray(np.stack([x[a0:a0 + n] for x in xs]))
        got = g.process(part)
        words = (n + 31) // 32 + 64
        dp, dm = ca.DeviceBuffer(part.nbytes), ca.DeviceBuffer(Cn * words * 4)
        dp.upload(part)
        st, hi = C_.c_void_p(), C_.c_void_p()
        assert L.csdr__noiseproc_batch_mask(m.h, C_.c_void_p(dp.ptr), n, None, 0, 0, n, C_.c_void_p(dm.ptr), words,
                                            C_.byref(st), C_.byref(hi), None) == 0
        ca.sync()
        mw = dm.download(np.uint32, Cn * words).reshape(Cn, words)
        for c in range(Cn):
            want = refs[c].ProcessBlanker(xs[c][a0:a0 + n].astype(np.complex128)).astype(np.complex64)
            assert np.array_equal(got[c], want), ("samples", win, n, c, np.nonzero(got[c] != want)[0][:5])
            bits = ((mw[c, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n].astype(bool)
            idx = np.arange(a0, a0 + n) - delay1
            delayed = np.where(idx >= 0, xs[c][np.maximum(idx, 0)], 0).astype(np.complex64)
            masked = np.where(bits, np.complex64(0), delayed)
            assert np.array_equal(masked, want), ("mask", win, n, c, np.nonzero(masked != want)[0][:5])
        a0 += n


def test_pipelined_packets_with_blanker_give_the_strict_mode_words():
    """csdr_demod_batch_process_packets with a blanker, pipelined mode: the blanker of call k+1 writes the batch's
    own staging buffer while the down-converters of call k (on the batch's internal streams) may still be reading
    it -- the call has to order itself behind them.  Four calls issued back to back without host synchronisation;
    every audio word equals the strict mode's."""
    import ctypes as C_
    import cutesdr_amd as ca
    from util_signals import fm_carrier, am_carrier
    import test_postchain_gpu as T
    L = ca.lib()
    pkt_len, per, fs, Cn, calls = 1444, 240, 2e6, 6, 4
    npk = 19968 * 30 // per * 2
    n = npk * per
    names = ["FM", "AM", "USB"]
    rng = np.random.default_rng(21)
    raws = []
    for c in range(Cn):
        x = (fm_carrier if c % 3 == 0 else am_carrier)(calls * n, fs, 100e3 + 900.0 * c, dbfs=-20.0, channel=c)
        x[rng.random(calls * n) < 5e-5] += 30000.0
        raws.append(_pack24(x))
    raw = np.stack(raws)                                       # [Cn, calls * npk, pkt_len]
    outs = []
    for pipelined in (False, True):
        b = ca.DemodBatch(Cn, 2048); b.set_input_rate(fs)
        for c in range(Cn):
            m, kw = T.MODES[names[c % 3]]
            b.set_demod(c, m, T.info(ca, **kw))
        b.commit()
        for c in range(Cn):
            b.set_freq(c, -(100e3 + 900.0 * c))
        nb = ca.NoiseProcBatch(Cn); nb.setup(True, 30.0, 10.0, fs)
        if pipelined:
            b.set_pipelined(True)
        cap = n // 8
        dps = []
        for k in range(calls):                                 # one datagram buffer per call, all resident before the first call
            part = np.ascontiguousarray(raw[:, k * npk:(k + 1) * npk])
            dp = ca.DeviceBuffer(part.nbytes); dp.upload(part); dps.append(dp)
        dout = ca.DeviceBuffer(4 * Cn * cap * calls)
        ca.sync()
        counts = []
        for k in range(calls):                                 # no synchronisation between the calls
            rc = L.csdr_demod_batch_process_packets(b.h, C_.c_void_p(dps[k].ptr), npk, pkt_len, nb.h,
                                                    C_.c_void_p(dout.ptr + 4 * Cn * cap * k), cap, None)
            assert rc == 0
            counts.append([b.out_count(c) for c in range(Cn)])
        b.flush()
        ca.sync()
        y = dout.download(np.float32, Cn * cap * calls).reshape(calls, Cn, cap)
        outs.append([[y[k, c, :counts[k][c]].copy() for c in range(Cn)] for k in range(calls)])
    for k in range(calls):
        for c in range(Cn):
            assert len(outs[0][k][c]) > 0
            assert np.array_equal(outs[0][k][c].view(np.uint32), outs[1][k][c].view(np.uint32)), (k, c)


def test_blanked_down_converter_runtime_plan_gives_the_compiled_plans_words():
    """The down-converter that applies the blanker's mask exists as a compiled-plan kernel per decimator sequence and as
    the run-time-plan kernel a sequence outside the table takes: the fused chain from fp32 rows and from 24-bit
    datagrams must give the same audio words through either (csdr__downconv_force_dynamic), FM / AM / CW receivers."""
    import ctypes as C_
    import cutesdr_amd as ca
    from util_signals import fm_carrier, am_carrier, tones_plus_noise
    import test_postchain_gpu as T
    L = ca.lib()
    L.csdr__downconv_force_dynamic.restype = C_.c_int
    L.csdr__downconv_force_dynamic.argtypes = [C_.c_int]
    fs, C = 2e6, 3
    npk = 19968 * 30 // 240
    n = npk * 240
    sig = [fm_carrier(n, fs, 100e3, dbfs=-20.0), am_carrier(n, fs, 101e3, dbfs=-20.0, channel=1),
           tones_plus_noise(12, n, fs, [102e3, 102e3 + 300.0])]
    rng = np.random.default_rng(8)
    for x in sig:
        x[rng.random(n) < 5e-5] += 30000.0
    raw = np.stack([_pack24(x) for x in sig])
    xs = np.stack([x.astype(np.complex64) for x in sig])
    outs = []
    try:
        for dyn in (0, 1):
            L.csdr__downconv_force_dynamic(dyn)
            res = []
            for form in ("rows", "packets"):
                b = ca.DemodBatch(C, 2048); b.set_input_rate(fs)
                for c, name in enumerate(("FM", "AM", "CWU")):
                    m, kw = T.MODES[name]
                    b.set_demod(c, m, T.info(ca, **kw))
                b.commit()
                for c in range(C):
                    b.set_freq(c, -100e3 - 1000.0 * c)
                nb = ca.NoiseProcBatch(C); nb.setup(True, 30.0, 10.0, fs)
                res.append(b.process_blanked(xs, nb) if form == "rows" else b.process_packets(raw, 1444, nb))
            outs.append(res)
    finally:
        L.csdr__downconv_force_dynamic(0)
    for f in range(2):
        for c in range(C):
            assert len(outs[0][f][c]) == len(outs[1][f][c]) > 0
            assert np.array_equal(outs[0][f][c], outs[1][f][c]), (f, c, np.abs(outs[0][f][c] - outs[1][f][c]).max())


def test_pipelined_blanked_rows_give_the_strict_mode_words():
    """csdr_demod_batch_process_blanked in pipelined mode (the mask buffer is single-buffered and the blanker's history
    halves alternate: call k+1's blanker must order itself behind call k's down-converters): four calls back to back
    without host synchronisation, every audio word equal to the strict mode's."""
    import ctypes as C_
    import cutesdr_amd as ca
    from util_signals import fm_carrier, am_carrier
    import test_postchain_gpu as T
    L = ca.lib()
    fs, Cn, calls = 2e6, 6, 4
    n = 19968 * 8
    names = ["FM", "AM", "USB"]
    rng = np.random.default_rng(23)
    xs = []
    for c in range(Cn):
        x = (fm_carrier if c % 3 == 0 else am_carrier)(calls * n, fs, 100e3 + 900.0 * c, dbfs=-20.0, channel=c)
        x[rng.random(calls * n) < 5e-5] += 30000.0
        xs.append(x.astype(np.complex64))
    xs = np.stack(xs)
    outs = []
    for pipelined in (False, True):
        b = ca.DemodBatch(Cn, 2048); b.set_input_rate(fs)
        for c in range(Cn):
            m, kw = T.MODES[names[c % 3]]
            b.set_demod(c, m, T.info(ca, **kw))
        b.commit()
        for c in range(Cn):
            b.set_freq(c, -(100e3 + 900.0 * c))
        nb = ca.NoiseProcBatch(Cn); nb.setup(True, 30.0, 10.0, fs)
        if pipelined:
            b.set_pipelined(True)
        cap = n // 8
        dins = []
        for k in range(calls):
            part = np.ascontiguousarray(xs[:, k * n:(k + 1) * n])
            d = ca.DeviceBuffer(part.nbytes); d.upload(part); dins.append(d)
        dout = ca.DeviceBuffer(4 * Cn * cap * calls)
        ca.sync()
        counts = []
        for k in range(calls):
            rc = L.csdr_demod_batch_process_blanked(b.h, C_.c_void_p(dins[k].ptr), n, n, nb.h,
                                                    C_.c_void_p(dout.ptr + 4 * Cn * cap * k), cap, None)
            assert rc == 0
            counts.append([b.out_count(c) for c in range(Cn)])
        b.flush()
        ca.sync()
        y = dout.download(np.float32, Cn * cap * calls).reshape(calls, Cn, cap)
        outs.append([[y[k, c, :counts[k][c]].copy() for c in range(Cn)] for k in range(calls)])
    for k in range(calls):
        for c in range(Cn):
            assert len(outs[0][k][c]) > 0
            assert np.array_equal(outs[0][k][c].view(np.uint32), outs[1][k][c].view(np.uint32)), (k, c)


@pytest.mark.parametrize("src", [1028, 1444])
def test_integer_mask_kernel_decides_what_the_general_one_does(oracle, src):
    """Round 6: on datagram input the mask pass runs in integers (noiseblank_mask_int_kernel: magnitudes and moving sums
    as integers in units of 2^-8, the trigger test screened once per thread in fp32, the exact fp64 test only for threads
    with a candidate).  Same datagrams through the general kernel (csdr__noiseproc_set_int(0)) and through the integer
    one: every mask word equal, call after call (the state each leaves feeds the next call), thresholds low enough for
    hundreds of triggers and blank windows that cross tiles, segments and calls; and after a call on FLOAT rows the
    object falls back to the general kernel by itself (its sums are no longer integral) and still matches."""
    import ctypes as C_
    import cutesdr_amd as ca
    L = ca.lib()
    L.csdr__noiseproc_set_int.restype = C_.c_int
    L.csdr__noiseproc_set_int.argtypes = [C_.c_int]
    L.csdr__noiseproc_batch_mask.restype = C_.c_int
    L.csdr__noiseproc_batch_mask.argtypes = [C_.c_void_p, C_.c_void_p, C_.c_longlong, C_.c_void_p, C_.c_int, C_.c_int, C_.c_int,
                                             C_.c_void_p, C_.c_longlong, C_.POINTER(C_.c_void_p), C_.POINTER(C_.c_void_p),
                                             C_.c_void_p]
    Cn, fs = 5, 2e6
    per = 240 if src == 1444 else 256
    counts = [2000, 7, 900, 1, 3000]
    tot = sum(counts) * per
    rng = np.random.default_rng(29)
    x = [900.0 * (1 + c) * (rng.standard_normal(tot) + 1j * rng.standard_normal(tot)) for c in range(Cn)]
    for xc in x:
        xc[rng.random(tot) < 4e-4] += 20000.0
        xc[tot // 3:tot // 3 + 3000] *= 0.01                   # a quiet stretch: the sum falls, the next impulse triggers harder
    raw = np.stack([(_pack24 if src == 1444 else _pack16)(xc) for xc in x])
    rows_call = 2                                              # after this call one float-rows call is put in between
    xs = [oracle.unpack_packets(raw[c], src).astype(np.complex64) for c in range(Cn)]

    def run(int_on):
        assert L.csdr__noiseproc_set_int(int_on) == int_on
        nb = ca.NoiseProcBatch(Cn)
        for c in range(Cn):
            nb.setup(True, 12.0 + 3.0 * c, 15.0 + 40.0 * c, fs, channel=c)
        masks, k0 = [], 0
        for call, npk in enumerate(counts):
            n = npk * per
            words = (n + 31) // 32 + 64
            dm = ca.DeviceBuffer(Cn * words * 4)
            st, hi = C_.c_void_p(), C_.c_void_p()
            if call == rows_call + 1:                          # the same samples as float rows: the object leaves the integer form
                part = np.ascontiguousarray(np.stack([xc[k0 * per:k0 * per + n] for xc in xs]))
                dp = ca.DeviceBuffer(part.nbytes); dp.upload(part)
                rc = L.csdr__noiseproc_batch_mask(nb.h, C_.c_void_p(dp.ptr), n, None, 0, 0, n, C_.c_void_p(dm.ptr), words,
                                                  C_.byref(st), C_.byref(hi), None)
            else:
                part = np.ascontiguousarray(raw[:, k0:k0 + npk])
                dp = ca.DeviceBuffer(part.nbytes); dp.upload(part)
                rc = L.csdr__noiseproc_batch_mask(nb.h, None, 0, C_.c_void_p(dp.ptr), npk, src, n, C_.c_void_p(dm.ptr), words,
                                                  C_.byref(st), C_.byref(hi), None)
            assert rc == 0
            ca.sync()
            masks.append(dm.download(np.uint32, Cn * words).reshape(Cn, words)[:, :(n + 31) // 32].copy())
            k0 += npk
        return masks
    try:
        general = run(0)
        integer = run(1)
    finally:
        L.csdr__noiseproc_set_int(1)
    blanked = 0
    for call, (g, i) in enumerate(zip(general, integer)):
        assert np.array_equal(g, i), (call, np.nonzero(g != i))
        blanked += int(np.unpackbits(i.view(np.uint8)).sum())
    assert blanked > 2000                                      # hundreds of triggers, each a window of 30 ... 350 samples
    # and the integer run against the oracle's blanker, channel 0, first call (what test_blank_mask_is_bit_exact does for all)
    q = oracle.CNoiseProc(); q.SetupBlanker(True, 12.0, 15.0, fs)
    n0 = counts[0] * per
    delay1 = int(15.0 * 1e-6 * fs) // 2 + 1
    bits = ((integer[0][0, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n0].astype(bool)
    idx = np.arange(n0) - delay1
    delayed = np.where(idx >= 0, xs[0][np.maximum(idx, 0)], 0).astype(np.complex64)
    want = q.ProcessBlanker(xs[0][:n0].astype(np.complex128)).astype(np.complex64)
    assert np.array_equal(np.where(bits, np.complex64(0), delayed), want)
