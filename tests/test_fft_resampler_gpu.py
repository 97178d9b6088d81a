"""GPU parity of the CFft display spectrum / plain transforms (K3) and of CFractResampler (K5)
against the fp64 oracle, through the C ABI.  K3 tolerance: 0.01 dB on every bin within 60 dB of
the frame's strongest bin, 0.5 dB down to 90 dB below it, 2 dB on every other bin above -150 dBFS
(an fp32 transform has its rounding floor ~125 dB below the strongest component, so weak bins beside a
strong carrier cannot hold 0.01 dB); plain transforms 2e-5 * N * max|x|; resampler 1e-5 * max|x|, counts exact,
int16 outputs within 1 LSB."""
import numpy as np
import pytest
from util_signals import tones_plus_noise, FULL_SCALE

pytestmark = pytest.mark.gpu


def assert_spectrum_close(got, want):
    """got / want in bels (display order)."""
    err = np.abs(got - want)
    near = want > want.max() - 6.0                           # within 60 dB of the strongest bin
    assert err[near].max() <= 0.001                          # 0.01 dB
    mid = (want > want.max() - 9.0) & ~near                  # 60 .. 90 dB below it
    if mid.any():
        assert err[mid].max() <= 0.05                        # 0.5 dB
    deep = (want > -15.0) & (want <= want.max() - 9.0)       # everything else above -150 dBFS: a bin where the noise of ONE
    if deep.any():                                           # frame happens to cancel sits at the fp32 transform's own floor
        assert err[deep].max() <= 0.2                        # (seen: 0.53 dB in one of 16384 bins x 72 frames, no averaging)


@pytest.mark.parametrize("n,ave", [(4096, 1), (2048, 4), (8192, 2), (16384, 3), (512, 2), (1024, 1), (32768, 2), (65536, 1)])
def test_display_spectrum_matches_oracle(oracle, n, ave):
    import cutesdr_amd as ca
    fs = 2e6
    f, r = ca.CFft(), oracle.CFft()
    for o in (f, r):
        o.SetFFTParams(n, False, 0.0, fs)
        o.SetFFTAve(ave)
    for k in range(7):                                       # C1: -20 dBFS tone at +250 kHz, noise -70 dBFS
        x = tones_plus_noise(k, n, fs, [250e3, -611e3 + 977.0 * k], start=k * n)
        assert f.PutInDisplayFFT(x) == r.PutInDisplayFFT(x) == k + 1
        got, want = f.ave_buf().astype(np.float64), r.ave_buf()
        assert_spectrum_close(got, want)
        assert np.argmax(got) == np.argmax(want)
    w = min(700, n // 2)                                     # never more pixels than table entries (fft.cpp:165)
    ovg, pg = f.GetScreenIntegerFFTData(1 << 16, w, 0.0, -160.0, -900000, 900000)
    ovr, pr = r.GetScreenIntegerFFTData(1 << 16, w, 0.0, -160.0, -900000, 900000)
    assert ovg == ovr is False
    assert np.abs(pg - pr).max() <= 8                         # 65536 px over 160 dB: 0.02 dB per 8 px
    w = min(n - 1, 30000)             # pixel*bins stays inside the reference's 32-bit arithmetic (fft.cpp:355)
    ovg, pg = f.GetScreenIntegerFFTData(300, w, 0.0, -220.0, -1000000, 1000000)   # more pixels than bins
    ovr, pr = r.GetScreenIntegerFFTData(300, w, 0.0, -220.0, -1000000, 1000000)
    assert np.abs(pg - pr).max() <= 1
    big = x.copy(); big[5] = 32500.0
    f.PutInDisplayFFT(big); r.PutInDisplayFFT(big)
    assert f.GetScreenIntegerFFTData(100, 50, 0.0, -100.0, 0, 500000)[0] is True


def test_display_anchor_c1():
    # -20 dBFS tone at +250 kHz, Fs 2 MHz, N 4096 -> display index 2560, -1.3982 bels
    import cutesdr_amd as ca
    n, fs = 4096, 2e6
    f = ca.CFft(); f.SetFFTParams(n, False, 0.0, fs); f.SetFFTAve(1)
    f.PutInDisplayFFT(3276.7 * np.exp(2j * np.pi * 250e3 * np.arange(n) / fs))
    a = f.ave_buf()
    assert np.argmax(a) == 2560 and a[2560] == pytest.approx(-1.3982, abs=2e-4)


def test_size_clamp_and_bad_sizes():
    # fft.cpp:140-145 clamps the size to 512..65536; a size that is not a power of two is refused
    import cutesdr_amd as ca
    from cutesdr_amd._capi import CsdrError
    f = ca.CFft()
    f.SetFFTParams(64, False, 0.0, 48000.0)
    assert f.size == 512
    f.SetFFTParams(1 << 20, False, 0.0, 48000.0)
    assert f.size == 65536
    with pytest.raises(CsdrError):
        f.SetFFTParams(3000, False, 0.0, 48000.0)


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536])
def test_plain_transforms(oracle, n):
    import cutesdr_amd as ca
    rng = np.random.default_rng(n)
    x = 1000 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    f = ca.CFft(); f.SetFFTParams(n, False, 0.0, 1.0)
    tol = 2e-5 * n * np.abs(x).max() / np.sqrt(n) * 4
    assert np.abs(f.FwdFFT(x) - oracle.fft(x, +1)).max() <= tol
    assert np.abs(f.RevFFT(x) - oracle.fft(x, -1)).max() <= tol
    back = f.RevFFT(f.FwdFFT(x)) / n                          # Rev(Fwd(x)) = N x
    assert np.abs(back - x).max() <= 2e-5 * np.abs(x).max() * 20


def test_fft_batch_frames_and_channels(oracle):
    import cutesdr_amd as ca
    n, C, frames, fs = 4096, 3, 5, 2e6
    b = ca.FftBatch(C)
    b.set_params(n, False, 0.0, fs); b.set_ave(2)
    x = np.stack([tones_plus_noise(40 + c, frames * n, fs, [100e3 * (c + 1)]) for c in range(C)])
    b.put_display(x)
    for c in range(C):
        r = oracle.CFft(); r.SetFFTParams(n, False, 0.0, fs); r.SetFFTAve(2)
        for k in range(frames):
            r.PutInDisplayFFT(x[c, k * n:(k + 1) * n])
        assert b.total_count(c) == frames
        want = r.ave_buf()
        assert_spectrum_close(b.ave_buf(c).astype(np.float64), want)


@pytest.mark.parametrize("rate", [1.0, 1.6276041666666667, 0.7312, 2.5])
def test_resampler_all_overloads(oracle, rate):
    import cutesdr_amd as ca
    rng = np.random.default_rng(11)
    x = 8000 * rng.standard_normal(3 * 2048)
    xc = x + 8000j * rng.standard_normal(3 * 2048)
    for data, gain in ((x, None), (xc, None), (x, 1.7), (xc, 0.9)):
        g, r = ca.CFractResampler(), oracle.CFractResampler()
        g.Init(4096); r.Init(4096)
        for i in range(3):
            part = data[i * 2048:(i + 1) * 2048]
            got, want = g.Resample(part, rate, gain), r.Resample(part, rate, gain)
            assert len(got) == len(want)
            if gain is None:
                assert np.abs(got - want).max() <= 1e-5 * 8000 * 6
            else:
                assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1
    g = ca.CFractResampler(); g.Init(8192)
    y = g.Resample(x[:4096], 1.6276)
    assert len(y) == 2517                                    # App. A.9 anchor


def test_resampler_rate_varies_per_call_like_the_sound_sink(oracle):
    """CSoundOut::PutOutQueue (interface/soundout.cpp:196-305) calls Resample with
    m_OutRatio*(1+m_RateCorrection), a rate that moves by up to +-500 ppm from call to call
    (CalcError, :456-468), and with the volume gain on the int16 variants: counts exact, values
    within 1 LSB, across a sequence of calls with changing rate and ragged lengths."""
    import cutesdr_amd as ca
    rng = np.random.default_rng(21)
    g, r = ca.CFractResampler(), oracle.CFractResampler()
    g.Init(8192); r.Init(8192)
    base = 62500.0 / 48000.0
    gain = 10 ** ((80 - 99.0) / 39.2)                         # SetVolume(80), soundout.cpp:181-190
    total = 0
    for k in range(40):
        n = int(rng.integers(200, 1200))
        x = 9000.0 * np.sin(2 * np.pi * 0.01 * (np.arange(n) + total)) + 200.0 * rng.standard_normal(n)
        total += n
        rate = base * (1.0 + 2.38e-7 * float(rng.integers(-2000, 2000)))
        got, want = g.Resample(x, rate, gain), r.Resample(x, rate, gain)
        assert len(got) == len(want), k
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1, k


def test_resampler_batch_rows_on_one_clock(oracle):
    import cutesdr_amd as ca
    C = 5
    rng = np.random.default_rng(31)
    b = ca.ResamplerBatch(C)
    refs = [oracle.CFractResampler() for _ in range(C)]
    for r in refs:
        r.Init(8192)
    rate = 62500.0 / 48000.0
    for call, n in enumerate((1024, 1, 777, 4096, 2048)):
        x = (6000.0 * rng.standard_normal((C, n))).astype(np.float32)
        gain = None if call % 2 == 0 else 0.8
        got = b.resample(x, rate * (1 + 1e-4 * call), gain)
        for c in range(C):
            want = refs[c].Resample(x[c].astype(np.float64), rate * (1 + 1e-4 * call), gain)
            assert got.shape[1] == len(want), (call, c)
            if gain is None:
                assert np.abs(got[c] - want).max() <= 1e-5 * 6000.0 * 5, (call, c)
            else:
                assert np.abs(got[c].astype(np.int32) - want.astype(np.int32)).max() <= 1, (call, c)


@pytest.mark.parametrize("invert", [False, True])
def test_screen_mapping_on_device_for_all_channels(oracle, invert):
    """csdr_fft_batch_get_screen_all (device, every channel) against the per-channel host mapping and
    the oracle: more bins than pixels (per-pixel strongest bin), fewer bins than pixels, a narrow span."""
    import ctypes as C
    import cutesdr_amd as ca
    n, Cn, fs = 4096, 3, 2e6
    b = ca.FftBatch(Cn)
    b.set_params(n, invert, 0.0, fs); b.set_ave(1)
    x = np.stack([tones_plus_noise(70 + c, n, fs, [150e3 * (c + 1), -333e3]) for c in range(Cn)])
    b.put_display(x)
    refs = []
    for c in range(Cn):
        r = oracle.CFft(); r.SetFFTParams(n, invert, 0.0, fs); r.SetFFTAve(1); r.PutInDisplayFFT(x[c]); refs.append(r)
    for (h, w, lo, hi) in ((255, 700, -900000, 900000), (1000, 3000, -250000, 250000), (400, 300, 100000, 180000)):
        ov, pix = b.screen_all(h, w, 0.0, -160.0, lo, hi)
        for c in range(Cn):
            one = np.full(w + 1, -1, dtype=np.int32)
            rc = ca.lib().csdr_fft_batch_get_screen(b.h, c, h, w, 0.0, -160.0, lo, hi, one.ctypes.data_as(C.c_void_p))
            assert rc >= 0
            assert np.array_equal(pix[c], one[:w]), (c, h, w)
            _, want = refs[c].GetScreenIntegerFFTData(h, w, 0.0, -160.0, lo, hi)
            touched = pix[c] >= 0                             # pixels no bin maps to stay as they were
            assert np.abs(pix[c][touched] - want[touched]).max() <= 1, (c, h, w)
        assert not ov.any()


@pytest.mark.parametrize("invert", [False, True])
def test_waterfall_line_on_device_for_all_channels(oracle, invert):
    """the new top line of CPlotter's waterfall (gui/plotter.cpp:425-441: levels at MaxHeight 255, then
    m_ColorTbl[255 - y]) for every channel on the device against the oracle -- wide span (more bins than pixels), narrow
    span (fewer), and the palette itself entry by entry."""
    import ctypes as C
    import cutesdr_amd as ca
    tbl = np.zeros(256, dtype=np.uint32)
    ca.lib().csdr_plotter_color_table(tbl.ctypes.data_as(C.c_void_p))
    want_tbl = oracle.plotter_color_table()
    assert np.array_equal(tbl, want_tbl)
    assert tbl[0] == 0xff000000 and tbl[255] == 0xffff0080 and len(set(tbl.tolist())) > 200
    n, Cn, fs = 4096, 3, 2e6
    b = ca.FftBatch(Cn)
    b.set_params(n, invert, 0.0, fs); b.set_ave(1)
    x = np.stack([tones_plus_noise(170 + c, n, fs, [150e3 * (c + 1), -333e3]) for c in range(Cn)])
    b.put_display(x)
    refs = []
    for c in range(Cn):
        r = oracle.CFft(); r.SetFFTParams(n, invert, 0.0, fs); r.SetFFTAve(1); r.PutInDisplayFFT(x[c]); refs.append(r)
    for (w, lo, hi) in ((700, -900000, 900000), (3000, -250000, 250000), (300, 100000, 180000)):
        ov, pix = b.waterfall_all(w, 0.0, -160.0, lo, hi, fill=0x12345678)
        _, lev = b.screen_all(255, w, 0.0, -160.0, lo, hi)
        for c in range(Cn):
            _, want = refs[c].WaterfallLine(w, 0.0, -160.0, lo, hi, fill=0x12345678)
            touched = lev[c] >= 0
            assert np.array_equal(pix[c][~touched], np.full((~touched).sum(), 0x12345678, dtype=np.uint32))
            assert np.array_equal(pix[c][touched], tbl[255 - lev[c][touched]])            # the palette of the device's own levels
            # against the oracle: levels may differ by one count (fp32 bels), i.e. by one palette step
            idx_want = np.array([int(np.nonzero(want_tbl == v)[0][0]) if v != 0x12345678 else -1 for v in want[touched]])
            assert (idx_want >= 0).all()
            assert np.abs(idx_want - (255 - lev[c][touched])).max() <= 1, (c, w)
            assert (pix[c][touched] == want[touched]).mean() > 0.9
        assert not ov.any()


@pytest.mark.parametrize("n", [2048, 4096, 8192, 16384])
@pytest.mark.parametrize("ave", [1, 3, 10])
def test_fft_batch_many_frames_are_split_into_groups(oracle, ave, n):
    """Calls with many frames on few channels cut each channel's frames into groups (one workgroup
    each, running sum folded afterwards as a linear map): same spectrum as the frame-by-frame oracle,
    across two calls (warm-up of the average inside the first, steady state in the second) and a third one after
    SetFFTAve with a shorter average (which resets counts and sums, fft.cpp:103-113: the running sums the groups fold
    into must start from zero again).  Every single-pass kernel: 2048, 4096, 8192 (lane-pair middle pass), 16384."""
    import cutesdr_amd as ca
    C, frames, fs = 2, 72, 2e6
    b = ca.FftBatch(C)
    b.set_params(n, False, 0.0, fs); b.set_ave(ave)
    refs = []
    for c in range(C):
        r = oracle.CFft(); r.SetFFTParams(n, False, 0.0, fs); r.SetFFTAve(ave); refs.append(r)
    for call in range(3):
        if call == 2:                                           # a shorter average from here on
            b.set_ave(max(1, ave // 2))
            for r in refs:
                r.SetFFTAve(max(1, ave // 2))
        x = np.stack([tones_plus_noise(90 + c + 7 * call, frames * n, fs, [120e3 * (c + 1), -400e3 + 50e3 * call]) for c in range(C)])
        b.put_display(x)
        for c in range(C):
            for k in range(frames):
                refs[c].PutInDisplayFFT(x[c, k * n:(k + 1) * n])
            assert b.total_count(c) == (frames * (call + 1) if call < 2 else frames)    # (SetFFTAve resets, fft.cpp:103-113)
            assert_spectrum_close(b.ave_buf(c).astype(np.float64), refs[c].ave_buf())


def test_fft_batch_one_channel_many_groups(oracle):
    """One channel, 600 frames of 2048 points in a call: 75 frame groups -- more than the 64 threads of the kernel that
    computes the groups' averaging weights (it takes them in rounds) -- against the frame-by-frame oracle, two calls."""
    import cutesdr_amd as ca
    n, frames, fs, ave = 2048, 600, 2e6, 7
    b = ca.FftBatch(1)
    b.set_params(n, False, 0.0, fs); b.set_ave(ave)
    r = oracle.CFft(); r.SetFFTParams(n, False, 0.0, fs); r.SetFFTAve(ave)
    for call in range(2):
        x = tones_plus_noise(300 + call, frames * n, fs, [333e3, -120e3 + 70e3 * call])[None, :]
        b.put_display(x)
        for k in range(frames):
            r.PutInDisplayFFT(x[0, k * n:(k + 1) * n])
        assert b.total_count(0) == frames * (call + 1)
        assert_spectrum_close(b.ave_buf(0).astype(np.float64), r.ave_buf())


@pytest.mark.parametrize("n", [512, 1024, 32768])
def test_fft_batch_many_frames_multi_kernel_sizes(oracle, n):
    """Many frames per call at the sizes that do not take a single-pass kernel (below 2048, above 16384): two calls,
    the average warming up inside the first."""
    import cutesdr_amd as ca
    C, frames, fs, ave = 2, 40, 2e6, 5
    b = ca.FftBatch(C)
    b.set_params(n, False, 0.0, fs); b.set_ave(ave)
    refs = []
    for c in range(C):
        r = oracle.CFft(); r.SetFFTParams(n, False, 0.0, fs); r.SetFFTAve(ave); refs.append(r)
    for call in range(2):
        x = np.stack([tones_plus_noise(400 + c + 5 * call, frames * n, fs, [150e3 * (c + 1), -300e3]) for c in range(C)])
        b.put_display(x)
        for c in range(C):
            for k in range(frames):
                refs[c].PutInDisplayFFT(x[c, k * n:(k + 1) * n])
            assert b.total_count(c) == frames * (call + 1)
            assert_spectrum_close(b.ave_buf(c).astype(np.float64), refs[c].ave_buf())


@pytest.mark.parametrize("dbc", [-6.0, 12.5])
def test_display_spectrum_with_db_compensation(oracle, dbc):
    """SetFFTParams' dBCompensation (fft.cpp:186-188: it moves K_B and K_C, the offset and the floor inside the log) --
    every other test passes 0: the averaged bels and the screen mapping with it set, drop-in and batch form."""
    import cutesdr_amd as ca
    n, fs = 4096, 2e6
    f, r = ca.CFft(), oracle.CFft()
    b = ca.FftBatch(2)
    for o in (f, r):
        o.SetFFTParams(n, False, dbc, fs); o.SetFFTAve(3)
    b.set_params(n, False, dbc, fs); b.set_ave(3)
    frames = 6
    x = tones_plus_noise(77, frames * n, fs, [250e3, -410e3])
    for k in range(frames):
        assert f.PutInDisplayFFT(x[k * n:(k + 1) * n]) == r.PutInDisplayFFT(x[k * n:(k + 1) * n])
    b.put_display(np.stack([x, x[::-1].copy()]))
    want = r.ave_buf()
    assert_spectrum_close(f.ave_buf().astype(np.float64), want)
    assert_spectrum_close(b.ave_buf(0).astype(np.float64), want)
    zero = oracle.CFft(); zero.SetFFTParams(n, False, 0.0, fs); zero.SetFFTAve(3)
    for k in range(frames):
        zero.PutInDisplayFFT(x[k * n:(k + 1) * n])
    assert abs((want.max() - zero.ave_buf().max()) - dbc / 10.0) < 0.05      # the compensation is there: bels move by dbc / 10
    ovg, pg = f.GetScreenIntegerFFTData(400, 600, 0.0, -140.0, -800000, 800000)
    ovr, pr = r.GetScreenIntegerFFTData(400, 600, 0.0, -140.0, -800000, 800000)
    assert ovg == ovr and np.abs(pg - pr).max() <= 1
