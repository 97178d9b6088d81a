"""A SECOND, independent statement of the reference's dsp/ algorithms in numpy / scipy idiom -- TEST INFRASTRUCTURE.

oracle/cutesdr_oracle.c restates the reference function by function in C, loop for loop.  Nothing pins that
transcription beyond seven scalar anchors (tests/golden/survey_anchors.json): the reference ships no vectors and
cannot be built here.  This module is written from the reference's source text a second time, in a different shape --
library routines (scipy.signal.lfilter / upfirdn / windows.kaiser / freqz, numpy.fft, numpy.convolve, sliding-window
maxima) wherever the published algorithm is a standard one, closed forms where one exists, and
plain per-sample Python only for the genuinely recursive stages (AGC averagers, PLLs) -- so that a slip in either
transcription shows up as a disagreement between the two (tests/test_oracle_independent.py).  It shares no code with
oracle/ and is never imported by the product.

Every function cites the reference lines it follows (paths under /root/reference).
"""
import math
import numpy as np
from scipy import signal

TWO_PI = 2.0 * math.pi


# ----------------------------------------------------------------------------------------------- CIir (dsp/iir.cpp)
def rbj_biquad(kind, f0, q, fs):
    """Audio-EQ-cookbook biquads as dsp/iir.cpp:86-165 normalises them: (b, a) with a[0] = 1."""
    w0 = TWO_PI * f0 / fs
    alpha = math.sin(w0) / (2.0 * q)
    cw = math.cos(w0)
    a = np.array([1.0 + alpha, -2.0 * cw, 1.0 - alpha])
    b = {"LP": [(1 - cw) / 2, 1 - cw, (1 - cw) / 2],
         "HP": [(1 + cw) / 2, -(1 + cw), (1 + cw) / 2],
         "BP": [alpha, 0.0, -alpha],
         "BR": [1.0, -2.0 * cw, 1.0]}[kind]
    return np.array(b) / a[0], a / a[0]


class Biquad:
    """CIir::ProcessFilter (iir.cpp:171-201): a direct-form-II section; lfilter runs the transposed form, which is
    the same rational function (differences are rounding).  The complex overload filters re and im separately."""

    def __init__(self, kind, f0, q, fs):
        self.b, self.a = rbj_biquad(kind, f0, q, fs)
        self.zr = np.zeros(2)
        self.zi = np.zeros(2)

    def run(self, x):
        x = np.asarray(x)
        if np.iscomplexobj(x):
            yr, self.zr = signal.lfilter(self.b, self.a, x.real, zi=self.zr)
            yi, self.zi = signal.lfilter(self.b, self.a, x.imag, zi=self.zi)
            return yr + 1j * yi
        y, self.zr = signal.lfilter(self.b, self.a, x, zi=self.zr)
        return y


# ----------------------------------------------------------------------------------------------- CFir (dsp/fir.cpp)
def kaiser_beta(astop):
    """fir.cpp:186-192 (the Kaiser / Oppenheim-Schafer formula)"""
    if astop < 20.96:
        return 0.0
    if astop >= 50.0:
        return 0.1102 * (astop - 8.71)
    return 0.5842 * (astop - 20.96) ** 0.4 + 0.07886 * (astop - 20.96)


def kaiser_lowpass(scale, astop, fpass, fstop, fs):
    """CFir::InitLPFilter (fir.cpp:173-261): length from the Kaiser estimate, clamped to [3, 75]; ideal low-pass at
    the mean of the two edges times a Kaiser window.  scipy's window is I0(beta sqrt(1 - x^2)) / I0(beta) on the same
    grid x = (n - (N-1)/2) / ((N-1)/2); the reference sums the I0 series to 1e-9."""
    npass, nstop = fpass / fs, fstop / fs
    fc = (npass + nstop) / 2.0
    n = int((astop - 8.0) / (2.285 * TWO_PI * (nstop - npass)) + 1)
    n = max(3, min(75, n))
    k = np.arange(n) - 0.5 * (n - 1)
    return scale * 2.0 * fc * np.sinc(2.0 * fc * k) * signal.windows.kaiser(n, kaiser_beta(astop), sym=True)


def kaiser_highpass(scale, astop, fpass, fstop, fs):
    """CFir::InitHPFilter (fir.cpp:278-367): at most 74 taps, then forced odd; delta minus the low-pass"""
    npass, nstop = fpass / fs, fstop / fs
    fc = (npass + nstop) / 2.0
    n = int((astop - 8.0) / (2.285 * TWO_PI * (npass - nstop)) + 1)
    n = max(3, min(74, n)) | 1
    k = np.arange(n) - 0.5 * (n - 1)
    ideal = np.sinc(k) - 2.0 * fc * np.sinc(2.0 * fc * k)
    return scale * ideal * signal.windows.kaiser(n, kaiser_beta(astop), sym=True)


def hilbert_pair(lp, offset, fs):
    """CFir::GenerateHBFilter (fir.cpp:374-407): the low-pass shifted by `offset` Hz, I and Q tap sets"""
    k = np.arange(len(lp)) - 0.5 * (len(lp) - 1)
    return 2.0 * lp * np.cos(TWO_PI * offset / fs * k), 2.0 * lp * np.sin(TWO_PI * offset / fs * k)


class Fir:
    """CFir::ProcessFilter (fir.cpp:72-127): y[n] = sum_k h[k] x[n-k] over a circular buffer = lfilter(h, 1, x) with the
    state carried from call to call; the complex overload runs the I taps on .re and the Q taps on .im (no cross terms)"""

    def __init__(self, taps, qtaps=None):
        self.h = np.asarray(taps, dtype=float)
        self.hq = self.h if qtaps is None else np.asarray(qtaps, dtype=float)
        self.zr = np.zeros(len(self.h) - 1)
        self.zi = np.zeros(len(self.h) - 1)

    def run(self, x):
        x = np.asarray(x)
        if np.iscomplexobj(x):
            yr, self.zr = signal.lfilter(self.h, [1.0], x.real, zi=self.zr)
            yi, self.zi = signal.lfilter(self.hq, [1.0], x.imag, zi=self.zi)
            return yr + 1j * yi
        y, self.zr = signal.lfilter(self.h, [1.0], x, zi=self.zr)
        return y


# --------------------------------------------------------------------------- CDownConvert (dsp/downconvert.cpp)
def nco_mix(x, freq, fs, first_sample=0):
    """downconvert.cpp:186-247 in closed form: the phasor has phase (n+1) delta and the amplitude of
    the stabilising recurrence a' = a (1.95 - a^2), a0 = 1."""
    n = len(x)
    a = np.empty(n)
    g = 1.0
    for _ in range(first_sample):
        g = g * (1.95 - g * g)
    for i in range(n):
        a[i] = g
        g = g * (1.95 - g * g)
    ph = TWO_PI * freq / fs * (np.arange(n) + first_sample + 1)
    return x * a * np.exp(1j * ph)


def halfband_taps(even, length):
    """the tap vector of a half-band of `length` taps from its distinct even-index taps (dsp/filtercoef.h:34-424:
    zeros on the odd indices except the centre 0.5, symmetric)"""
    h = np.zeros(length)
    half = (length - 1) // 2
    idx = np.arange(0, half, 2)
    h[idx] = even[:len(idx)]
    h[length - 1 - idx] = even[:len(idx)]
    h[half] = 0.5
    return h


class DecimateBy2:
    """CHalfBandDecimateBy2 / CHalfBand11TapDecimateBy2 / CCicN3DecimateBy2 (downconvert.cpp:286-460) as ONE polyphase
    statement: y[j] = sum_k h[k] xe[2j + k], xe = [history | input] -- every second sample of the convolution of the
    history-extended stream with the (reversed) taps: scipy.signal.upfirdn(..., up=1, down=2).  History: the last
    len(h) - 1 inputs for a half band; two inputs for the CIC-3, h = (1, 3, 3, 1) / 8 (downconvert.cpp:453-454:
    y[j] = (x[2j+1] + 3 x[2j] + 3 x[2j-1] + x[2j-2]) / 8)."""

    def __init__(self, h):
        self.h = np.asarray(h, dtype=float)
        self.nh = 2 if len(self.h) == 4 else len(self.h) - 1
        self.hist = np.zeros(self.nh, dtype=complex)

    def run(self, x):
        x = np.asarray(x, dtype=complex)
        xe = np.concatenate([self.hist, x])
        L = len(self.h)
        # full[m'] = sum_k h[L-1-k] s[m'-k]; the wanted sum starts at xe[2j]: m' = 2j + L - 1 on the stream s.  upfirdn
        # keeps the even m', so a stream whose L - 1 is odd (the CIC) is shifted by one sample first.
        pad = (L - 1) & 1
        s = np.concatenate([np.zeros(pad, dtype=complex), xe])
        dec = signal.upfirdn(self.h[::-1], s, 1, 2)
        first = (L - 1 + pad) // 2
        y = dec[first:first + len(x) // 2]
        self.hist = xe[len(xe) - self.nh:]
        return y


def decimator_chain(in_rate, max_bw, hb_even, hb_len, hb_maxbw, cic3_maxbw):
    """CDownConvert::SetDataRate (downconvert.cpp:127-166): stage kinds and output rate"""
    f, kinds = in_rate, []
    while f > max_bw / hb_maxbw[-1] and f > 15800.0 and len(kinds) < 9:
        if f >= max_bw / cic3_maxbw:
            kinds.append(3)
        else:
            kinds.append(next(hb_len[i] for i, m in enumerate(hb_maxbw) if f >= max_bw / m))
        f /= 2.0
    return kinds, f


# ----------------------------------------------------------------------------------------- CFastFIR (dsp/fastfir.cpp)
def fastfir_taps(flo, fhi, offset, fs, nfft):
    """CFastFIR::SetupParameters (fastfir.cpp:178-259): P = N/2 + 1 complex taps (without the 1/N the reference folds
    in for its unscaled inverse transform): windowed sinc at (fhi - flo)/2, shifted to the band centre."""
    p = nfft // 2 + 1
    i = np.arange(p)
    win = (0.3635819 - 0.4891775 * np.cos(TWO_PI * i / (p - 1)) + 0.1365995 * np.cos(2 * TWO_PI * i / (p - 1))
           - 0.0106411 * np.cos(3 * TWO_PI * i / (p - 1)))          # fastfir.cpp:93-101, Blackman-Nuttall
    nfl, nfh = (flo + offset) / fs, (fhi + offset) / fs
    nfc = (nfh - nfl) / 2.0
    x = i - 0.5 * (p - 1)
    z = 2.0 * nfc * np.sinc(2.0 * nfc * x) * win                # the centre tap is unwindowed there; the window is 1 at the centre
    return z * np.exp(1j * TWO_PI * (nfh + nfl) / 2.0 * x)


def fastfir_stream(x, taps, nfft):
    """CFastFIR::ProcessData (fastfir.cpp:268-306): overlap-save with a zero first overlap IS the linear convolution
    of the stream with the taps, delivered in whole hops of L = N - P + 1 samples."""
    hop = nfft - len(taps) + 1
    n_out = (len(x) // hop) * hop
    return np.convolve(x, taps)[:n_out]


# ----------------------------------------------------------------------------------------------- CFft (dsp/fft.cpp)
class DisplayFft:
    """CFft::SetFFTParams / PutInDisplayFFT / the averaging of CpxFFT (fft.cpp:118-243, 267-288, 562-589) with
    numpy.fft: Hann x 2, I/Q swapped into a positive-exponent transform = the conventional spectrum, fft-shifted;
    running mean over AveSize frames in the reference's sum / replace-the-mean form; bels."""

    def __init__(self, n, db_comp, fs, ave):
        self.n, self.fs, self.ave = n, fs, max(1, ave)
        kb = db_comp - 20.0 * math.log10(n * 32767.0 / 2.0)
        self.kc = 10.0 ** ((-220.0 - kb) / 10.0)
        self.kb = kb / 10.0
        self.win = 2.0 * (0.5 - 0.5 * np.cos(TWO_PI * np.arange(n) / (n - 1)))
        self.sum = np.zeros(n)
        self.mean = np.zeros(n)
        self.count = 0
        self.total = 0
        self.overload = False
        self.bels = np.zeros(n)

    def put(self, x):
        x = np.asarray(x, dtype=complex)
        self.overload = bool((x.real > 32000).any())
        buf = self.win * (x.imag + 1j * x.real)                   # fft.cpp:280-281
        spec = np.fft.ifft(buf) * self.n                          # positive exponent, unscaled
        p = np.fft.fftshift(np.abs(spec) ** 2)                    # natural bin k -> display index (k + N/2) mod N
        self.count = min(self.count + 1, self.ave)                # fft.cpp:515-517
        self.total += 1
        self.sum = self.sum + p if self.total <= self.ave else self.sum - self.mean + p
        self.mean = self.sum / self.count
        self.bels = np.log10(self.mean + self.kc) + self.kb
        return self.total

    def screen(self, max_h, max_w, max_db, min_db, start_hz, stop_hz):
        """CFft::GetScreenIntegerFFTData (fft.cpp:308-410, m_Invert off), as array operations.  More bins than pixels:
        every pixel shows its STRONGEST bin (smallest y) -- a group-wise minimum, which is what the reference's carried
        `ymax` computes because equal x are consecutive; pixels no bin maps to are not written (-1 here).  Otherwise
        pixel x shows bin lo + x (hi - lo) / width."""
        n = self.n
        gain = -10.0 / (max_db - min_db)
        lo = int(start_hz * float(n) / self.fs) + n // 2
        hi = int(stop_hz * float(n) / self.fs) + n // 2
        lo, hi = min(max(lo, 0), n - 1), min(max(hi, 0), n - 1)

        def level(v):
            return np.clip(np.trunc(max_h * gain * (v - max_db / 10.0)).astype(np.int64), 0, max_h)
        out = np.full(max(max_w, 1) + 1, -1, dtype=np.int64)
        if hi - lo > max_w:
            bins = np.arange(lo, hi + 1)
            xs = ((bins - lo) * max_w) // (hi - lo)
            ys = level(self.bels[bins])
            big = np.full(len(out), np.iinfo(np.int64).max)
            np.minimum.at(big, xs, ys)
            hit = np.zeros(len(out), dtype=bool); hit[xs] = True
            out[hit] = big[hit]
        else:
            xs = np.arange(max_w)
            out[:max_w] = level(self.bels[lo + (xs * (hi - lo)) // max(max_w, 1)])
        return out[:max(max_w, 1)]


# --------------------------------------------------------------------------------------------- CSMeter (dsp/smeter.cpp)
def smeter(x, fs, state=None):
    """CSMeter::ProcessData (smeter.cpp:62-93): attack average (10 ms), decay average (500 ms) snapped up to the attack
    average whenever that is higher, peak hold.  Returns (ave + 5 dB, peak, state)."""
    aa = 1.0 - math.exp(-1.0 / (fs * 0.01))
    da = 1.0 - math.exp(-1.0 / (fs * 0.5))
    att, dec, peak = state or (-120.0, -120.0, 0.0)
    db = 10.0 * np.log10((x.real ** 2 + x.imag ** 2) / (32767.0 * 32767.0) + 1e-50)
    ave = att
    for m in db:
        att += aa * (m - att)
        dec += da * (m - dec)
        if att > dec:
            dec = att
        ave = dec if dec >= att else att
        if m > peak:
            peak = m
    return ave + 5.0, peak, (att, dec, peak)


# ----------------------------------------------------------------------------------------------- CAgc (dsp/agc.cpp)
class Agc:
    """CAgc::SetParameters / ProcessData (agc.cpp:104-167, 174-296).  The window peak of the reference (compare, equality
    test, rescan: :210-231) is the maximum of the last WindowSamples log magnitudes: a sliding-window maximum over
    [history | block].  The averagers are the genuinely sequential part: per-sample Python."""

    def __init__(self, on, hang, thresh, manual, slope, decay_ms, fs):
        self.on, self.hang = on, hang
        self.manual = 32767.0 * 10.0 ** (-(100.0 - manual) / 20.0)
        self.knee = thresh / 20.0
        self.gslope = slope / 100.0
        self.fixed = 0.7 * 10.0 ** (self.knee * (self.gslope - 1.0))
        e = lambda tc: 1.0 - math.exp(-1.0 / (fs * tc))
        self.ar, self.af = e(0.002), e(0.005)
        self.dr = e(decay_ms * 0.001 * 0.3)
        self.df = e(0.05) if hang else e(decay_ms * 0.001)
        self.hang_time = int(fs * decay_ms * 0.001)
        self.delay = min(int(fs * 0.015), 2047)
        self.window = int(fs * 0.018)
        self.dly = np.zeros(self.delay, dtype=complex)
        self.mags = np.full(self.window, -16.0)                  # the ring starts at -16 (agc.cpp:121-136)
        self.att = self.dec = -5.0
        self.timer = 0

    def run(self, x):
        x = np.asarray(x, dtype=complex)
        if not self.on:
            return self.manual * x
        n = len(x)
        mag = np.log10(np.maximum(np.abs(x.real), np.abs(x.imag)) + 3.2767e-4) - math.log10(32767.0)
        ext = np.concatenate([self.mags, mag])
        # the window peak: maximum of the last W log magnitudes, the current one included
        W = self.window
        peak = np.lib.stride_tricks.sliding_window_view(ext, W).max(axis=1)[1:n + 1]
        self.mags = ext[len(ext) - W:]
        delayed = np.concatenate([self.dly, x])[:n]
        self.dly = np.concatenate([self.dly, x])[n:]
        out = np.empty(n, dtype=complex)
        att, dec, timer = self.att, self.dec, self.timer
        for i in range(n):
            pk = peak[i]
            att += (self.ar if pk > att else self.af) * (pk - att)
            if pk > dec:
                dec += self.dr * (pk - dec)
                timer = 0
            elif self.hang and timer < self.hang_time:
                timer += 1
            else:
                dec += self.df * (pk - dec)
            m = att if att > dec else dec
            g = self.fixed if m <= self.knee else 0.7 * 10.0 ** (m * (self.gslope - 1.0))
            out[i] = delayed[i] * g
        self.att, self.dec, self.timer = att, dec, timer
        return out


# ------------------------------------------------------------------------------------- demodulators (dsp/*demod.cpp)
def dc_block(x, state=0.0):
    """H(z) = (1 - z^-1) / (1 - 0.99 z^-1) (amdemod.cpp:70-80, samdemod.cpp:100-104) as lfilter, the reference's z1
    being the filter's internal state"""
    y, z = signal.lfilter([1.0, -1.0], [1.0, -0.99], x, zi=[state])
    return y, z[0]


class AmDemod:
    """CAmDemod (amdemod.cpp:50-104): envelope -> DC block -> Kaiser low-pass (50 dB, pass = bw, stop = 1.8 bw)"""

    def __init__(self, fs, bw=10000.0):
        self.fs = fs
        self.z = 0.0
        self.fir = Fir(kaiser_lowpass(1.0, 50.0, bw, bw * 1.8, fs))

    def set_bandwidth(self, bw):
        self.fir = Fir(kaiser_lowpass(1.0, 50.0, bw, bw * 1.8, self.fs))

    def run(self, x):
        # lfilter's state for this section is (0.99 - 1) * z1 ... keep the reference's own variable instead
        env = np.abs(x)
        y = np.empty(len(env))
        z1 = self.z
        for i, m in enumerate(env):
            z0 = m + z1 * 0.99
            y[i] = z0 - z1
            z1 = z0
        self.z = z1
        return self.fir.run(y)


class Pll:
    """the second-order loop CFmDemod and CSamDemod share (fmdemod.cpp:62-76, 166-177; samdemod.cpp:54-66, 83-97)"""

    def __init__(self, fs, bw, zeta, limit):
        norm = TWO_PI / fs
        self.hi, self.lo = limit * norm, -limit * norm
        self.alpha = 2.0 * zeta * bw * norm
        self.beta = self.alpha * self.alpha / (4.0 * zeta * zeta)
        self.phase = self.freq = 0.0


class FmDemod:
    """CFmDemod (fmdemod.cpp:62-236): PLL discriminator, DC removal of the loop frequency, then per CALL the noise
    squelch: Kaiser high-pass of the audio -> |.| -> 20 ms average -> threshold with +-100 hysteresis -> zeros or the
    3 kHz biquad."""

    def __init__(self, fs):
        self.fs = fs
        self.p = Pll(fs, 6000.0, 0.707, 6000.0)
        self.gain = 25000.0 / self.p.hi
        self.dc = 0.0
        self.dc_alpha = 1.0 - math.exp(-1.0 / (fs * 0.01))
        self.sq_alpha = 1.0 - math.exp(-1.0 / (fs * 0.02))
        self.sq_ave = 0.0
        self.squelched = True
        self.thresh = None                                       # SetSquelch not called: the member is uninitialised there
        self.hp_freq = 3000.0
        self.lp = Biquad("LP", 3000.0, 1.0, fs)
        self.hp = Fir(kaiser_highpass(1.0, 50.0, self.hp_freq, self.hp_freq * 0.6, fs))

    def set_squelch(self, value):
        self.thresh = 5000.0 - (5000.0 * value) / 99

    def run(self, x, fm_bw):
        if fm_bw != self.hp_freq:
            self.hp_freq = fm_bw
            self.hp = Fir(kaiser_highpass(1.0, 50.0, fm_bw, fm_bw * 0.6, self.fs))
        p = self.p
        out = np.empty(len(x))
        ph, fr, dc = p.phase, p.freq, self.dc
        for i, v in enumerate(x):
            r = v * complex(math.cos(ph), math.sin(ph))
            err = -math.atan2(r.imag, r.real)
            fr = min(max(fr + p.beta * err, p.lo), p.hi)
            ph += fr + p.alpha * err
            dc = (1.0 - self.dc_alpha) * dc + self.dc_alpha * fr
            out[i] = (fr - dc) * self.gain
        p.phase, p.freq, self.dc = math.fmod(ph, TWO_PI), fr, dc
        noise = np.abs(self.hp.run(out))
        # one-pole average over the call: lfilter with the carried state
        y, z = signal.lfilter([self.sq_alpha], [1.0, -(1.0 - self.sq_alpha)], noise, zi=[(1.0 - self.sq_alpha) * self.sq_ave])
        self.sq_ave = y[-1] if len(y) else self.sq_ave
        if self.thresh == 0:
            self.squelched = True
        elif self.squelched:
            if self.sq_ave < self.thresh - 100.0:
                self.squelched = False
        elif self.sq_ave >= self.thresh + 100.0:
            self.squelched = True
        return np.zeros(len(x)) if self.squelched else self.lp.run(out)


class SamDemod:
    """CSamDemod (samdemod.cpp:54-158): PLL on the carrier, synchronous I (mono) or I/Q through the Hilbert pair and
    the sum / difference (stereo).  Mono rotates by e^{-j phi} and takes +atan2, stereo by e^{+j phi} and -atan2."""

    def __init__(self, fs):
        self.fs = fs
        self.p = Pll(fs, 100.0, 0.707, 1000.0)
        self.z1 = self.y1 = 0.0
        i, q = hilbert_pair(kaiser_lowpass(1.0, 40.0, 4500.0, 5500.0, fs), 5000.0, fs)
        self.fir = Fir(i, q)

    def run(self, x, stereo=False):
        p = self.p
        sgn = 1.0 if stereo else -1.0
        ph, fr = p.phase, p.freq
        re = np.empty(len(x)); im = np.empty(len(x))
        for i, v in enumerate(x):
            r = v * complex(math.cos(ph), sgn * math.sin(ph))
            err = -sgn * math.atan2(r.imag, r.real)
            fr = min(max(fr + p.beta * err, p.lo), p.hi)
            ph += fr + p.alpha * err
            re[i], im[i] = r.real, r.imag
        p.phase, p.freq = math.fmod(ph, TWO_PI), fr
        # the DC blockers carry the reference's own z1 / y1 (the recursion's internal variable)
        def block(v, z1):
            y = np.empty(len(v))
            for i, m in enumerate(v):
                z0 = m + z1 * 0.99
                y[i] = z0 - z1
                z1 = z0
            return y, z1
        a, self.z1 = block(re, self.z1)
        if not stereo:
            return a
        b, self.y1 = block(im, self.y1)
        f = self.fir.run(a + 1j * b)
        return (f.real + f.imag) + 1j * (f.real - f.imag)


# ----------------------------------------------------------------------------- CFractResampler (dsp/fractresampler.cpp)
SINC_PTS, SINC_PERIODS = 10000, 28
_sinc_table = None


def sinc_table():
    """fractresampler.cpp:85-135: sinc(pi (i - 140000) / 10000) x Blackman-Harris over 280 001 points"""
    global _sinc_table
    if _sinc_table is None:
        n = SINC_PERIODS * SINC_PTS + 1
        i = np.arange(n)
        win = (0.35875 - 0.48829 * np.cos(TWO_PI * i / (n - 1)) + 0.14128 * np.cos(2 * TWO_PI * i / (n - 1))
               - 0.01168 * np.cos(3 * TWO_PI * i / (n - 1)))
        t = win * np.sinc((i - n // 2) / float(SINC_PTS))
        t[n // 2] = 1.0
        _sinc_table = t
    return _sinc_table


class Resampler:
    """CFractResampler::Resample (fractresampler.cpp:144-184): output k sits at input time t_k (accumulated in floating
    point exactly as the reference does, so the truncations agree) and is the 28-tap dot product of the inputs
    t+1 .. t+28 with the table sampled at floor((j - t_k) * 10000) -- evaluated here as one gather per call."""

    def __init__(self):
        self.tail = np.zeros(SINC_PERIODS, dtype=complex)
        self.t = 0.0

    def run(self, x, rate):
        x = np.asarray(x, dtype=complex)
        buf = np.concatenate([self.tail, x])
        times = []
        t = self.t
        while int(t) < len(x):
            times.append(t)
            t += rate
        self.t = t - float(len(x))
        self.tail = buf[len(x):len(x) + SINC_PERIODS]
        if not times:
            return np.zeros(0, dtype=complex)
        tk = np.array(times)
        base = tk.astype(np.int64)
        j = base[:, None] + np.arange(1, SINC_PERIODS + 1)[None, :]
        idx = ((j - tk[:, None]) * float(SINC_PTS)).astype(np.int64)
        return (buf[j] * sinc_table()[idx]).sum(axis=1)
