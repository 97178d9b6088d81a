"""ctypes binding of the fp64 CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under cutesdr_amd/ does.  See oracle/cutesdr_oracle.h for pinning status.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcutesdr_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "cutesdr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def constants():
    """{"<reference file>:<#define>": value} as the oracle's arithmetic uses them (orc_constants)"""
    L = lib()
    n = L.orc_constants(None, None, 0)
    names = (C.c_char_p * n)()
    vals = (C.c_double * n)()
    L.orc_constants(C.cast(names, C.c_void_p), C.cast(vals, C.c_void_p), n)
    return {names[i].decode(): float(vals[i]) for i in range(n)}


class DemodInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "HiCut", "HiCutmin", "HiCutmax", "LowCut", "LowCutmin", "LowCutmax",
        "FilterClickResolution", "Offset", "SquelchValue",
        "AgcSlope", "AgcThresh", "AgcManualGain", "AgcDecay",
        "AgcOn", "AgcHangOn", "Symetric")]


P = C.c_void_p
D = C.c_double
I = C.c_int


def _declare(L):
    def f(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)
    f("orc_fft", None, I, I, P)
    f("orc_constants", I, P, P, I)
    f("orc_cfft_new", P); f("orc_cfft_free", None, P)
    f("orc_cfft_set_params", None, P, I, I, D, D)
    f("orc_cfft_set_ave", None, P, I); f("orc_cfft_reset", None, P)
    f("orc_cfft_put_display", I, P, I, P)
    f("orc_cfft_get_screen", I, P, I, I, D, D, I, I, P)
    f("orc_plotter_color_table", None, P)
    f("orc_plotter_waterfall_line", I, P, I, D, D, I, I, P, P)
    f("orc_cfft_fwd", None, P, P); f("orc_cfft_rev", None, P, P)
    f("orc_cfft_size", I, P); f("orc_cfft_avebuf", P, P)
    f("orc_fastfir_new", P, I); f("orc_fastfir_free", None, P)
    f("orc_fastfir_set_faithful", None, P, I)
    f("orc_fastfir_setup", I, P, D, D, D, D)
    f("orc_fastfir_process", I, P, I, P, P)
    f("orc_fastfir_coef", P, P)
    f("orc_downconv_new", P); f("orc_downconv_free", None, P)
    f("orc_downconv_set_cw_offset", None, P, D)
    f("orc_downconv_set_frequency", None, P, D)
    f("orc_downconv_set_data_rate", D, P, D, D)
    f("orc_downconv_process", I, P, I, P, P)
    f("orc_downconv_stages", I, P, P)
    f("orc_downconv_nco_freq", D, P)
    f("orc_fir_new", P); f("orc_fir_free", None, P)
    f("orc_fir_init_const", None, P, I, P)
    f("orc_fir_init_lp", I, P, D, D, D, D, D)
    f("orc_fir_init_hp", I, P, D, D, D, D, D)
    f("orc_fir_gen_hilbert", None, P, D)
    f("orc_fir_process_real", None, P, I, P, P)
    f("orc_fir_process_cpx", None, P, I, P, P)
    f("orc_fir_taps", I, P, P, P, P)
    f("orc_iir_new", P); f("orc_iir_free", None, P)
    f("orc_iir_init", None, P, I, D, D, D)
    f("orc_iir_process_real", None, P, I, P, P)
    f("orc_iir_process_cpx", None, P, I, P, P)
    f("orc_iir_coefs", None, P, P)
    f("orc_agc_new", P); f("orc_agc_free", None, P)
    f("orc_agc_set", None, P, I, I, I, I, I, I, D)
    f("orc_agc_process_cpx", None, P, I, P, P)
    f("orc_agc_process_real", None, P, I, P, P)
    f("orc_noiseproc_new", P); f("orc_noiseproc_free", None, P)
    f("orc_noiseproc_setup", I, P, I, D, D, D); f("orc_noiseproc_process", None, P, I, P, P)
    f("orc_unpack_packet", I, P, I, P); f("orc_spurcal", None, P, I, P)
    f("orc_soundsink_new", P, I); f("orc_soundsink_free", None, P)
    f("orc_soundsink_change_rate", None, P, D); f("orc_soundsink_set_volume", None, P, I)
    f("orc_soundsink_set_blocking", None, P, I)
    f("orc_soundsink_put", I, P, I, P); f("orc_soundsink_get", None, P, I, P)
    f("orc_soundsink_rate_correction", D, P); f("orc_soundsink_ave_level", D, P)
    f("orc_soundsink_level", I, P); f("orc_soundsink_ppm", I, P)
    f("orc_smeter_new", P); f("orc_smeter_free", None, P)
    f("orc_smeter_process", None, P, I, P, D)
    f("orc_smeter_peak", D, P); f("orc_smeter_ave", D, P)
    f("orc_amdemod_new", P, D); f("orc_amdemod_free", None, P)
    f("orc_amdemod_set_bandwidth", None, P, D)
    f("orc_amdemod_process_mono", I, P, I, P, P)
    f("orc_amdemod_process_stereo", I, P, I, P, P)
    f("orc_samdemod_new", P, D); f("orc_samdemod_free", None, P)
    f("orc_samdemod_process_mono", I, P, I, P, P)
    f("orc_samdemod_process_stereo", I, P, I, P, P)
    f("orc_fmdemod_new", P, D); f("orc_fmdemod_free", None, P)
    f("orc_fmdemod_set_squelch", None, P, I)
    f("orc_fmdemod_process_mono", I, P, I, D, P, P)
    f("orc_fmdemod_process_stereo", I, P, I, D, P, P)
    f("orc_fmdemod_squelched", I, P)
    f("orc_ssbdemod_process_mono", I, I, P, P)
    f("orc_ssbdemod_process_stereo", I, I, P, P)
    f("orc_resampler_new", P); f("orc_resampler_free", None, P)
    f("orc_resampler_init", None, P, I)
    f("orc_resampler_real", I, P, I, D, P, P)
    f("orc_resampler_cpx", I, P, I, D, P, P)
    f("orc_resampler_real_i16", I, P, I, D, P, P, D)
    f("orc_resampler_cpx_i16", I, P, I, D, P, P, D)
    f("orc_demod_new", P, I); f("orc_demod_free", None, P)
    f("orc_demod_set_input_rate", None, P, D)
    f("orc_demod_set_demod", None, P, I, P)
    f("orc_demod_set_freq", None, P, D)
    f("orc_demod_output_rate", D, P)
    f("orc_demod_smeter_peak", D, P); f("orc_demod_smeter_ave", D, P)
    f("orc_demod_buf_limit", I, P)
    f("orc_demod_process_mono", I, P, I, P, P)
    f("orc_demod_process_stereo", I, P, I, P, P)
    f("orc_demod_process_mono_append", I, P, I, P, P)
    f("orc_demod_enable_taps", None, P, I)
    f("orc_demod_perturb_filter_output", None, P, I, D, C.c_ulonglong)
    f("orc_demod_tap_len", I, P, I)
    f("orc_demod_tap_data", P, P, I)
    f("orc_demod_clear_taps", None, P)


def _c128(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fft(x, sign=+1):
    a = _c128(x).copy()
    lib().orc_fft(len(a), sign, _ptr(a))
    return a


class _Handle:
    _new = None
    _free = None

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            getattr(lib(), self._free)(h)
            self.h = None


def plotter_color_table():
    """CPlotter's 256-entry waterfall palette (gui/plotter.cpp:67-83) as 0xFFRRGGBB"""
    t = np.zeros(256, dtype=np.uint32)
    lib().orc_plotter_color_table(_ptr(t))
    return t


class CFft(_Handle):
    _free = "orc_cfft_free"

    def __init__(self):
        self.h = lib().orc_cfft_new()

    def SetFFTParams(self, size, invert, db_comp, fs):
        lib().orc_cfft_set_params(self.h, size, int(invert), db_comp, fs)

    def SetFFTAve(self, ave):
        lib().orc_cfft_set_ave(self.h, ave)

    def ResetFFT(self):
        lib().orc_cfft_reset(self.h)

    def PutInDisplayFFT(self, x):
        a = _c128(x)
        return lib().orc_cfft_put_display(self.h, len(a), _ptr(a))

    def GetScreenIntegerFFTData(self, max_h, max_w, max_db, min_db, start_hz, stop_hz):
        out = np.zeros(max(max_w, 1) + 1, dtype=np.int32)     # the reference's loop writes OutBuf[MaxWidth] too
        ov = lib().orc_cfft_get_screen(self.h, max_h, max_w, max_db, min_db, start_hz, stop_hz, _ptr(out))
        return bool(ov), out[:max(max_w, 1)].copy()

    def WaterfallLine(self, max_w, max_db, min_db, start_hz, stop_hz, fill=0):
        """CPlotter::draw's new waterfall line (gui/plotter.cpp:425-441) -> (overload, 0xFFRRGGBB pixels); pixels no
        bin maps to come back as `fill`"""
        lv = np.full(max(max_w, 1) + 1, -1, dtype=np.int32)
        rgb = np.full(max(max_w, 1), fill, dtype=np.uint32)
        ov = lib().orc_plotter_waterfall_line(self.h, max_w, max_db, min_db, start_hz, stop_hz, _ptr(lv), _ptr(rgb))
        return bool(ov), rgb

    def FwdFFT(self, x):
        a = _c128(x).copy(); lib().orc_cfft_fwd(self.h, _ptr(a)); return a

    def RevFFT(self, x):
        a = _c128(x).copy(); lib().orc_cfft_rev(self.h, _ptr(a)); return a

    def ave_buf(self):
        n = lib().orc_cfft_size(self.h)
        p = lib().orc_cfft_avebuf(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)).copy()


class CFastFIR(_Handle):
    _free = "orc_fastfir_free"

    def __init__(self, fft_size=2048):
        self.n = fft_size
        self.h = lib().orc_fastfir_new(fft_size)

    def set_faithful(self, on):
        lib().orc_fastfir_set_faithful(self.h, int(on))

    def SetupParameters(self, flo, fhi, offset, fs):
        return lib().orc_fastfir_setup(self.h, flo, fhi, offset, fs)

    def ProcessData(self, x):
        a = _c128(x)
        out = np.zeros(len(a) + self.n, dtype=np.complex128)
        k = lib().orc_fastfir_process(self.h, len(a), _ptr(a), _ptr(out))
        return out[:k]

    def coef(self):
        p = lib().orc_fastfir_coef(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(2 * self.n,)).copy().view(np.complex128)


class CDownConvert(_Handle):
    _free = "orc_downconv_free"

    def __init__(self):
        self.h = lib().orc_downconv_new()

    def SetCwOffset(self, off):
        lib().orc_downconv_set_cw_offset(self.h, off)

    def SetFrequency(self, f):
        lib().orc_downconv_set_frequency(self.h, f)

    def SetDataRate(self, in_rate, max_bw):
        return lib().orc_downconv_set_data_rate(self.h, in_rate, max_bw)

    def ProcessData(self, x):
        a = _c128(x).copy()
        out = np.zeros(len(a), dtype=np.complex128)
        k = lib().orc_downconv_process(self.h, len(a), _ptr(a), _ptr(out))
        return out[:k]

    def stages(self):
        codes = np.zeros(16, dtype=np.int32)
        n = lib().orc_downconv_stages(self.h, _ptr(codes))
        return list(codes[:n])

    def nco_freq(self):
        return lib().orc_downconv_nco_freq(self.h)


class CFir(_Handle):
    _free = "orc_fir_free"

    def __init__(self):
        self.h = lib().orc_fir_new()

    def InitConstFir(self, coef):
        c = np.ascontiguousarray(coef, dtype=np.float64)
        lib().orc_fir_init_const(self.h, len(c), _ptr(c))

    def InitLPFilter(self, scale, astop, fpass, fstop, fs):
        return lib().orc_fir_init_lp(self.h, scale, astop, fpass, fstop, fs)

    def InitHPFilter(self, scale, astop, fpass, fstop, fs):
        return lib().orc_fir_init_hp(self.h, scale, astop, fpass, fstop, fs)

    def GenerateHBFilter(self, off):
        lib().orc_fir_gen_hilbert(self.h, off)

    def taps(self):
        c = np.zeros(75); i = np.zeros(75); q = np.zeros(75)
        n = lib().orc_fir_taps(self.h, _ptr(c), _ptr(i), _ptr(q))
        return c[:n], i[:n], q[:n]

    def ProcessFilter(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.zeros_like(a)
            lib().orc_fir_process_cpx(self.h, len(a), _ptr(a), _ptr(out))
        else:
            a = np.ascontiguousarray(x, dtype=np.float64); out = np.zeros_like(a)
            lib().orc_fir_process_real(self.h, len(a), _ptr(a), _ptr(out))
        return out


class CIir(_Handle):
    _free = "orc_iir_free"
    KIND = {"LP": 0, "HP": 1, "BP": 2, "BR": 3}

    def __init__(self):
        self.h = lib().orc_iir_new()

    def Init(self, kind, f0, q, fs):
        lib().orc_iir_init(self.h, self.KIND[kind], f0, q, fs)

    def coefs(self):
        c = np.zeros(5); lib().orc_iir_coefs(self.h, _ptr(c)); return c

    def ProcessFilter(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.zeros_like(a)
            lib().orc_iir_process_cpx(self.h, len(a), _ptr(a), _ptr(out))
        else:
            a = np.ascontiguousarray(x, dtype=np.float64); out = np.zeros_like(a)
            lib().orc_iir_process_real(self.h, len(a), _ptr(a), _ptr(out))
        return out


class CAgc(_Handle):
    _free = "orc_agc_free"

    def __init__(self):
        self.h = lib().orc_agc_new()

    def SetParameters(self, on, hang, thresh, manual, slope, decay, fs):
        lib().orc_agc_set(self.h, int(on), int(hang), thresh, manual, slope, decay, fs)

    def ProcessData(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.zeros_like(a)
            lib().orc_agc_process_cpx(self.h, len(a), _ptr(a), _ptr(out))
        else:
            a = np.ascontiguousarray(x, dtype=np.float64); out = np.zeros_like(a)
            lib().orc_agc_process_real(self.h, len(a), _ptr(a), _ptr(out))
        return out


class CSMeter(_Handle):
    _free = "orc_smeter_free"

    def __init__(self):
        self.h = lib().orc_smeter_new()

    def ProcessData(self, x, fs):
        a = _c128(x); lib().orc_smeter_process(self.h, len(a), _ptr(a), fs)

    def GetPeak(self):
        return lib().orc_smeter_peak(self.h)

    def GetAve(self):
        return lib().orc_smeter_ave(self.h)


class CNoiseProc(_Handle):
    """dsp/noiseproc.h:23-58"""
    _free = "orc_noiseproc_free"

    def __init__(self):
        self.h = lib().orc_noiseproc_new()

    def SetupBlanker(self, On, Threshold, Width, SampleRate):
        if lib().orc_noiseproc_setup(self.h, int(On), Threshold, Width, SampleRate) < 0:
            raise ValueError("sample rate too high for the reference's 32768-entry average buffer")

    def ProcessBlanker(self, x):
        a = _c128(x)
        out = a.copy()                                   # off: the data passes untouched (in-place call)
        lib().orc_noiseproc_process(self.h, len(a), _ptr(a), _ptr(out))
        return out


def unpack_packets(raw, pkt_len):
    """raw: uint8 [npackets, pkt_len] -> complex128 samples (interface/netiobase.cpp:479-527)"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, pkt_len)
    per = {1028: 256, 1444: 240}[pkt_len]
    out = np.empty(raw.shape[0] * per, dtype=np.complex128)
    for k in range(raw.shape[0]):
        row = np.ascontiguousarray(raw[k])
        n = lib().orc_unpack_packet(_ptr(row), pkt_len, C.c_void_p(out.ctypes.data + 16 * per * k))
        assert n == per
    return out


def spurcal(dc, x):
    """NcoSpurCalibrate running means (interface/sdrinterface.cpp:829-848); dc = [I, Q] updated in place"""
    a = _c128(x); d = np.ascontiguousarray(dc, dtype=np.float64)
    lib().orc_spurcal(_ptr(d), 2 * len(a), _ptr(a))
    return d


class CAmDemod(_Handle):
    _free = "orc_amdemod_free"

    def __init__(self, fs):
        self.h = lib().orc_amdemod_new(fs)

    def SetBandwidth(self, bw):
        lib().orc_amdemod_set_bandwidth(self.h, bw)

    def ProcessData(self, x, stereo=False):
        a = _c128(x)
        if stereo:
            out = np.zeros_like(a); lib().orc_amdemod_process_stereo(self.h, len(a), _ptr(a), _ptr(out))
        else:
            out = np.zeros(len(a)); lib().orc_amdemod_process_mono(self.h, len(a), _ptr(a), _ptr(out))
        return out


class CSamDemod(_Handle):
    _free = "orc_samdemod_free"

    def __init__(self, fs):
        self.h = lib().orc_samdemod_new(fs)

    def ProcessData(self, x, stereo=False):
        a = _c128(x)
        if stereo:
            out = np.zeros_like(a); lib().orc_samdemod_process_stereo(self.h, len(a), _ptr(a), _ptr(out))
        else:
            out = np.zeros(len(a)); lib().orc_samdemod_process_mono(self.h, len(a), _ptr(a), _ptr(out))
        return out


class CFmDemod(_Handle):
    _free = "orc_fmdemod_free"

    def __init__(self, fs):
        self.h = lib().orc_fmdemod_new(fs)

    def SetSquelch(self, v):
        lib().orc_fmdemod_set_squelch(self.h, v)

    def squelched(self):
        return bool(lib().orc_fmdemod_squelched(self.h))

    def ProcessData(self, x, fm_bw, stereo=False):
        a = _c128(x)
        if stereo:
            out = np.zeros_like(a); lib().orc_fmdemod_process_stereo(self.h, len(a), fm_bw, _ptr(a), _ptr(out))
        else:
            out = np.zeros(len(a)); lib().orc_fmdemod_process_mono(self.h, len(a), fm_bw, _ptr(a), _ptr(out))
        return out


def ssb_demod(x, stereo=False):
    a = _c128(x)
    if stereo:
        out = np.zeros_like(a); lib().orc_ssbdemod_process_stereo(len(a), _ptr(a), _ptr(out))
    else:
        out = np.zeros(len(a)); lib().orc_ssbdemod_process_mono(len(a), _ptr(a), _ptr(out))
    return out


class CFractResampler(_Handle):
    _free = "orc_resampler_free"

    def __init__(self):
        self.h = lib().orc_resampler_new()

    def Init(self, max_input):
        lib().orc_resampler_init(self.h, max_input)

    def Resample(self, x, rate, gain=None):
        cap = int(len(x) / rate) + 8
        if np.iscomplexobj(x):
            a = _c128(x)
            if gain is None:
                out = np.zeros(cap, dtype=np.complex128)
                k = lib().orc_resampler_cpx(self.h, len(a), rate, _ptr(a), _ptr(out))
                return out[:k]
            out = np.zeros(2 * cap, dtype=np.int16)
            k = lib().orc_resampler_cpx_i16(self.h, len(a), rate, _ptr(a), _ptr(out), gain)
            return out[:2 * k].reshape(-1, 2)
        a = np.ascontiguousarray(x, dtype=np.float64)
        if gain is None:
            out = np.zeros(cap)
            k = lib().orc_resampler_real(self.h, len(a), rate, _ptr(a), _ptr(out))
            return out[:k]
        out = np.zeros(cap, dtype=np.int16)
        k = lib().orc_resampler_real_i16(self.h, len(a), rate, _ptr(a), _ptr(out), gain)
        return out[:k]


DEMOD_AM, DEMOD_SAM, DEMOD_FM, DEMOD_USB, DEMOD_LSB, DEMOD_CWU, DEMOD_CWL = range(7)


class CDemodulator(_Handle):
    _free = "orc_demod_free"

    def __init__(self, fastfir_n=2048):
        self.h = lib().orc_demod_new(fastfir_n)

    def SetInputSampleRate(self, r):
        lib().orc_demod_set_input_rate(self.h, r)

    def SetDemod(self, mode, info):
        lib().orc_demod_set_demod(self.h, mode, C.byref(info))

    def SetDemodFreq(self, f):
        lib().orc_demod_set_freq(self.h, f)

    def GetOutputRate(self):
        return lib().orc_demod_output_rate(self.h)

    def GetSMeterPeak(self):
        return lib().orc_demod_smeter_peak(self.h)

    def GetSMeterAve(self):
        return lib().orc_demod_smeter_ave(self.h)

    def buf_limit(self):
        return lib().orc_demod_buf_limit(self.h)

    def ProcessData(self, x, stereo=False, out_cap=None):
        """Reference call semantics (every inner pass writes at out[0]; returns the sum)."""
        a = _c128(x)
        cap = out_cap or (len(a) + 65536)
        if stereo:
            out = np.zeros(cap, dtype=np.complex128)
            k = lib().orc_demod_process_stereo(self.h, len(a), _ptr(a), _ptr(out))
        else:
            out = np.zeros(cap)
            k = lib().orc_demod_process_mono(self.h, len(a), _ptr(a), _ptr(out))
        return k, out

    def process_append(self, x):
        a = _c128(x)
        out = np.zeros(len(a) + 65536)
        k = lib().orc_demod_process_mono_append(self.h, len(a), _ptr(a), _ptr(out))
        return out[:k]

    def perturb_filter_output(self, mode, eps=0.0, seed=1):
        """test-of-the-tests hook (cutesdr_oracle.c): 0 off, 1 fp32 rounding, 2 fp32 +-1 ulp, 3 additive eps * max|z|"""
        lib().orc_demod_perturb_filter_output(self.h, int(mode), float(eps), int(seed))

    def enable_taps(self, on=True):
        lib().orc_demod_enable_taps(self.h, int(on))

    def clear_taps(self):
        lib().orc_demod_clear_taps(self.h)

    def tap(self, k):
        n = lib().orc_demod_tap_len(self.h, k)
        if n == 0:
            return np.zeros(0, dtype=np.complex128 if k < 4 else np.float64)
        p = lib().orc_demod_tap_data(self.h, k)
        v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)).copy()
        return v.view(np.complex128) if k < 4 else v


def fm_defaults():
    """GUI FM defaults (gui/mainwindow.cpp:442-452, 1019-1023)."""
    return DemodInfo(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000,
                     LowCutmax=-5000, FilterClickResolution=100, Offset=0, SquelchValue=0,
                     AgcSlope=0, AgcThresh=-100, AgcManualGain=30, AgcDecay=200,
                     AgcOn=1, AgcHangOn=0, Symetric=1)


class CSoundOut(_Handle):
    """interface/soundout.cpp:155-468: queue + rate-error loop around CFractResampler (blocking mode: the put that
    would have to wait returns a negative count)"""
    _free = "orc_soundsink_free"

    def __init__(self, stereo=False):
        self.stereo = bool(stereo)
        self.h = lib().orc_soundsink_new(int(stereo))

    def ChangeUserDataRate(self, rate):
        lib().orc_soundsink_change_rate(self.h, rate)

    def SetVolume(self, vol):
        lib().orc_soundsink_set_volume(self.h, vol)

    def SetBlocking(self, on):
        lib().orc_soundsink_set_blocking(self.h, int(on))

    def PutOutQueue(self, x):
        a = _c128(x) if self.stereo else np.ascontiguousarray(x, dtype=np.float64)
        return lib().orc_soundsink_put(self.h, len(a), _ptr(a))

    def GetOutQueue(self, n):
        out = np.zeros((n, 2) if self.stereo else n, dtype=np.int16)
        lib().orc_soundsink_get(self.h, n, _ptr(out))
        return out

    def rate_correction(self):
        return lib().orc_soundsink_rate_correction(self.h)

    def ave_level(self):
        return lib().orc_soundsink_ave_level(self.h)

    def level(self):
        return lib().orc_soundsink_level(self.h)

    def ppm_error(self):
        return lib().orc_soundsink_ppm(self.h)
