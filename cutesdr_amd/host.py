"""Host-side mirror of the reference dsp/ class surface over the C ABI.

Class and method names follow the reference headers (dsp/fastfir.h:17-44 ...), argument
meaning and return values too, so the parity tests read like calls into the reference.
Numpy arrays stand in for the caller-owned TYPECPX buffers.
"""
import ctypes as C
import numpy as np
from . import _capi
from ._capi import check, check_ptr, lib


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A raw device allocation made through the C ABI (csdr_dev_alloc)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        self.ptr = check_ptr(lib().csdr_dev_alloc(device, self.nbytes), "csdr_dev_alloc")

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        check(lib().csdr_dev_upload(self.device, C.c_void_p(self.ptr + offset), _vp(a), a.nbytes), "upload")

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(lib().csdr_dev_download(self.device, _vp(out), C.c_void_p(self.ptr + offset), out.nbytes), "download")
        return out

    def free(self):
        if self.ptr:
            lib().csdr_dev_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(device=0):
    check(lib().csdr_dev_sync(device), "csdr_dev_sync")


class CFastFIR:
    """dsp/fastfir.h:17-44 -- single-channel overlap-save band-pass, host double buffers."""

    def __init__(self, fft_size=2048, device=0):
        self.n = fft_size
        self.h = check_ptr(lib().csdr_fastfir_create(device, fft_size), "csdr_fastfir_create")

    def SetupParameters(self, FLoCut, FHiCut, Offset, SampleRate):
        rc = lib().csdr_fastfir_setup(self.h, FLoCut, FHiCut, Offset, SampleRate)
        if rc == _capi.CSDR_EINVAL:
            return rc            # reference: qDebug + keep old taps
        return check(rc, "csdr_fastfir_setup")

    def ProcessData(self, InBuf):
        a = _c128(InBuf)
        out = np.empty(len(a) + self.n // 2, dtype=np.complex128)
        k = check(lib().csdr_fastfir_process(self.h, len(a), _vp(a), _vp(out)), "csdr_fastfir_process")
        return out[:k]

    def close(self):
        if self.h:
            lib().csdr_fastfir_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastFirBatch:
    """Batched, device-resident CFastFIR over [channels][T] interleaved fp32 I/Q."""

    def __init__(self, channels, fft_size=16384, device=0):
        self.channels, self.n, self.device = channels, fft_size, device
        self.h = check_ptr(lib().csdr_fastfir_batch_create(device, channels, fft_size),
                           "csdr_fastfir_batch_create")

    def setup(self, flo, fhi, offset, fs, channel=-1):
        return check(lib().csdr_fastfir_batch_setup(self.h, channel, flo, fhi, offset, fs), "batch_setup")

    def reset(self):
        check(lib().csdr_fastfir_batch_reset(self.h), "batch_reset")

    def process_ptr(self, d_in, in_stride, n_per_channel, d_out, out_stride, stream=None, blocks_per_wg=0):
        check(lib().csdr_fastfir_batch_process(self.h, C.c_void_p(d_in), in_stride, n_per_channel,
                                               C.c_void_p(d_out), out_stride,
                                               C.c_void_p(stream) if stream else None, blocks_per_wg),
              "csdr_fastfir_batch_process")

    def process(self, x, blocks_per_wg=0):
        """x: complex array [channels, T]; returns complex64 [channels, T] (host round trip)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        assert x.shape[0] == self.channels
        T = x.shape[1]
        din = DeviceBuffer(x.nbytes, self.device)
        dout = DeviceBuffer(x.nbytes, self.device)
        din.upload(x)
        self.process_ptr(din.ptr, T, T, dout.ptr, T, None, blocks_per_wg)
        sync(self.device)
        y = dout.download(np.complex64, x.size).reshape(x.shape)
        din.free(); dout.free()
        return y

    def response(self, channel=0):
        h = np.empty(self.n, dtype=np.complex128)
        check(lib().csdr_fastfir_batch_get_response(self.h, channel, _vp(h)), "get_response")
        return h

    def close(self):
        if self.h:
            lib().csdr_fastfir_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CDownConvert:
    """dsp/downconvert.h:24-120 -- NCO + decimate-by-2^n chain, host double buffers."""

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_downconvert_create(device), "csdr_downconvert_create")

    def SetCwOffset(self, offset):
        check(lib().csdr_downconvert_set_cw_offset(self.h, offset))

    def SetFrequency(self, NcoFreq):
        check(lib().csdr_downconvert_set_frequency(self.h, NcoFreq))

    def SetDataRate(self, InRate, MaxBW):
        r = lib().csdr_downconvert_set_data_rate(self.h, InRate, MaxBW)
        if r < 0:
            raise _capi.CsdrError(_capi.last_error())
        return r

    def ProcessData(self, InData):
        a = _c128(InData)
        out = np.empty(len(a), dtype=np.complex128)
        k = check(lib().csdr_downconvert_process(self.h, len(a), _vp(a), _vp(out)), "csdr_downconvert_process")
        return out[:k]

    def stages(self):
        codes = np.zeros(16, dtype=np.int32)
        n = check(lib().csdr_downconvert_get_stages(self.h, _vp(codes), 16))
        return list(int(c) for c in codes[:n])

    def nco_freq(self):
        return lib().csdr_downconvert_get_nco_freq(self.h)

    def close(self):
        if self.h:
            lib().csdr_downconvert_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DownConvertBatch:
    """Batched, device-resident CDownConvert over [channels][T] interleaved fp32 I/Q."""

    def __init__(self, channels, device=0):
        self.channels, self.device = channels, device
        self.h = check_ptr(lib().csdr_downconvert_batch_create(device, channels), "csdr_downconvert_batch_create")

    def set_cw_offset(self, offset, channel=-1):
        check(lib().csdr_downconvert_batch_set_cw_offset(self.h, channel, offset))

    def set_frequency(self, freq, channel=-1):
        check(lib().csdr_downconvert_batch_set_frequency(self.h, channel, freq))

    def set_data_rate(self, in_rate, max_bw, channel=-1):
        r = lib().csdr_downconvert_batch_set_data_rate(self.h, channel, in_rate, max_bw)
        if r < 0:
            raise _capi.CsdrError(_capi.last_error())
        return r

    def stages(self, channel=0):
        codes = np.zeros(16, dtype=np.int32)
        n = check(lib().csdr_downconvert_batch_get_stages(self.h, channel, _vp(codes), 16))
        return list(int(c) for c in codes[:n])

    def out_count(self, channel, n_in):
        return check(lib().csdr_downconvert_batch_out_count(self.h, channel, n_in))

    def process_ptr(self, d_in, in_stride, n_per_channel, d_out, out_stride, stream=None):
        check(lib().csdr_downconvert_batch_process(self.h, C.c_void_p(d_in), in_stride, n_per_channel,
                                                   C.c_void_p(d_out), out_stride,
                                                   C.c_void_p(stream) if stream else None),
              "csdr_downconvert_batch_process")

    def process(self, x):
        """x: complex [channels, T] -> list of complex64 arrays (one per channel)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        T = x.shape[1]
        din = DeviceBuffer(x.nbytes, self.device)
        dout = DeviceBuffer(x.nbytes, self.device)
        din.upload(x)
        self.process_ptr(din.ptr, T, T, dout.ptr, T)
        sync(self.device)
        y = dout.download(np.complex64, x.size).reshape(x.shape)
        din.free(); dout.free()
        return [y[c, :self.out_count(c, T)].copy() for c in range(self.channels)]

    def close(self):
        if self.h:
            lib().csdr_downconvert_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class _Obj:
    _destroy = None

    def close(self):
        if getattr(self, "h", None):
            getattr(lib(), self._destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CAgc(_Obj):
    """dsp/agc.h:19-62"""
    _destroy = "csdr_agc_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_agc_create(device), "csdr_agc_create")

    def SetParameters(self, AgcOn, UseHang, Threshold, ManualGain, Slope, Decay, SampleRate):
        check(lib().csdr_agc_set_parameters(self.h, int(AgcOn), int(UseHang), Threshold, ManualGain, Slope, Decay, SampleRate))

    def ProcessData(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.empty_like(a)
            check(lib().csdr_agc_process_cpx(self.h, len(a), _vp(a), _vp(out)), "agc_process_cpx")
        else:
            a = _f64(x); out = np.empty_like(a)
            check(lib().csdr_agc_process_real(self.h, len(a), _vp(a), _vp(out)), "agc_process_real")
        return out


class CNoiseProc(_Obj):
    """dsp/noiseproc.h:23-58 -- impulse blanker in front of the down-converter"""
    _destroy = "csdr_noiseproc_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_noiseproc_create(device), "csdr_noiseproc_create")

    def SetupBlanker(self, On, Threshold, Width, SampleRate):
        check(lib().csdr_noiseproc_setup(self.h, int(On), Threshold, Width, SampleRate), "csdr_noiseproc_setup")

    def ProcessBlanker(self, x):
        a = _c128(x); out = np.empty_like(a)
        check(lib().csdr_noiseproc_process(self.h, len(a), _vp(a), _vp(out)), "csdr_noiseproc_process")
        return out


class NoiseProcBatch(_Obj):
    """device-resident blanker over [channels][n] complex fp32"""
    _destroy = "csdr_noiseproc_batch_destroy"

    def __init__(self, channels, device=0):
        self.device, self.channels = device, channels
        self.h = check_ptr(lib().csdr_noiseproc_batch_create(device, channels), "csdr_noiseproc_batch_create")

    def setup(self, On, Threshold, Width, SampleRate, channel=-1):
        check(lib().csdr_noiseproc_batch_setup(self.h, channel, int(On), Threshold, Width, SampleRate))

    def process_ptr(self, d_in, in_stride, n, d_out, out_stride, stream=0):
        check(lib().csdr_noiseproc_batch_process(self.h, C.c_void_p(d_in), in_stride, n, C.c_void_p(d_out), out_stride,
                                                 C.c_void_p(stream)), "csdr_noiseproc_batch_process")

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        assert x.shape[0] == self.channels
        n = x.shape[1]
        din, dout = DeviceBuffer(x.nbytes, self.device), DeviceBuffer(x.nbytes, self.device)
        din.upload(x)
        self.process_ptr(din.ptr, n, n, dout.ptr, n)
        sync(self.device)
        return dout.download(np.complex64, x.size).reshape(x.shape)


def unpack_packets(raw, pkt_len, device=0):
    """UDP datagrams (uint8 [npackets, pkt_len], 1028 = 16 bit / 1444 = 24 bit) -> complex128 samples
    (interface/netiobase.cpp:479-527)"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, pkt_len)
    per = 240 if pkt_len == 1444 else 256
    out = np.empty(raw.shape[0] * per, dtype=np.complex128)
    k = check(lib().csdr_ingest_unpack_host(device, _vp(raw), raw.shape[0], pkt_len, _vp(out)), "csdr_ingest_unpack_host")
    return out[:k]


def unpack_packets_batch(raw, pkt_len, dc=None, device=0):
    """raw uint8 [channels, npackets, pkt_len] -> complex64 [channels, samples]; dc: optional [channels, 2] offsets"""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    ch, npk = raw.shape[0], raw.shape[1]
    per = 240 if pkt_len == 1444 else 256
    dp, do = DeviceBuffer(raw.nbytes, device), DeviceBuffer(ch * npk * per * 8, device)
    dp.upload(raw)
    ddc = None
    if dc is not None:
        ddc = DeviceBuffer(ch * 16, device); ddc.upload(np.ascontiguousarray(dc, dtype=np.float64))
    check(lib().csdr_ingest_unpack(device, C.c_void_p(dp.ptr), ch, npk, pkt_len, C.c_void_p(do.ptr), npk * per,
                                   C.c_void_p(ddc.ptr if ddc else 0), C.c_void_p(0)), "csdr_ingest_unpack")
    sync(device)
    return do.download(np.complex64, ch * npk * per).reshape(ch, npk * per)


def spurcal(dc, x, device=0):
    """NcoSpurCalibrate running I/Q means (interface/sdrinterface.cpp:829-848); returns the new [I, Q]"""
    a = _c128(x); d = np.ascontiguousarray(dc, dtype=np.float64).copy()
    check(lib().csdr_ingest_spurcal_host(device, len(a), _vp(a), _vp(d)), "csdr_ingest_spurcal_host")
    return d


class CSMeter(_Obj):
    """dsp/smeter.h:13-28"""
    _destroy = "csdr_smeter_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_smeter_create(device), "csdr_smeter_create")

    def ProcessData(self, x, SampleRate):
        a = _c128(x)
        check(lib().csdr_smeter_process(self.h, len(a), _vp(a), SampleRate), "smeter_process")

    def GetPeak(self):
        return lib().csdr_smeter_get_peak(self.h)

    def GetAve(self):
        return lib().csdr_smeter_get_ave(self.h)


class CFir(_Obj):
    """dsp/fir.h:20-43"""
    _destroy = "csdr_fir_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_fir_create(device), "csdr_fir_create")

    def InitConstFir(self, coef):
        c = _f64(coef)
        check(lib().csdr_fir_init_const(self.h, len(c), _vp(c)))

    def InitLPFilter(self, Scale, Astop, Fpass, Fstop, Fsamprate):
        return check(lib().csdr_fir_init_lp(self.h, Scale, Astop, Fpass, Fstop, Fsamprate))

    def InitHPFilter(self, Scale, Astop, Fpass, Fstop, Fsamprate):
        return check(lib().csdr_fir_init_hp(self.h, Scale, Astop, Fpass, Fstop, Fsamprate))

    def GenerateHBFilter(self, FreqOffset):
        check(lib().csdr_fir_generate_hb(self.h, FreqOffset))

    def taps(self):
        c = np.zeros(75); i = np.zeros(75); q = np.zeros(75)
        n = check(lib().csdr_fir_get_taps(self.h, _vp(c), _vp(i), _vp(q)))
        return c[:n], i[:n], q[:n]

    def ProcessFilter(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.empty_like(a)
            check(lib().csdr_fir_process_cpx(self.h, len(a), _vp(a), _vp(out)))
        else:
            a = _f64(x); out = np.empty_like(a)
            check(lib().csdr_fir_process_real(self.h, len(a), _vp(a), _vp(out)))
        return out


class CIir(_Obj):
    """dsp/iir.h:17-39"""
    _destroy = "csdr_iir_destroy"
    KIND = {"LP": 0, "HP": 1, "BP": 2, "BR": 3}

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_iir_create(device), "csdr_iir_create")

    def Init(self, kind, F0Freq, FilterQ, SampleRate):
        check(lib().csdr_iir_init(self.h, self.KIND[kind], F0Freq, FilterQ, SampleRate))

    def coefs(self):
        c = np.zeros(5); check(lib().csdr_iir_get_coefs(self.h, _vp(c))); return c

    def ProcessFilter(self, x):
        if np.iscomplexobj(x):
            a = _c128(x); out = np.empty_like(a)
            check(lib().csdr_iir_process_cpx(self.h, len(a), _vp(a), _vp(out)))
        else:
            a = _f64(x); out = np.empty_like(a)
            check(lib().csdr_iir_process_real(self.h, len(a), _vp(a), _vp(out)))
        return out


class _Demod(_Obj):
    _prefix = None

    def _run(self, x, stereo, *extra):
        a = _c128(x)
        if stereo:
            out = np.empty_like(a)
            k = check(getattr(lib(), self._prefix + "_process_stereo")(self.h, len(a), *extra, _vp(a), _vp(out)))
        else:
            out = np.empty(len(a))
            k = check(getattr(lib(), self._prefix + "_process_mono")(self.h, len(a), *extra, _vp(a), _vp(out)))
        return out[:k]


class CAmDemod(_Demod):
    """dsp/amdemod.h:14-25"""
    _destroy, _prefix = "csdr_amdemod_destroy", "csdr_amdemod"

    def __init__(self, samplerate, device=0):
        self.h = check_ptr(lib().csdr_amdemod_create(device, samplerate), "csdr_amdemod_create")

    def SetBandwidth(self, Bandwidth):
        check(lib().csdr_amdemod_set_bandwidth(self.h, Bandwidth))

    def ProcessData(self, x, stereo=False):
        return self._run(x, stereo)


class CSamDemod(_Demod):
    """dsp/samdemod.h:14-32"""
    _destroy, _prefix = "csdr_samdemod_destroy", "csdr_samdemod"

    def __init__(self, samplerate, device=0):
        self.h = check_ptr(lib().csdr_samdemod_create(device, samplerate), "csdr_samdemod_create")

    def ProcessData(self, x, stereo=False):
        return self._run(x, stereo)


class CFmDemod(_Demod):
    """dsp/fmdemod.h:17-54"""
    _destroy, _prefix = "csdr_fmdemod_destroy", "csdr_fmdemod"

    def __init__(self, samplerate, device=0):
        self.h = check_ptr(lib().csdr_fmdemod_create(device, samplerate), "csdr_fmdemod_create")

    def SetSquelch(self, Value):
        check(lib().csdr_fmdemod_set_squelch(self.h, Value))

    def squelched(self):
        return bool(check(lib().csdr_fmdemod_get_squelched(self.h)))

    def ProcessData(self, x, FmBW, stereo=False):
        return self._run(x, stereo, float(FmBW))


def ssb_demod(x, stereo=False):
    """dsp/ssbdemod.cpp:48-60"""
    a = _c128(x)
    if stereo:
        out = np.empty_like(a); check(lib().csdr_ssbdemod_process_stereo(len(a), _vp(a), _vp(out)))
    else:
        out = np.empty(len(a)); check(lib().csdr_ssbdemod_process_mono(len(a), _vp(a), _vp(out)))
    return out


class DemodInfo(C.Structure):
    """tDemodInfo (dsp/demodulator.h:35-54) without the QString label"""
    _fields_ = [(n, C.c_int) for n in (
        "HiCut", "HiCutmin", "HiCutmax", "LowCut", "LowCutmin", "LowCutmax",
        "FilterClickResolution", "Offset", "SquelchValue",
        "AgcSlope", "AgcThresh", "AgcManualGain", "AgcDecay",
        "AgcOn", "AgcHangOn", "Symetric")]


DEMOD_AM, DEMOD_SAM, DEMOD_FM, DEMOD_USB, DEMOD_LSB, DEMOD_CWU, DEMOD_CWL = range(7)


def fm_defaults():
    """The GUI's FM demodulator settings (gui/mainwindow.cpp:442-452, 1019-1023)."""
    return DemodInfo(HiCut=5000, HiCutmin=5000, HiCutmax=15000, LowCut=-5000, LowCutmin=-15000,
                     LowCutmax=-5000, FilterClickResolution=100, Offset=0, SquelchValue=0,
                     AgcSlope=0, AgcThresh=-100, AgcManualGain=30, AgcDecay=200,
                     AgcOn=1, AgcHangOn=0, Symetric=1)


class CDemodulator(_Obj):
    """dsp/demodulator.h:56-100 -- the whole chain, host double buffers, reference call semantics."""
    _destroy = "csdr_demod_destroy"

    def __init__(self, fastfir_n=2048, device=0):
        self.n = fastfir_n
        self.h = check_ptr(lib().csdr_demod_create(device, fastfir_n), "csdr_demod_create")

    def SetInputSampleRate(self, InputRate):
        check(lib().csdr_demod_set_input_rate(self.h, InputRate))

    def SetDemod(self, Mode, CurrentDemodInfo):
        check(lib().csdr_demod_set_demod(self.h, Mode, C.byref(CurrentDemodInfo)), "csdr_demod_set_demod")

    def SetDemodFreq(self, Freq):
        check(lib().csdr_demod_set_freq(self.h, Freq))

    def GetOutputRate(self):
        return lib().csdr_demod_get_output_rate(self.h)

    def GetSMeterPeak(self):
        return lib().csdr_demod_get_smeter_peak(self.h)

    def GetSMeterAve(self):
        return lib().csdr_demod_get_smeter_ave(self.h)

    def buf_limit(self):
        return check(lib().csdr_demod_get_buf_limit(self.h))

    def ProcessData(self, x, stereo=False):
        a = _c128(x)
        cap = len(a) + self.n + 65536
        if stereo:
            out = np.zeros(cap, dtype=np.complex128)
            k = check(lib().csdr_demod_process_stereo(self.h, len(a), _vp(a), _vp(out)), "csdr_demod_process_stereo")
        else:
            out = np.zeros(cap)
            k = check(lib().csdr_demod_process_mono(self.h, len(a), _vp(a), _vp(out)), "csdr_demod_process_mono")
        return k, out

    def process_append(self, x):
        a = _c128(x)
        out = np.zeros(len(a) + self.n + 65536)
        k = check(lib().csdr_demod_process_mono_append(self.h, len(a), _vp(a), _vp(out)), "process_append")
        return out[:k]

    def set_deferred(self, on=True):
        """deferred output (include/cutesdr_mi.h): a pass returns the previous pass's samples, the call never waits for
        the device; flush() hands over the last pass"""
        check(lib().csdr_demod_set_deferred(self.h, 1 if on else 0), "csdr_demod_set_deferred")

    def flush(self, stereo=False):
        cap = 2 * (self.n + 65536)
        out = np.zeros(cap)
        k = check(lib().csdr_demod_flush(self.h, _vp(out), cap), "csdr_demod_flush")
        return out[:2 * k].view(np.complex128) if stereo else out[:k]

    def enable_taps(self, mask=15, callback=None):
        """the chain's test points PROFILE_1..4 (dsp/demodulator.cpp:175,180,187,208): bit k-1 of mask switches tap k on;
        callback(profile, n, data, is_complex, rate) is called per pass, else the samples accumulate for tap()"""
        self._tap_cb = None
        fn = None
        if callback is not None:
            proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double)
            def thunk(user, profile, n, data, cpx, rate):
                v = np.ctypeslib.as_array(data, shape=(n * (2 if cpx else 1),)).copy() if n else np.zeros(0)
                callback(profile, n, v.view(np.complex128) if cpx else v, bool(cpx), rate)
            self._tap_cb = proto(thunk)                       # keep the thunk alive as long as the taps are on
            fn = C.cast(self._tap_cb, C.c_void_p)
        check(lib().csdr_demod_set_taps(self.h, int(mask), fn, None), "csdr_demod_set_taps")

    def tap(self, k, cap=1 << 22):
        """what tap k (1..4) has accumulated since it was last read (complex for 1..3; tap 4 as the doubles it holds)"""
        out = np.zeros(cap)
        n = check(lib().csdr_demod_get_tap(self.h, k, _vp(out), cap), "csdr_demod_get_tap")
        return out[:n].view(np.complex128).copy() if k < 4 else out[:n].copy()


class DemodBatch(_Obj):
    """Batched device-resident receive chains: [channels][T] fp32 I/Q in, mono fp32 audio out."""
    _destroy = "csdr_demod_batch_destroy"

    def __init__(self, channels, fastfir_n=2048, device=0):
        self.channels, self.n, self.device = channels, fastfir_n, device
        self.h = check_ptr(lib().csdr_demod_batch_create(device, channels, fastfir_n), "csdr_demod_batch_create")

    def set_input_rate(self, rate):
        check(lib().csdr_demod_batch_set_input_rate(self.h, rate))

    def set_taps(self, mask):
        """stage taps PROFILE_1..3 of every receiver (strict mode): the last call's samples stay on the device for tap()"""
        check(lib().csdr_demod_batch_set_taps(self.h, int(mask)), "csdr_demod_batch_set_taps")

    def tap(self, channel, k, cap=1 << 20):
        out = np.zeros(cap, dtype=np.float32)
        n = check(lib().csdr_demod_batch_get_tap(self.h, channel, k, _vp(out), cap), "csdr_demod_batch_get_tap")
        return out[:n].astype(np.float64).view(np.complex128).copy()

    def set_demod(self, channel, mode, info):
        check(lib().csdr_demod_batch_set_demod(self.h, channel, mode, C.byref(info)), "batch_set_demod")

    def commit(self):
        check(lib().csdr_demod_batch_commit(self.h), "batch_commit")

    def set_freq(self, channel, freq):
        check(lib().csdr_demod_batch_set_freq(self.h, channel, freq))

    def output_rate(self, channel):
        return lib().csdr_demod_batch_get_output_rate(self.h, channel)

    def smeter_ave(self, channel):
        return lib().csdr_demod_batch_get_smeter_ave(self.h, channel)

    def out_count(self, channel):
        return check(lib().csdr_demod_batch_out_count(self.h, channel))

    def group_count(self):
        """(plan groups, rows of all groups incl. muted ones)"""
        rows = C.c_int(0)
        g = check(lib().csdr_demod_batch_group_count(self.h, C.byref(rows)))
        return g, rows.value

    def set_pipelined(self, on=True):
        check(lib().csdr_demod_batch_set_pipelined(self.h, int(on)), "set_pipelined")

    def set_input_rows(self, rows=None):
        """receiver c reads input row rows[c] (several receivers cut from one stream); None: row c again"""
        if rows is None:
            check(lib().csdr_demod_batch_set_input_rows(self.h, None), "set_input_rows")
            return
        a = np.ascontiguousarray(rows, dtype=np.int32)
        assert a.shape == (self.channels,)
        check(lib().csdr_demod_batch_set_input_rows(self.h, _vp(a)), "set_input_rows")

    def flush(self, stream=None):
        check(lib().csdr_demod_batch_flush(self.h, C.c_void_p(stream) if stream else None), "flush")

    def smeter_all_ptr(self, d_ave, d_peak=None, stream=None):
        check(lib().csdr_demod_batch_get_smeter_all(self.h, C.c_void_p(d_ave) if d_ave else None,
                                                    C.c_void_p(d_peak) if d_peak else None,
                                                    C.c_void_p(stream) if stream else None), "get_smeter_all")

    def smeter_all(self, peak=False):
        """S-meter averages (and optionally peaks, which resets them) of every channel"""
        d = DeviceBuffer(8 * self.channels, self.device)
        self.smeter_all_ptr(d.ptr, d.ptr + 4 * self.channels if peak else None)
        sync(self.device)
        v = d.download(np.float32, 2 * self.channels)
        d.free()
        return (v[:self.channels], v[self.channels:]) if peak else v[:self.channels]

    def process_ptr(self, d_in, in_stride, n, d_out, out_stride, stream=None):
        check(lib().csdr_demod_batch_process(self.h, C.c_void_p(d_in), in_stride, n, C.c_void_p(d_out), out_stride,
                                             C.c_void_p(stream) if stream else None), "csdr_demod_batch_process")

    def process_packets(self, raw, pkt_len, blanker=None):
        """raw uint8 [channels, npackets, pkt_len] -> list of audio arrays (unpack, optional blanker, chain)"""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.shape[0] == self.channels and raw.shape[2] == pkt_len
        npk = raw.shape[1]
        T = npk * (240 if pkt_len == 1444 else 256)
        cap = T // 8 + self.n + 4096
        dp, dout = DeviceBuffer(raw.nbytes, self.device), DeviceBuffer(self.channels * cap * 4, self.device)
        dp.upload(raw)
        check(lib().csdr_demod_batch_process_packets(self.h, C.c_void_p(dp.ptr), npk, pkt_len,
                                                     blanker.h if blanker is not None else None,
                                                     C.c_void_p(dout.ptr), cap, None), "csdr_demod_batch_process_packets")
        sync(self.device)
        flat = dout.download(np.float32, self.channels * cap).reshape(self.channels, cap)
        return [flat[c, :check(lib().csdr_demod_batch_out_count(self.h, c))].copy() for c in range(self.channels)]

    def process_blanked(self, x, blanker):
        """x complex [channels, T] -> list of audio rows, CNoiseProc's blanker (a NoiseProcBatch) fused in front"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        assert x.shape[0] == self.channels
        T = x.shape[1]
        cap = T // 8 + self.n + 4096
        din, dout = DeviceBuffer(x.nbytes, self.device), DeviceBuffer(self.channels * cap * 4, self.device)
        din.upload(x)
        check(lib().csdr_demod_batch_process_blanked(self.h, C.c_void_p(din.ptr), T, T, blanker.h, C.c_void_p(dout.ptr), cap,
                                                     None), "csdr_demod_batch_process_blanked")
        sync(self.device)
        flat = dout.download(np.float32, self.channels * cap).reshape(self.channels, cap)
        return [flat[c, :check(lib().csdr_demod_batch_out_count(self.h, c))].copy() for c in range(self.channels)]

    def process(self, x, stereo=False):
        """x complex [channels, T] -> list of audio rows (float32 mono, or complex64 with stereo=True)"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        T = x.shape[1]
        cap = T + self.n
        w = 8 if stereo else 4
        din = DeviceBuffer(x.nbytes, self.device)
        dout = DeviceBuffer(w * self.channels * cap, self.device)
        din.upload(x)
        fn = lib().csdr_demod_batch_process_stereo if stereo else lib().csdr_demod_batch_process
        check(fn(self.h, C.c_void_p(din.ptr), T, T, C.c_void_p(dout.ptr), cap, None), "csdr_demod_batch_process")
        sync(self.device)
        y = dout.download(np.complex64 if stereo else np.float32, self.channels * cap).reshape(self.channels, cap)
        din.free(); dout.free()
        return [y[c, :self.out_count(c)].copy() for c in range(self.channels)]


class ShardedDemodBatch(_Obj):
    """csdr_demod_shard: one host object, N shards of the batched chain on N devices (several shards may share one);
    global channel ids, contiguous ranges, no collective on the data path."""
    _destroy = "csdr_demod_shard_destroy"

    def __init__(self, devices, channels, fastfir_n=2048):
        self.devices, self.channels, self.n = list(devices), channels, fastfir_n
        arr = (C.c_int * len(self.devices))(*self.devices)
        self.h = check_ptr(lib().csdr_demod_shard_create(arr, len(self.devices), channels, fastfir_n), "csdr_demod_shard_create")
        self.ranges = []
        for k in range(len(self.devices)):
            f, c, d = C.c_int(), C.c_int(), C.c_int()
            check(lib().csdr_demod_shard_range(self.h, k, C.byref(f), C.byref(c), C.byref(d)))
            self.ranges.append((f.value, c.value, d.value))

    def set_input_rate(self, rate):
        check(lib().csdr_demod_shard_set_input_rate(self.h, rate))

    def set_demod(self, channel, mode, info):
        check(lib().csdr_demod_shard_set_demod(self.h, channel, mode, C.byref(info)), "shard_set_demod")

    def commit(self):
        check(lib().csdr_demod_shard_commit(self.h), "shard_commit")

    def set_freq(self, channel, freq):
        check(lib().csdr_demod_shard_set_freq(self.h, channel, freq))

    def output_rate(self, channel):
        return lib().csdr_demod_shard_get_output_rate(self.h, channel)

    def set_pipelined(self, on=True):
        check(lib().csdr_demod_shard_set_pipelined(self.h, int(on)), "shard_set_pipelined")

    def out_count(self, channel):
        return check(lib().csdr_demod_shard_out_count(self.h, channel))

    def sync(self):
        check(lib().csdr_demod_shard_sync(self.h), "shard_sync")

    def smeter_all(self, peak=False):
        ave = np.zeros(self.channels, dtype=np.float32)
        pk = np.zeros(self.channels, dtype=np.float32)
        check(lib().csdr_demod_shard_get_smeter_all(self.h, _vp(ave), _vp(pk) if peak else None), "shard_get_smeter_all")
        return (ave, pk) if peak else ave

    def set_input_rows(self, rows, nrows):
        if rows is None:
            check(lib().csdr_demod_shard_set_input_rows(self.h, None, 0)); return
        a = np.ascontiguousarray(rows, dtype=np.int32)
        assert a.shape == (self.channels,)
        check(lib().csdr_demod_shard_set_input_rows(self.h, _vp(a), int(nrows)), "shard_set_input_rows")

    def _outs(self, cap):
        return [DeviceBuffer(4 * c * cap, d) for (_, c, d) in self.ranges]

    def _collect(self, outs, cap):
        res = []
        for (f, c, d), o in zip(self.ranges, outs):
            y = o.download(np.float32, c * cap).reshape(c, cap)
            res += [y[i, :self.out_count(f + i)].copy() for i in range(c)]
            o.free()
        return res

    def process(self, x):
        """x complex [channels, T] (host): every shard's rows go to its device; -> list of audio rows by global channel"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        T = x.shape[1]
        cap = T + self.n
        ins = []
        for (f, c, d) in self.ranges:
            b = DeviceBuffer(8 * c * T, d); b.upload(np.ascontiguousarray(x[f:f + c])); ins.append(b)
        outs = self._outs(cap)
        pin = (C.c_void_p * len(ins))(*[b.ptr for b in ins]); pout = (C.c_void_p * len(outs))(*[b.ptr for b in outs])
        check(lib().csdr_demod_shard_process(self.h, pin, T, T, pout, cap, None), "csdr_demod_shard_process")
        self.sync()
        for b in ins:
            b.free()
        return self._collect(outs, cap)

    def set_blanker(self, On, Threshold, Width, SampleRate):
        check(lib().csdr_demod_shard_set_blanker(self.h, int(On), Threshold, Width, SampleRate), "shard_set_blanker")

    def process_blanked(self, x):
        """process() with the blanker of set_blanker fused in front"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        T = x.shape[1]
        cap = T + self.n
        ins = []
        for (f, c, d) in self.ranges:
            b = DeviceBuffer(8 * c * T, d); b.upload(np.ascontiguousarray(x[f:f + c])); ins.append(b)
        outs = self._outs(cap)
        pin = (C.c_void_p * len(ins))(*[b.ptr for b in ins]); pout = (C.c_void_p * len(outs))(*[b.ptr for b in outs])
        check(lib().csdr_demod_shard_process_blanked(self.h, pin, T, T, pout, cap, None), "csdr_demod_shard_process_blanked")
        self.sync()
        for b in ins:
            b.free()
        return self._collect(outs, cap)

    def process_packets(self, raw, pkt_len):
        """raw uint8 [channels, npackets, pkt_len] (host): every shard's receivers' datagrams go to its device"""
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.shape[0] == self.channels and raw.shape[2] == pkt_len
        npk = raw.shape[1]
        T = npk * (240 if pkt_len == 1444 else 256)
        cap = T // 8 + self.n + 4096
        ins = []
        for (f, c, d) in self.ranges:
            b = DeviceBuffer(c * npk * pkt_len, d); b.upload(np.ascontiguousarray(raw[f:f + c])); ins.append(b)
        outs = self._outs(cap)
        pin = (C.c_void_p * len(ins))(*[b.ptr for b in ins]); pout = (C.c_void_p * len(outs))(*[b.ptr for b in outs])
        check(lib().csdr_demod_shard_process_packets(self.h, pin, npk, pkt_len, pout, cap, None), "csdr_demod_shard_process_packets")
        self.sync()
        for b in ins:
            b.free()
        return self._collect(outs, cap)

    def process_shared_ptr(self, d_block, src_device, T, in_stride, outs, cap, src_stream=None):
        """the raw call: block [nrows][in_stride] resident on src_device, one output buffer per shard; asynchronous"""
        pout = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        check(lib().csdr_demod_shard_process_shared(self.h, C.c_void_p(d_block), src_device,
                                                    C.c_void_p(src_stream) if src_stream else None, in_stride, T, pout, cap),
              "csdr_demod_shard_process_shared")

    def process_shared(self, block, src_device=0):
        """block complex [nrows, T] (host) -> uploaded once to src_device, broadcast by the object"""
        block = np.ascontiguousarray(block, dtype=np.complex64)
        T = block.shape[1]
        cap = T + self.n
        b = DeviceBuffer(block.nbytes, src_device); b.upload(block)
        sync(src_device)
        outs = self._outs(cap)
        pout = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        check(lib().csdr_demod_shard_process_shared(self.h, C.c_void_p(b.ptr), src_device, None, T, T, pout, cap),
              "csdr_demod_shard_process_shared")
        self.sync()
        b.free()
        return self._collect(outs, cap)


class CFft(_Obj):
    """dsp/fft.h:24-85 -- display spectrum + plain transforms, 512..65536 points."""
    _destroy = "csdr_fft_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_fft_create(device), "csdr_fft_create")
        self.size = 2048

    def SetFFTParams(self, size, invert, dBCompensation, SampleFreq):
        check(lib().csdr_fft_set_params(self.h, size, int(invert), dBCompensation, SampleFreq), "csdr_fft_set_params")
        self.size = max(512, min(65536, size))

    def SetFFTAve(self, ave):
        check(lib().csdr_fft_set_ave(self.h, ave))

    def ResetFFT(self):
        check(lib().csdr_fft_reset(self.h))

    def PutInDisplayFFT(self, InBuf):
        a = _c128(InBuf)
        return check(lib().csdr_fft_put_display(self.h, len(a), _vp(a)), "csdr_fft_put_display")

    def GetScreenIntegerFFTData(self, MaxHeight, MaxWidth, MaxdB, MindB, StartFreq, StopFreq):
        out = np.zeros(max(MaxWidth, 1), dtype=np.int32)
        ov = check(lib().csdr_fft_get_screen(self.h, MaxHeight, MaxWidth, MaxdB, MindB, StartFreq, StopFreq, _vp(out)))
        return bool(ov), out

    def ave_buf(self):
        out = np.zeros(self.size, dtype=np.float32)
        check(lib().csdr_fft_get_ave(self.h, _vp(out)))
        return out

    def FwdFFT(self, x):
        a = _c128(x).copy(); check(lib().csdr_fft_fwd(self.h, _vp(a))); return a

    def RevFFT(self, x):
        a = _c128(x).copy(); check(lib().csdr_fft_rev(self.h, _vp(a))); return a


class FftBatch(_Obj):
    """Batched display spectra over [channels][frames*size] fp32 I/Q on the device."""
    _destroy = "csdr_fft_batch_destroy"

    def __init__(self, channels, device=0):
        self.channels, self.device = channels, device
        self.h = check_ptr(lib().csdr_fft_batch_create(device, channels), "csdr_fft_batch_create")

    def set_params(self, size, invert, db_comp, fs):
        check(lib().csdr_fft_batch_set_params(self.h, size, int(invert), db_comp, fs), "fft_batch_set_params")

    def set_ave(self, ave):
        check(lib().csdr_fft_batch_set_ave(self.h, ave))

    def size(self):
        return check(lib().csdr_fft_batch_size(self.h))

    def put_display_ptr(self, d_in, in_stride, nframes, stream=None):
        check(lib().csdr_fft_batch_put_display(self.h, C.c_void_p(d_in), in_stride, nframes,
                                               C.c_void_p(stream) if stream else None), "fft_batch_put_display")

    def put_display(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = self.size()
        din = DeviceBuffer(x.nbytes, self.device)
        din.upload(x)
        self.put_display_ptr(din.ptr, x.shape[1], x.shape[1] // n)
        sync(self.device)
        din.free()

    def ave_buf(self, channel):
        out = np.zeros(self.size(), dtype=np.float32)
        check(lib().csdr_fft_batch_get_ave(self.h, channel, _vp(out)))
        return out

    def screen_all(self, MaxHeight, MaxWidth, MaxdB, MindB, StartFreq, StopFreq):
        """GetScreenIntegerFFTData of every channel at once on the device -> (overload [C], pixels [C, MaxWidth])"""
        dout = DeviceBuffer(self.channels * max(MaxWidth, 1) * 4, self.device)
        dov = DeviceBuffer(self.channels * 4, self.device)
        dout.upload(np.full(self.channels * max(MaxWidth, 1), -1, dtype=np.int32))
        check(lib().csdr_fft_batch_get_screen_all(self.h, MaxHeight, MaxWidth, MaxdB, MindB, StartFreq, StopFreq,
                                                  C.c_void_p(dout.ptr), max(MaxWidth, 1), C.c_void_p(dov.ptr), None),
              "csdr_fft_batch_get_screen_all")
        sync(self.device)
        return (dov.download(np.int32, self.channels) != 0,
                dout.download(np.int32, self.channels * max(MaxWidth, 1)).reshape(self.channels, -1))

    def waterfall_all(self, MaxWidth, MaxdB, MindB, StartFreq, StopFreq, fill=0):
        """The new waterfall line of every channel (CPlotter::draw, gui/plotter.cpp:425-441) on the device ->
        (overload [C], 0xFFRRGGBB pixels [C, MaxWidth]); pixels no bin maps to come back as `fill`"""
        w = max(MaxWidth, 1)
        dout = DeviceBuffer(self.channels * w * 4, self.device)
        dov = DeviceBuffer(self.channels * 4, self.device)
        dout.upload(np.full(self.channels * w, fill, dtype=np.uint32))
        check(lib().csdr_fft_batch_get_waterfall_all(self.h, MaxWidth, C.c_double(MaxdB), C.c_double(MindB), StartFreq, StopFreq,
                                                     C.c_void_p(dout.ptr), C.c_longlong(w), C.c_void_p(dov.ptr), None),
              "csdr_fft_batch_get_waterfall_all")
        sync(self.device)
        ov = dov.download(np.int32, self.channels) != 0
        pix = dout.download(np.uint32, self.channels * w).reshape(self.channels, -1)
        dout.free(); dov.free()
        return ov, pix

    def total_count(self, channel):
        return check(lib().csdr_fft_batch_get_total_count(self.h, channel))


class CFractResampler(_Obj):
    """dsp/fractresampler.h:17-33"""
    _destroy = "csdr_resampler_destroy"

    def __init__(self, device=0):
        self.h = check_ptr(lib().csdr_resampler_create(device), "csdr_resampler_create")

    def Init(self, MaxInputSize):
        check(lib().csdr_resampler_init(self.h, MaxInputSize))

    def Resample(self, x, Rate, gain=None):
        cap = int(len(x) / Rate) + 8
        if np.iscomplexobj(x):
            a = _c128(x)
            if gain is None:
                out = np.zeros(cap, dtype=np.complex128)
                k = check(lib().csdr_resampler_resample_cpx(self.h, len(a), Rate, _vp(a), _vp(out)))
                return out[:k]
            out = np.zeros(2 * cap, dtype=np.int16)
            k = check(lib().csdr_resampler_resample_cpx_i16(self.h, len(a), Rate, _vp(a), _vp(out), gain))
            return out[:2 * k].reshape(-1, 2)
        a = _f64(x)
        if gain is None:
            out = np.zeros(cap)
            k = check(lib().csdr_resampler_resample_real(self.h, len(a), Rate, _vp(a), _vp(out)))
            return out[:k]
        out = np.zeros(cap, dtype=np.int16)
        k = check(lib().csdr_resampler_resample_real_i16(self.h, len(a), Rate, _vp(a), _vp(out), gain))
        return out[:k]


class ResamplerBatch(_Obj):
    """CFractResampler over the chain's mono audio rows: [channels][n] fp32, all channels on one clock"""
    _destroy = "csdr_resampler_batch_destroy"

    def __init__(self, channels, device=0):
        self.channels, self.device = channels, device
        self.h = check_ptr(lib().csdr_resampler_batch_create(device, channels), "csdr_resampler_batch_create")

    def resample_ptr(self, d_in, in_stride, n, rate, d_out, out_stride, gain=None, stream=None):
        f32 = C.c_void_p(d_out) if gain is None else None
        i16 = None if gain is None else C.c_void_p(d_out)
        return check(lib().csdr_resampler_batch_resample(self.h, C.c_void_p(d_in), in_stride, n, rate, f32, i16, out_stride,
                                                         0.0 if gain is None else gain,
                                                         C.c_void_p(stream) if stream else None), "csdr_resampler_batch_resample")

    def resample(self, x, rate, gain=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape[0] == self.channels
        n = x.shape[1]
        cap = int(n / rate) + 8
        din = DeviceBuffer(max(x.nbytes, 4), self.device)
        dout = DeviceBuffer(self.channels * cap * (4 if gain is None else 2), self.device)
        din.upload(x)
        k = self.resample_ptr(din.ptr, n, n, rate, dout.ptr, cap, gain)
        sync(self.device)
        out = dout.download(np.float32 if gain is None else np.int16, self.channels * cap).reshape(self.channels, cap)
        return out[:, :k].copy()


class CSoundOut(_Obj):
    """interface/soundout.cpp -- queue and rate-error loop of the sound sink (both modes) around the
    device resampler; PutOutQueue / GetOutQueue / ChangeUserDataRate / SetVolume as in the reference"""
    _destroy = "csdr_soundsink_destroy"

    def __init__(self, stereo=False, device=0):
        self.stereo = bool(stereo)
        self.h = check_ptr(lib().csdr_soundsink_create(device, int(stereo)), "csdr_soundsink_create")

    def ChangeUserDataRate(self, UsrDataRate):
        check(lib().csdr_soundsink_change_user_data_rate(self.h, UsrDataRate))

    def SetVolume(self, vol):
        check(lib().csdr_soundsink_set_volume(self.h, vol))

    def SetBlocking(self, on):
        """CSoundOut::Start's BlockingMode: PutOutQueue waits while the queue is full, GetOutQueue skips the rate loop"""
        check(lib().csdr_soundsink_set_blocking(self.h, int(on)))

    def PutOutQueue(self, x):
        a = _c128(x) if self.stereo else _f64(x)
        return check(lib().csdr_soundsink_put(self.h, len(a), _vp(a)), "csdr_soundsink_put")

    def GetOutQueue(self, n):
        out = np.zeros((n, 2) if self.stereo else n, dtype=np.int16)
        check(lib().csdr_soundsink_get(self.h, n, _vp(out)), "csdr_soundsink_get")
        return out

    def rate_correction(self):
        return lib().csdr_soundsink_get_rate_correction(self.h)

    def ave_level(self):
        return lib().csdr_soundsink_get_ave_level(self.h)

    def level(self):
        return check(lib().csdr_soundsink_get_level(self.h))

    def ppm_error(self):
        return lib().csdr_soundsink_get_ppm_error(self.h)
