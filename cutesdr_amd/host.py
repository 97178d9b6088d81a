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
