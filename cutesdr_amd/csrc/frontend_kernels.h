// frontend_kernels.h -- launch interface of the input-rate stages in front of the down-converter
// (SURVEY 8(f) rows f1, f2): noise blanker, wire-format unpack, DC (NCO spur) estimate.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include "wire_format.hpp"

namespace csdr {

constexpr int NB_HIST = 32768;          // raw samples kept per channel between calls (>= MAX_AVE, noiseproc.cpp:51)
constexpr int NB_MAX_WIDTH = 4096;      // noiseproc.cpp:49

struct NbChan {                         // CNoiseProc state (dsp/noiseproc.h:36-52), stream form
    int on, delay_n, mag_n, width_n;    // m_On, m_DelaySamples, m_MagSamples, m_WidthSamples
    double ratio;                       // m_Ratio
    double sum;                         // m_MagAveSum
    long long since_trig;               // samples since the last trigger (>= width_n: not blanking)
};

struct NbArgs {
    const NbChan *chan;                 // [channels] state at the start of the call
    NbChan *chan_next;                  // [channels] state after it (ping-pong, like the history)
    const float *in;  long in_stride;   // complex fp32 [channels][in_stride]; unused when wire.pk is set
    WireIn wire;                        // optional: the call's samples as datagrams (wire_format.hpp)
    float *out;       long out_stride;  // complex fp32 [channels][out_stride]; may alias `in` only if hist is kept
    unsigned *mask;   long mask_stride; // MASK MODE (out == nullptr): instead of the blanked, delayed samples the kernel
                                        // leaves one bit per sample -- bit i & 31 of word i >> 5 of row [channels]
                                        // [mask_stride] set = sample i of the call is blanked -- and the consumer (the
                                        // down-converter) takes x[i - delay_n - 1] itself and zeroes it under the mask:
                                        // no 8-byte write and re-read per sample, no third input stream here
    const float *hist; float *hist_next;    // [channels][NB_HIST] complex: the last NB_HIST inputs, ping-pong
    int channels, n;
    int ring;                           // the window's magnitudes stay in an LDS ring (frontend_kernels.hip) -- only when
                                        // EVERY channel with the blanker on has noiseblank_ring_min(mask form?) <=
                                        // mag_n + 1 <= noiseblank_ring_max(mask form?)
    int int_ok;                         // mask form on datagrams: every sample the blanker has seen since its set-up came from
                                        // datagrams -- sum and history are integral in units of 2^-8 and the integer kernel
                                        // (noiseblank_mask_int_kernel) decides exactly what the general one does
    int nseg, seg_len;                  // each channel's call is cut into nseg segments of seg_len samples (a
                                        // multiple of 1024, >= 4 blank widths), one workgroup each
};
hipError_t noiseblank_launch(const NbArgs &a, hipStream_t stream);
int noiseblank_ring_min(bool mask);
int noiseblank_ring_max(bool mask);
int noiseblank_tile(bool mask);        // samples per tile of the kernel (its mask form: a.out == nullptr): segment lengths are multiples of it

// packets: [channels][npackets][pkt_len] bytes, pkt_len 1028 (16 bit, 256 samples) or 1444 (24 bit, 240);
// out complex fp32 [channels][out_stride]; dc: optional [channels][2] doubles subtracted (I, Q)
hipError_t unpack_launch(const unsigned char *pk, long chan_stride_bytes, int channels, int npackets, int pkt_len,
                         float *out, long out_stride, const double *dc, hipStream_t stream);
// dc[ch] <- running means of I and Q over n more samples, alpha = 1e-5 (sdrinterface.cpp:829-848)
hipError_t spurcal_launch(const float *iq, long in_stride, int channels, int n, double *dc, hipStream_t stream);

}  // namespace csdr
