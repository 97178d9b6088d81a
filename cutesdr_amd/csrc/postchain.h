// postchain.h -- per-channel state of the sample-rate stages that follow the band-pass filter:
// CSMeter, CAgc, the AM / SAM / FM / SSB demodulators and their CFir / CIir helpers.
// Plain structs living in HBM, shared by the host setup code and the device kernels.
//
// Recurrence state (averagers, PLL phase/frequency, biquad memory) is kept in fp64, the sample
// data path and the transcendentals run in fp32.
#pragma once
#include <hip/hip_runtime.h>
#include "wg_trace.hpp"

namespace csdr {

constexpr int PC_FIR_MAX = 75;          // MAX_NUMCOEF, dsp/fir.h:17
constexpr int PC_AGC_RING = 2048;       // MAX_DELAY_BUF, dsp/agc.h:19

struct PcFir {                          // CFir (dsp/fir.h:20-43)
    int ntaps, pos;                     // pos: ring index of the newest sample
    float coef[PC_FIR_MAX], icoef[PC_FIR_MAX], qcoef[PC_FIR_MAX];
    float zreal[PC_FIR_MAX];            // m_rZBuf: delay line of the real filter
    float zr[PC_FIR_MAX], zi[PC_FIR_MAX];   // m_cZBuf: delay line of the complex filter (shared index)
};

struct PcIir {                          // CIir (dsp/iir.h:17-39), direct form II
    double b0, b1, b2, a1, a2, w1a, w2a, w1b, w2b;
};

struct PcSMeter {                       // CSMeter (dsp/smeter.h:13-28)
    double att_ave, dec_ave, ave_mag, peak_mag, att_a, dec_a, fs;
};

struct PcAgc {                          // CAgc (dsp/agc.h:19-62); rings live in separate arrays
    int on, hang, dly_n, win_n, hang_time;
    int dly_pos, mag_pos, hang_timer;
    double manual_gain, knee, gain_slope, fixed_gain;
    double att_rise, att_fall, dec_rise, dec_fall;
    double peak, attack_ave, decay_ave;
};

struct PcAm {                           // CAmDemod (dsp/amdemod.h:14-25)
    double z1;
    PcFir fir;
};
struct PcSam {                          // CSamDemod (dsp/samdemod.h:14-32)
    double z1, y1, phase, freq, lo, hi, alpha, beta;
    PcFir fir;
};
struct PcFm {                           // CFmDemod (dsp/fmdemod.h:17-54)
    int squelched;
    double hp_freq, out_gain, err_dc, dc_alpha, phase, freq, lo, hi, alpha, beta;
    double sq_thresh, sq_ave, sq_alpha;
    PcFir hp;
    PcIir lp;
};

enum { PC_MODE_NONE = -1, PC_MODE_AM = 0, PC_MODE_SAM, PC_MODE_FM, PC_MODE_USB, PC_MODE_LSB,
       PC_MODE_CWU, PC_MODE_CWL };      // DEMOD_* (dsp/demodulator.h:20-26)

struct PcChannel {                      // everything CDemodulator owns after the band-pass
    int mode;
    PcSMeter sm;
    PcAgc agc;
    PcAm am;
    PcSam sam;
    PcFm fm;
};

// stage selection of one launch
enum { PC_DO_SMETER = 1, PC_DO_AGC = 2, PC_DO_DEMOD = 4, PC_STEREO = 8, PC_AGC_REAL = 16,
       PC_FM_DEFER = 32,                // FM: the walk leaves the raw audio in the output rows; squelch filter, decision
                                        // and audio low-pass follow as their own burst-parallel launches (fm_squelch_launch)
       PC_AGC_PRE = 64,               // AGC: the log magnitudes and their sliding maximum (agc.cpp:196-231) come from a
                                        // burst-parallel launch in front of the walk (agc_peaks_launch), in pkbuf
       PC_PLL_SEQ = 128,                // FM: an unlocked tile is walked by one thread (no overlapped walks; A/B, tests)
       PC_LEAN = 256 };                 // set by PcUnit::run: no S-meter in this launch, every AGC row has its peaks in pkbuf,
                                        // every FM row's squelch is deferred -- the walk may take its lean instantiation

struct PcArgs {
    PcChannel *chan;                    // [channels]
    float *agc_dly;                     // [channels][PC_AGC_RING] complex (2 floats each)
    float *agc_mag;                     // [channels][PC_AGC_RING]
    const float *in;  long in_stride;   // complex input [channels][in_stride] (floats = 2*stride)
    float *out;       long out_stride;  // mono: float [channels][out_stride]; stereo: complex
    const int *out_rows;                // optional [channels]: output row of each channel
    float *scratch;   long scratch_stride;  // per-channel float scratch, >= burst samples
    int channels, nbursts, burst;       // each channel: nbursts bursts of `burst` samples
    int flags;
    double *sqbuf;                      // PC_FM_DEFER: [channels][nbursts][PC_SQ_REC] doubles of scratch
    int sq_bpw;                         // PC_FM_DEFER: bursts per workgroup of the maps / apply kernels
    float *pkbuf;                       // PC_AGC_PRE: [channels][nbursts * burst] sliding peak of the log magnitudes
    float *magtail;                     // PC_AGC_PRE: [channels][PC_AGC_RING] the call's last win_n-1 log magnitudes
    int pre_bpw;                        // PC_AGC_PRE: bursts per workgroup of the peaks kernel
#ifdef CSDR_WG_TRACE
    WgTraceArg trace;
#endif
};
constexpr int PC_SQ_REC = 12;           // per burst: squelch-average map (2), low-pass state map (4 + 2), decision, start state (2)

hipError_t postchain_launch(const PcArgs &a, hipStream_t stream);
// the deferred part of CFmDemod (fmdemod.cpp:113-152) for the FM channels of a launch made with PC_FM_DEFER
hipError_t fm_squelch_launch(const PcArgs &a, hipStream_t stream);
// CAgc's log magnitudes and sliding maximum of every sample of the call, for the walk that follows (PC_AGC_PRE)
hipError_t agc_peaks_launch(const PcArgs &a, hipStream_t stream);
// CSMeter over the whole call as one scan per receiver (the walk then runs without PC_DO_SMETER)
hipError_t smeter_call_launch(const PcArgs &a, hipStream_t stream);
hipError_t smeter_collect_launch(PcChannel *chan, int channels, const int *rows, float *ave, float *peak, hipStream_t stream);
hipError_t smeter_collect_launch(PcChannel *chan, int channels, const int *rows, double *ave, double *peak, hipStream_t stream);
hipError_t filter_leaf_launch(PcFir *fir, PcIir *iir, const float *in, float *out, int n, int op, hipStream_t stream);

}  // namespace csdr
