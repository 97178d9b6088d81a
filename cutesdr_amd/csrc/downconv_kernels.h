// downconv_kernels.h -- launch interface of the NCO + decimator cascade kernel (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "wire_format.hpp"
#include "frontend_kernels.h"
#include "wg_trace.hpp"

namespace csdr {

constexpr int DC_MAX_STAGES = 9;     // MAX_DECSTAGES-1, reference dsp/downconvert.h:18
constexpr int DC_MAX_PAIRS = 13;     // 51-tap half band: 13 symmetric even-tap pairs + centre
constexpr int DC_AMP_N = 512;        // NCO amplitude envelope table length
constexpr int DC_TILE_SAMPLES = 512;

typedef float dc_v2f __attribute__((ext_vector_type(2)));

// one decimate-by-2 stage: y[j] = ccoef*xe[2j+center] + sum_q c[q]*(xe[2j+a[q]] + xe[2j+b[q]]),
// xe = [hist samples of stage history | stage input]
struct DcStage {
    int hist, npairs, center;
    float ccoef;
    short a[DC_MAX_PAIRS], b[DC_MAX_PAIRS];
    float c[DC_MAX_PAIRS];
};

struct DcChan {                      // per-channel NCO state at the start of the call
    unsigned long long phase;        // angle of the stored phasor Osc1, 2^64 = one turn
    unsigned long long inc;          // per-sample increment
    unsigned long long age;          // samples mixed so far (selects the amplitude a_n)
};

struct DcArgs {
    const dc_v2f *in;  long in_stride;       // raw IQ [channels][in_stride]; unused when wire.pk is set
    WireIn wire;                             // optional: the call's samples as datagrams (wire_format.hpp)
    dc_v2f *out;       long out_stride;      // [channels][out_stride], n_in >> nstages valid
    const dc_v2f *hist; dc_v2f *hist_next;   // [channels][hist_stride], first W valid: mixed samples, ping-pong
    long hist_stride;
    const DcChan *chan;                      // [channels] NCO state at the start of the call
    DcChan *chan_next;                       // [channels] state after it, written by segment 0 of every channel
                                             // (ping-pong like the history: no workgroup reads what another writes)
    const int *chan_list;                    // optional [nchan] channel ids of this launch
    const int *in_rows;                      // optional [channels]: input row of each channel id
    const float *amp;                        // [DC_AMP_N] amplitude envelope a_n
    // Optional noise blanker in front (CNoiseProc::ProcessBlanker, dsp/noiseproc.cpp:121-176, in MASK MODE:
    // noiseblank_kernel has left one bit per sample): sample i of the call, as the reference's in-place blanker would
    // have handed it over (interface/sdrinterface.cpp:884), is
    //     mask bit i set ? 0 : raw[i - delay_n - 1]          raw[j < 0] = nb_hist[NB_HIST + j], the inputs before this call
    // taken in this kernel's own loads -- no blanked copy of the input is ever written.  All indexed by INPUT ROW.
    const unsigned *nb_mask; long nb_mask_stride;   // [rows][stride] words; nullptr: no blanker
    const NbChan *nb_state;                         // [rows]: on (off: no delay, mask all zero), delay_n
    const dc_v2f *nb_hist;                          // [rows][NB_HIST]
    int nchan, n_in, nstages, W, seg_len, nseg;
    int roff[DC_MAX_STAGES + 2];             // LDS region of each stage's input (the last one: tile outputs)
    int ooff[DC_MAX_STAGES + 1];             // offset of the odd-sample half inside region s
    int kind[DC_MAX_STAGES];                 // 3 = CIC3, otherwise the half-band length (11, 15, .. 51)
    DcStage st[DC_MAX_STAGES];
#ifdef CSDR_WG_TRACE
    WgTraceArg trace;
#endif
};

// what a caller hands over to have the blanker's mask applied (csdr__downconvert_batch_process_rows)
struct DcBlank {
    const unsigned *mask; long mask_stride;
    const void *state;                   // [rows] NbChan
    const float *hist;                   // [rows][NB_HIST] complex
};

hipError_t downconv_launch(DcArgs &a, hipStream_t stream);
int downconv_force_dynamic(int on);     // tests: 1 = always the run-time-plan kernel, -1 = query; returns #precompiled plans

}  // namespace csdr
