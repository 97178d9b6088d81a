// resampler_kernels.h -- launch interface of the fractional resampler kernel (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

constexpr int RS_PTS = 10000;            // SINC_PERIOD_PTS, dsp/fractresampler.cpp:50
constexpr int RS_PERIODS = 28;           // SINC_PERIODS, :53
constexpr int RS_LEN = RS_PERIODS * RS_PTS + 1;

struct ResampleArgs {
    const float *buf;        // [28 history + n] complex pairs (real data uses .re)
    float *buf_rw;
    const float *sinc;       // [RS_LEN]
    const double *times;     // [nout] output times in input-sample units, relative to buf[0]
    float *out_f32; short *out_i16;
    float gain;
    int nout, cpx;
};
hipError_t resample_launch(const ResampleArgs &a, int n_in, hipStream_t s);

// batch form: mono fp32 rows, every channel on the same clock (one set of output times)
struct ResampleBatchArgs {
    const float *in;  long in_stride;    // [channels][in_stride] real fp32, n valid per row
    const float *hist; float *hist_next; // [channels][RS_PERIODS]: the last 28 inputs, ping-pong
    const float *sinc;                   // [RS_LEN]
    const double *times;                 // [nout], relative to the first history sample
    float *out_f32; short *out_i16;      // [channels][out_stride], one of them
    long out_stride;
    float gain;
    int channels, n, nout;
};
hipError_t resample_batch_launch(const ResampleBatchArgs &a, hipStream_t s);

}  // namespace csdr
