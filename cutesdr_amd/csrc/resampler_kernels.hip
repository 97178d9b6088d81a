// resampler_kernels.hip -- CFractResampler for gfx950 (K5 in DESIGN.md).
//
// Replaces the inner loops of CFractResampler::Resample (reference dsp/fractresampler.cpp:144-352):
// every output sample is a 28-tap dot product of the input with a Blackman-Harris windowed sinc,
// looked up (truncated index, no interpolation) in a 280001-entry table at 10000 points per
// zero crossing.  Outputs are independent once their fractional times are known, so the kernel
// runs one thread per output; the times t_m = t_0 + m*rate are accumulated on the host in fp64
// exactly as the reference does (m_FloatTime += dt), which keeps the output COUNT and every
// table index bit-identical.
#include <hip/hip_runtime.h>
#include "resampler_kernels.h"
#include "ref_constants.hpp"

namespace csdr {

__global__ void resample_kernel(ResampleArgs a)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nout) return;
    const double t = a.times[m];
    const int it = (int)t;
    float ar = 0.f, ai = 0.f;
#pragma unroll 4
    for (int i = 1; i <= RS_PERIODS; i++) {
        const int j = it + i;
        const int k = (int)(((double)j - t) * (double)RS_PTS);        // fractresampler.cpp:166
        const float w = a.sinc[k];
        ar += a.buf[2 * j] * w;
        if (a.cpx) ai += a.buf[2 * j + 1] * w;
    }
    if (a.out_i16) {
        float x = ar * a.gain, y = ai * a.gain;                      // :215-227: scale, clip, truncate
        x = fminf(fmaxf(x, -refc::RS_MAX_SOUNDCARDVAL_F), refc::RS_MAX_SOUNDCARDVAL_F);
        y = fminf(fmaxf(y, -refc::RS_MAX_SOUNDCARDVAL_F), refc::RS_MAX_SOUNDCARDVAL_F);
        if (a.cpx) { a.out_i16[2 * m] = (short)x; a.out_i16[2 * m + 1] = (short)y; }
        else a.out_i16[m] = (short)x;
    } else if (a.cpx) {
        a.out_f32[2 * m] = ar; a.out_f32[2 * m + 1] = ai;
    } else {
        a.out_f32[m] = ar;
    }
}

// slide the last RS_PERIODS input samples to the front (fractresampler.cpp:179-182)
__global__ void resample_tail_kernel(float *buf, int n)
{
    const int i = threadIdx.x;
    float re = 0.f, im = 0.f;
    if (i < RS_PERIODS) { re = buf[2 * (n + i)]; im = buf[2 * (n + i) + 1]; }
    __syncthreads();
    if (i < RS_PERIODS) { buf[2 * i] = re; buf[2 * i + 1] = im; }
}

hipError_t resample_launch(const ResampleArgs &a, int n_in, hipStream_t s)
{
    if (a.nout > 0) hipLaunchKernelGGL(resample_kernel, dim3((a.nout + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(resample_tail_kernel, dim3(1), dim3(64), 0, s, a.buf_rw, n_in);
    return hipGetLastError();
}

// batch form: the chain's mono audio rows, all channels on one clock
__global__ void resample_batch_kernel(ResampleBatchArgs a)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    const float *in = a.in + (long)ch * a.in_stride, *hist = a.hist + (long)ch * RS_PERIODS;
    if (m < a.nout) {
        const double t = a.times[m];
        const int it = (int)t;
        float acc = 0.f;
#pragma unroll 4
        for (int i = 1; i <= RS_PERIODS; i++) {
            const int j = it + i;
            const int k = (int)(((double)j - t) * (double)RS_PTS);    // fractresampler.cpp:166
            acc += (j < RS_PERIODS ? hist[j] : in[j - RS_PERIODS]) * a.sinc[k];
        }
        if (a.out_i16) {
            const float x = fminf(fmaxf(acc * a.gain, -refc::RS_MAX_SOUNDCARDVAL_F), refc::RS_MAX_SOUNDCARDVAL_F);
            a.out_i16[(long)ch * a.out_stride + m] = (short)x;
        } else {
            a.out_f32[(long)ch * a.out_stride + m] = acc;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < RS_PERIODS) {                // next call's history (fractresampler.cpp:179-182)
        const int j = a.n + threadIdx.x;
        a.hist_next[(long)ch * RS_PERIODS + threadIdx.x] = j < RS_PERIODS ? hist[j] : in[j - RS_PERIODS];
    }
}

hipError_t resample_batch_launch(const ResampleBatchArgs &a, hipStream_t s)
{
    const int gx = a.nout > 0 ? (a.nout + 255) / 256 : 1;
    hipLaunchKernelGGL(resample_batch_kernel, dim3(gx, a.channels), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace csdr
