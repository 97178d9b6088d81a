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
        x = fminf(fmaxf(x, -32767.0f), 32767.0f);
        y = fminf(fmaxf(y, -32767.0f), 32767.0f);
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

}  // namespace csdr
