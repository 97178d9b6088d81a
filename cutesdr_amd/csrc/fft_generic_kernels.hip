// fft_generic_kernels.hip -- CFft for the sizes outside the single-pass kernels' range
// (512, 1024, 32768, 65536; the reference accepts 512..65536, dsp/fft.h:21-22, fft.cpp:140-145).
//
// These sizes are display-rate only (no FastFIR uses them), so the transform is a plain Stockham
// autosort radix-2 through HBM: log2 N launches ping-ponging between two work buffers, natural
// order in and out, twiddles from sincospi of an exact dyadic argument.  Window/swap, the running
// power mean and the log10 of CFft::PutInDisplayFFT / CpxFFT (fft.cpp:267-288, 562-589) are two
// element-wise kernels around it with exactly the arithmetic of spectrum_kernel.
#include "fft_core.hpp"
#include "spectrum_kernels.h"
#include "ref_constants.hpp"

namespace csdr {

// one radix-2 DIF Stockham pass over `batch` transforms of n0 points: the current sub-transform
// length is n (m = n/2), the stride s = n0/n.  y[q + s(2p)] = a + b, y[q + s(2p+1)] = (a - b) w^p
__global__ void stockham_pass_kernel(const v2f *x, v2f *y, int n0, int n, int s, float sign, long stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;              // < n0/2
    if (i >= n0 / 2) return;
    x += (long)blockIdx.y * stride; y += (long)blockIdx.y * stride;
    const int m = n >> 1, q = i & (s - 1), p = i / s;
    const v2f a = x[q + s * p], b = x[q + s * (p + m)];
    float sn, cs;
    sincospif(2.0f * (float)p / (float)n, &sn, &cs);                   // p/n is dyadic: exact argument
    const v2f w = {cs, sign * sn};
    y[q + s * (2 * p)] = a + b;
    y[q + s * (2 * p + 1)] = cmul(a - b, w);
}

// frame f of every channel: window, I/Q swap (fft.cpp:280-281), overload flag (fft.cpp:275-276)
__global__ void spec_prep_kernel(SpectrumArgs a, int n, int frame, v2f *work, long wstride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (i >= n) return;
    const v2f s = reinterpret_cast<const v2f *>(a.in)[(long)ch * a.in_stride + (long)frame * n + i];
    const float w = a.win[i];
    if (s.x > refc::FFT_OVER_LIMIT_F) a.overload[ch] = 1;
    work[(long)ch * wstride + i] = v2f{w * s.y, w * s.x};
}
__global__ void spec_count_kernel(SpectrumArgs a)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= a.channels) return;
    a.counters[2 * ch + 1]++;                                        // CpxFFT counters, fft.cpp:515-517
    if (a.counters[2 * ch] < a.ave_size) a.counters[2 * ch]++;
}
// power, running mean, log10, display order (fft.cpp:564-589)
__global__ void spec_finish_kernel(SpectrumArgs a, int n, const v2f *X, long wstride)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (k >= n) return;
    const int ave_count = a.counters[2 * ch], total = a.counters[2 * ch + 1];
    const v2f v = X[(long)ch * wstride + k];
    const int j = (k + n / 2) & (n - 1);
    const long o = (long)ch * n + j;
    const float p = v.x * v.x + v.y * v.y;
    float sm = a.sum[o];
    if (total <= a.ave_size) sm = sm + p;
    else sm = sm - a.pwr[o] + p;
    a.sum[o] = sm;
    const float m = sm / (float)ave_count;
    a.pwr[o] = m;
    a.ave[o] = (float)((double)log10f(m + a.kc) + a.kb);
}

// transforms `batch` rows of work buffer A (stride wstride, B = A + half) ; returns where the result is
static const v2f *stockham_run(int log2n, float sign, v2f *A, v2f *B, long wstride, int batch, hipStream_t st)
{
    const int n0 = 1 << log2n;
    v2f *x = A, *y = B;
    for (int n = n0, s = 1; n > 1; n >>= 1, s <<= 1) {
        hipLaunchKernelGGL(stockham_pass_kernel, dim3(n0 / 2 / 256, batch), dim3(256), 0, st, x, y, n0, n, s, sign, wstride);
        v2f *t = x; x = y; y = t;
    }
    return x;
}

// work: [2][channels][N] complex
hipError_t spectrum_generic_launch(int log2n, const SpectrumArgs &a, float *work, hipStream_t st)
{
    const int n = 1 << log2n;
    v2f *A = reinterpret_cast<v2f *>(work), *B = A + (long)a.channels * n;
    for (int f = 0; f < a.nframes; f++) {
        hipLaunchKernelGGL(spec_prep_kernel, dim3(n / 256, a.channels), dim3(256), 0, st, a, n, f, A, (long)n);
        hipLaunchKernelGGL(spec_count_kernel, dim3((a.channels + 63) / 64), dim3(64), 0, st, a);
        const v2f *X = stockham_run(log2n, +1.0f, A, B, n, a.channels, st);
        hipLaunchKernelGGL(spec_finish_kernel, dim3(n / 256, a.channels), dim3(256), 0, st, a, n, X, (long)n);
    }
    return hipGetLastError();
}

// plain N-point transform, natural order; `in` may equal `out`; work: [2][N] complex
hipError_t fft_generic_plain_launch(int log2n, int sign, const float *in, float *out, float *work, hipStream_t st)
{
    const int n = 1 << log2n;
    v2f *A = reinterpret_cast<v2f *>(work), *B = A + n;
    hipError_t e = hipMemcpyAsync(A, in, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    const v2f *X = stockham_run(log2n, sign > 0 ? 1.0f : -1.0f, A, B, n, 1, st);
    e = hipMemcpyAsync(out, X, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// ---------------- screen mapping for a batch of spectra ----------------
__device__ __forceinline__ int screen_level(const ScreenArgs &a, const float *ave, int bin)
{
    int b = a.invert ? (a.n - bin) : bin;
    if (b >= a.n) b = a.n - 1;                          // the reference reads one past the end here
    int v = (int)((double)a.max_h * a.gain * ((double)ave[b] - a.off));
    return v < 0 ? 0 : (v > a.max_h ? a.max_h : v);
}
// more bins than pixels: pixel x shows the smallest y (= strongest bin) of the bins that map to it
// (fft.cpp:372-389: first bin of a pixel sets it, later smaller values replace it)
__global__ void screen_bins_kernel(ScreenArgs a)
{
    const int i = a.bin_min + blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (i > a.bin_max) return;
    const int x = ((i - a.bin_min) * a.plot_w) / (a.bin_max - a.bin_min);
    if (x >= a.plot_w) return;                          // the reference writes OutBuf[MaxWidth] here
    atomicMin(&a.out[(long)ch * a.out_stride + x], screen_level(a, a.ave + (long)ch * a.n, i));
}
__global__ void screen_fill_kernel(ScreenArgs a, int value)
{   // pixels that receive at least one bin start from "+infinity" for the atomicMin
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (x >= a.plot_w) return;
    // pixel x receives a bin iff some i in [bin_min, bin_max] has ((i-bin_min)*plot_w)/(range) == x
    const long range = a.bin_max - a.bin_min;
    const long lo = ((long)x * range + a.plot_w - 1) / a.plot_w;       // smallest i-bin_min with value >= x
    if (lo <= range && (lo * a.plot_w) / range == x) a.out[(long)ch * a.out_stride + x] = value;
}
// at least as many pixels as bins: pixel x shows bin bin_min + x*range/plot_w (fft.cpp:391-407)
__global__ void screen_pixels_kernel(ScreenArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (x >= a.plot_w) return;
    const int xi = x < a.n ? x : a.n - 1;               // the translate table has n entries
    const int bin = a.bin_min + (xi * (a.bin_max - a.bin_min)) / a.plot_w;
    a.out[(long)ch * a.out_stride + x] = screen_level(a, a.ave + (long)ch * a.n, bin);
}

hipError_t screen_launch(const ScreenArgs &a, hipStream_t st)
{
    if (a.plot_w <= 0) return hipSuccess;
    if ((a.bin_max - a.bin_min) > a.plot_w) {
        hipLaunchKernelGGL(screen_fill_kernel, dim3((a.plot_w + 255) / 256, a.channels), dim3(256), 0, st, a, 0x7fffffff);
        const int nb = a.bin_max - a.bin_min + 1;
        hipLaunchKernelGGL(screen_bins_kernel, dim3((nb + 255) / 256, a.channels), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(screen_pixels_kernel, dim3((a.plot_w + 255) / 256, a.channels), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

// ---------------- waterfall line: CPlotter::draw's palette look-up (gui/plotter.cpp:436-441) ----------------
__host__ __device__ static inline unsigned plotter_color_of(int i)
{   // the constructor's ramp (gui/plotter.cpp:67-83): blue -> cyan -> green -> yellow -> red -> magenta-ish
    int r = 0, g = 0, b = 0;
    if (i < 43) { b = 255 * i / 43; }
    else if (i < 87) { g = 255 * (i - 43) / 43; b = 255; }
    else if (i < 120) { g = 255; b = 255 - (255 * (i - 87) / 32); }
    else if (i < 154) { r = 255 * (i - 120) / 33; g = 255; }
    else if (i < 217) { r = 255; g = 255 - (255 * (i - 154) / 62); }
    else { r = 255; b = 128 * (i - 217) / 38; }
    return 0xff000000u | ((unsigned)(r & 255) << 16) | ((unsigned)(g & 255) << 8) | (unsigned)(b & 255);
}
unsigned plotter_color(int i) { return plotter_color_of(i < 0 ? 0 : (i > 255 ? 255 : i)); }

__global__ void waterfall_color_kernel(const int *levels, long stride, unsigned *rgb, long rgb_stride, int w)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (x >= w) return;
    const int y = levels[(long)ch * stride + x];
    if (y < 0) return;                                   // no bin maps to this pixel
    rgb[(long)ch * rgb_stride + x] = plotter_color_of(255 - (y > 255 ? 255 : y));
}
hipError_t waterfall_color_launch(const int *levels, long stride, unsigned *rgb, long rgb_stride, int w, int channels,
                                  hipStream_t st)
{
    if (w <= 0) return hipSuccess;
    hipLaunchKernelGGL(waterfall_color_kernel, dim3((w + 255) / 256, channels), dim3(256), 0, st, levels, stride, rgb, rgb_stride, w);
    return hipGetLastError();
}

}  // namespace csdr
