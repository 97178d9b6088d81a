// spectrum_kernels.h -- launch interface of the CFft kernels (internal).
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

struct SpectrumArgs {
    const float *in;  long in_stride;   // complex fp32 [channels][in_stride], frames back to back
    const float *win;                   // [N] window (Hann*2, dsp/fft.cpp:196-198)
    const float *tw1, *tw2;             // twiddle tables as for the FastFIR kernel
    float *sum, *pwr;                   // [channels][N] running sum / mean power (display order)
    float *ave;                         // [channels][N] log10(mean + K_C) + K_B, in bels
    int *counters;                      // [channels][2]: ave_count, total_count
    int *overload;                      // [channels]
    int channels, nframes, ave_size;
    int nparts;                         // frame groups per channel (1: the frames of a channel run in one workgroup)
    float *part;                        // [channels][nparts][N] partial sums, nparts > 1 only
    float *alpha;                       // [channels][nparts] group weights + [channels] average count after the call (nparts > 1)
    float kc; double kb;
};
hipError_t spectrum_launch(int log2n, const SpectrumArgs &a, hipStream_t stream);

// plain transform of one N-point block: sign=+1 CFft::FwdFFT, -1 RevFFT; natural order in/out
hipError_t fft_plain_launch(int log2n, int sign, const float *in, float *out, const float *tw1,
                            const float *tw2, hipStream_t stream);

// CFft::GetScreenIntegerFFTData (fft.cpp:308-410) for every channel: ave [channels][n] bels (display order)
// -> out [channels][out_stride] pixels; the bin range / pixel count come precomputed from the host
struct ScreenArgs {
    const float *ave; int *out; long out_stride;
    int n, channels, bin_min, bin_max, plot_w, max_h, invert;
    double off, gain;                   // MaxdB/10 and -10/(MaxdB-MindB)
};
hipError_t screen_launch(const ScreenArgs &a, hipStream_t stream);
// CPlotter's palette entry i (gui/plotter.cpp:67-83) as 0xFFRRGGBB
unsigned plotter_color(int i);
// levels [channels][stride] (0..255, or < 0 = pixel not touched) -> palette[255 - level] in rgb [channels][rgb_stride]
hipError_t waterfall_color_launch(const int *levels, long stride, unsigned *rgb, long rgb_stride, int w, int channels,
                                  hipStream_t stream);

// sizes outside 2048..16384 (512, 1024, 32768, 65536): multi-launch transform through HBM.
// work: [2][channels][N] complex for the spectrum, [2][N] for the plain transform
hipError_t spectrum_generic_launch(int log2n, const SpectrumArgs &a, float *work, hipStream_t stream);
hipError_t fft_generic_plain_launch(int log2n, int sign, const float *in, float *out, float *work, hipStream_t stream);

}  // namespace csdr
