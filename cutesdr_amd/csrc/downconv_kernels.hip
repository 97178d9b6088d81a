// downconv_kernels.hip -- NCO mixer + decimate-by-2^n cascade for gfx950 (K2 in DESIGN.md).
//
// Replaces CDownConvert::ProcessData (reference dsp/downconvert.cpp:186-263): the gain-stabilised
// rotating-phasor NCO (:210-216) and the chain of CIC-3 (:444-460), fixed 11-tap (:348-423) and
// generic 15..51-tap (:286-320) half-band decimators picked by SetDataRate (:114-173), batched
// over many channels and fused into ONE pass over HBM: 8 B read per input sample, 8 B written
// per output sample, every intermediate rate lives in LDS.
//
// Every stage is the FIR  y[j] = sum_k h[k] xe[2j+k],  xe = [stage history | stage input]
// (SURVEY App. A.3), so the cascade is a pure feed-forward function of the mixed input stream
// and can be cut anywhere: a workgroup owns one segment of one channel, rebuilds the stage
// histories by running the W >= sum_s (L_s-1) 2^s samples in front of its segment through the
// cascade (outputs discarded), then walks its segment tile by tile with the histories carried
// in LDS.  Segment 0 warms up from the W mixed samples the previous call left behind.
//
// NCO: the reference phasor is e^{j(phi0+(n+1)delta)} times the amplitude a_n of the recurrence
// a_{n+1} = a_n (1.95 - a_n^2), a_0 = 1 (-> sqrt(0.95)).  Here the phase is a 64-bit fixed-point
// accumulator (exact per-sample phase, no drift), re-anchored with an accurate sincospi every
// 16 rows and advanced by one complex multiply per row in between; a_n comes from a 512-entry
// table for a channel's first samples and is constant afterwards.
#include "fft_core.hpp"
#include "downconv_kernels.h"

namespace csdr {

constexpr int DC_T = 256;                  // threads per workgroup
constexpr int DC_ROW = 2 * DC_T;           // samples per row (16 B per lane)
constexpr int DC_TILE = 4096;              // input samples per tile
constexpr int DC_ANCHOR_ROWS = 16;

// e^{j * 2*pi * phase/2^64}
__device__ __forceinline__ v2f phasor_of(unsigned long long phase)
{
    const float halfturns = (float)((int)(phase >> 32)) * 4.6566128730773926e-10f;   // 2^-31
    float s, c;
    sincospif(halfturns, &s, &c);
    return v2f{c, s};
}

// One decimate-by-2 stage with the tap geometry fixed at compile time: L = 3 is the CIC-3
// (downconvert.cpp:444-460), otherwise an L-tap half band whose non-zero taps are the even ones
// (symmetric pairs 2q / L-1-2q) and the centre (downconvert.cpp:286-320, 348-423).  The pair
// coefficients are wave-uniform and stay in scalar registers.
template <int L>
__device__ __forceinline__ void dc_stage(const v2f *xe, v2f *y, int nout, const DcStage &st, int t)
{
    constexpr int NP = (L == 3) ? 2 : (L + 1) / 4;
    float c[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) c[q] = st.c[q];
    const float cc = st.ccoef;
    for (int j = t; j < nout; j += DC_T) {
        const v2f *p = xe + 2 * j;
        v2f acc;
        if (L == 3) {
            acc = (p[0] + p[3]) * c[0] + (p[1] + p[2]) * c[1];
        } else {
            acc = p[(L - 1) / 2] * cc;
#pragma unroll
            for (int q = 0; q < NP; q++) acc += (p[2 * q] + p[L - 1 - 2 * q]) * c[q];
        }
        y[j] = acc;
    }
}

__global__ __launch_bounds__(DC_T)
void downconv_kernel(DcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x;
    const int ns = a.nstages;
    const int wg = blockIdx.x;
    const int ci = wg / a.nseg, seg = wg % a.nseg;
    if (ci >= a.nchan) return;
    const int ch = a.chan_list ? a.chan_list[ci] : ci;
    const DcChan cs = a.chan[ch];

    // LDS regions R_s = [hist_s | stage-s input] at a.roff[s] (host computed); R_ns = tile outputs
    const int *roff = a.roff;
    for (int s = 0; s < ns; s++)
        for (int i = t; i < a.st[s].hist; i += DC_T) lds[roff[s] + i] = v2f{0.f, 0.f};

    const v2f *in = a.in + (long)(a.in_rows ? a.in_rows[ch] : ch) * a.in_stride;
    v2f *out = a.out + (long)ch * a.out_stride;
    const v2f *hist = a.hist + (long)ch * a.hist_stride;
    v2f *hist_next = a.hist_next + (long)ch * a.hist_stride;
    const long seg_start = (long)seg * a.seg_len;
    long seg_end = seg_start + a.seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;
    const v2f rowstep = phasor_of(cs.inc * (unsigned long long)DC_ROW);
    const float a_inf = a.amp[DC_AMP_N - 1];

    // pos: index of the tile's first sample in this call's input (negative inside the warm-up)
    long pos = seg_start - a.W;
    while (pos < seg_end) {
        const bool warm = pos < seg_start;
        const long lim = (warm ? seg_start : seg_end) - pos;
        const int n = (int)(lim < DC_TILE ? lim : DC_TILE);
        v2f *r0 = lds + roff[0] + (ns > 0 ? a.st[0].hist : 0);
        // ---------------- stage-0 input: mix with the NCO (or take the mixed history) -----------
        if (warm && seg == 0) {
            for (int i = t; i < n; i += DC_T) r0[i] = hist[pos + a.W + i];
        } else {
            v2f p0 = {1.f, 0.f}, p1 = {1.f, 0.f};
            int row = 0;
            for (int i0 = 0; i0 < n; i0 += DC_ROW, row++) {
                const int i = i0 + 2 * t;
                const long gi = pos + i;                       // sample index within the call
                if ((row & (DC_ANCHOR_ROWS - 1)) == 0) {
                    p0 = phasor_of(cs.phase + cs.inc * (unsigned long long)(gi + 1));
                    p1 = phasor_of(cs.phase + cs.inc * (unsigned long long)(gi + 2));
                }
                if (i < n) {
                    const v4f v = *reinterpret_cast<const v4f *>(in + gi);
                    v2f x0 = cmul(v2f{v.x, v.y}, p0), x1 = cmul(v2f{v.z, v.w}, p1);
                    const unsigned long long age = cs.age + (unsigned long long)gi;
                    if (age + 1 >= DC_AMP_N) {                  // wave-uniform except at the seam
                        x0 *= a_inf; x1 *= a_inf;
                    } else {
                        x0 *= a.amp[age]; x1 *= a.amp[age + 1];
                    }
                    *reinterpret_cast<v4f *>(&r0[i]) = v4f{x0.x, x0.y, x1.x, x1.y};
                    // the last W mixed samples of the call are the next call's warm-up
                    const long hj = gi - (a.n_in - a.W);
                    if (!warm && hj >= 0 && a.W > 0)
                        *reinterpret_cast<v4f *>(&hist_next[hj]) = v4f{x0.x, x0.y, x1.x, x1.y};
                }
                p0 = cmul(p0, rowstep);
                p1 = cmul(p1, rowstep);
            }
        }
        __syncthreads();
        // ---------------- the cascade, LDS -> LDS ---------------------------------------------
        int len = n;
        for (int s = 0; s < ns; s++) {
            const v2f *xe = lds + roff[s];
            const int hnext = (s + 1 < ns) ? a.st[s + 1].hist : 0;
            v2f *y = lds + roff[s + 1] + hnext;
            const int nout = len >> 1;
            switch (a.kind[s]) {
            case 3:  dc_stage<3>(xe, y, nout, a.st[s], t); break;
            case 11: dc_stage<11>(xe, y, nout, a.st[s], t); break;
            case 15: dc_stage<15>(xe, y, nout, a.st[s], t); break;
            case 19: dc_stage<19>(xe, y, nout, a.st[s], t); break;
            case 23: dc_stage<23>(xe, y, nout, a.st[s], t); break;
            case 27: dc_stage<27>(xe, y, nout, a.st[s], t); break;
            case 31: dc_stage<31>(xe, y, nout, a.st[s], t); break;
            case 35: dc_stage<35>(xe, y, nout, a.st[s], t); break;
            case 39: dc_stage<39>(xe, y, nout, a.st[s], t); break;
            case 43: dc_stage<43>(xe, y, nout, a.st[s], t); break;
            case 47: dc_stage<47>(xe, y, nout, a.st[s], t); break;
            default: dc_stage<51>(xe, y, nout, a.st[s], t); break;
            }
            __syncthreads();
            // slide the history: last hist_s inputs of this stage move to the front
            const int h = a.st[s].hist;
            v2f keep = {0.f, 0.f};
            if (t < h) keep = xe[len + t];
            __syncthreads();
            if (t < h) lds[roff[s] + t] = keep;
            len = nout;
        }
        if (ns > 0) __syncthreads();
        // ---------------- tile outputs -> HBM ------------------------------------------------------
        if (!warm) {
            const v2f *y = lds + roff[ns];
            const long obase = pos >> ns;
            for (int j = t; j < len; j += DC_T) out[obase + j] = y[j];
        }
        __syncthreads();
        pos += n;
    }

    // calls shorter than the warm-up length keep the tail of the old history in front
    if (seg == a.nseg - 1 && a.n_in < a.W)
        for (int j = t; j < a.W - a.n_in; j += DC_T) hist_next[j] = hist[j + a.n_in];
}

int downconv_layout(DcArgs &a)
{
    int o = 0;
    for (int s = 0; s <= a.nstages; s++) {
        a.roff[s] = o;
        o += ((s < a.nstages ? a.st[s].hist : 0) + (DC_TILE >> s) + 1) & ~1;
    }
    return o * 8 + 64;
}

hipError_t downconv_launch(DcArgs &a, hipStream_t stream)
{
    const int lds = downconv_layout(a);
    static int attr_bytes = 0;
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&downconv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(downconv_kernel, dim3(a.nchan * a.nseg), dim3(DC_T), lds, stream, a);
    return hipGetLastError();
}

}  // namespace csdr
