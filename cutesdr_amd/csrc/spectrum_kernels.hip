// spectrum_kernels.hip -- CFft display spectrum and plain N-point transforms for gfx950 (K3).
//
// Replaces CFft::PutInDisplayFFT (reference dsp/fft.cpp:267-288: Hann*2 window, I/Q swap, forward
// transform) together with the tail of CFft::CpxFFT (:562-589: |X|^2, running mean over AveSize
// frames, log10, fft-shifted into display order), and CFft::FwdFFT / RevFFT (:416-426).
// One workgroup of N/32 threads per channel walks that channel's frames in order (the running
// mean makes frames sequential); the transform is a three-pass decimation-in-time FFT with the FMA
// butterflies of the overlap-save kernel (fft_core.hpp), its output is scattered straight
// into display order.  8 B in + 4 B out per bin.
#define CSDR_FMA_BFLY 1          // FMA-form decimation-in-time butterflies (fft_core.hpp)
#define CSDR_PLAIN_CONST_FMA 1
#include "launch_once.hpp"
#include "fft_core.hpp"
#include "spectrum_kernels.h"
#include "ref_constants.hpp"

namespace csdr {

#define K3_SB() __builtin_amdgcn_sched_barrier(0)

template <int LOG2N>
struct SpecCfg {
    static constexpr int N = 1 << LOG2N, T = N / 32, R0 = N / 1024, G = 32 / R0;
    static constexpr int LDS_DATA = N + 2 * (N / 32);
    static constexpr int LDS_BYTES = (LDS_DATA + 1024) * 8;
    // the G columns (of the 1024 x R0 input matrix) a thread owns in pass A, in PAIRS that the workgroup's threads
    // take side by side: column pair t + T k, k < G/2.  A wave's load of one element pair is then 64 x 16 contiguous
    // bytes (with G consecutive columns per thread -- 64 bytes at N = 4096 -- every 128-byte line was consumed by four
    // separate load instructions; measured: no difference in time, the line sat in L2 either way).
#ifdef CSDR_SPEC_CONSECUTIVE_COLUMNS
    static __device__ __forceinline__ int col(int t, int e) { return G * t + e; }
#else
    static __device__ __forceinline__ int col(int t, int e) { return 2 * (t + T * (e >> 1)) + (e & 1); }
#endif
};

// forward (positive exponent) transform of the block held as x[e*R0+n1] <-> sample 1024*n1+Cfg::col(t,e);
// on return x[k2] is spectrum bin  (t>>5) + R0*((t&31) + 32*k2).
// Three decimation-in-time passes with FMA-form butterflies (fft_core.hpp, as in the round-2 overlap-save kernel):
// the bit-reversed input order a DIT network wants costs nothing -- pass A's samples sit in registers, passes B and C
// read their points from LDS in any order -- and its outputs come out in natural order.  The 1024-point
// sub-transform k0 lives in ONE half-wave (threads 32 k0 .. 32 k0 + 31), so the exchange between passes B and C needs
// no workgroup barrier, only the wave's own program order.
template <int LOG2N>
__device__ __forceinline__ void fft_fwd_passes(v2f (&x)[32], v2f *lds, const v2f *tw2, const v2f *w1)
{
    using Cfg = SpecCfg<LOG2N>;
    constexpr int R0 = Cfg::R0, G = Cfg::G;
    const int t = threadIdx.x;
    // ---- pass A: radix-R0 over the rows n1 of column G t + e, outer twiddle W_N^{(G t + e) k0}
#pragma unroll
    for (int e = 0; e < G; e++) {
        v2f y[R0];
        static_for<0, R0>([&](auto N1) { y[bitrev<R0>(N1.value)] = x[e * R0 + N1.value]; });
        dft_dit<R0, +1>(y);
        v2f pw[R0];
        twiddle_powers<R0>(opaque(w1[e]), pw);
        static_for<1, R0>([&](auto K0) { y[K0.value] = cmul(y[K0.value], pw[K0.value]); });
#pragma unroll
        for (int i = 0; i < R0; i++) x[e * R0 + i] = y[i];
    }
    __syncthreads();                       // the previous transform's pass C has read its rows
    static_for<0, R0>([&](auto K0) {
        constexpr int k0 = K0.value;
#pragma unroll
        for (int e = 0; e < G; e++) lds[lds_pad(1024 * k0 + Cfg::col(t, e))] = x[e * R0 + k0];   // (a column pair shares a 32-group: adjacent)
    });
    __syncthreads();
    // ---- pass B: radix-32 over the 32 points of column sn of sub-transform sb, twiddle W_1024^{sn k1}, in place.
    // Cut into groups of four points like the passes of the overlap-save kernel (fft_core.hpp: head4 / tail): the
    // points are fetched in the order the first stages need them, three groups ahead of the butterflies; a tail
    // group's results are stored while the next group's butterflies issue; sched_barrier pins that order.
    const int sb = t >> 5, sn = t & 31;
    v2f *const col = lds + lds_pad(1024 * sb) + sn;          // point n1 at col[34 * n1]
    const v2f *const twc = tw2 + sn;                         // twiddle k1 at twc[32 * k1]
    {
        auto fetch = [&](auto Gg) {
            static_for<0, 4>([&](auto Q) {
                constexpr int p = 4 * Gg.value + Q.value;
                x[p] = lds_ld8(col + 34 * bitrev<32>(p));
            });
        };
        static_for<0, 3>(fetch);
        K3_SB();
        static_for<0, 8>([&](auto Gg) {
            if constexpr (Gg.value + 3 < 8) fetch(std::integral_constant<int, Gg.value + 3>{});
            dit_head4<Gg.value, 32, +1>(x);
            if constexpr ((Gg.value & 1) == 1) K3_SB();
        });
        dit_single<8, 32, +1>(x);
        K3_SB();
        v2f tw[2][4];
        static_for<1, 4>([&](auto P) { tw[0][P.value] = lds_ld8(twc + 32 * (8 * P.value)); });
        K3_SB();
        static_for<0, 9>([&](auto Ii) {
            constexpr int i = Ii.value;                      // tail group i finishes k1 = i, i+8, i+16, i+24
            if constexpr (i < 7)
                static_for<0, 4>([&](auto P) { tw[(i + 1) & 1][P.value] = lds_ld8(twc + 32 * (i + 1 + 8 * P.value)); });
            if constexpr (i < 8) {
                dit_tail<i, 32, +1>(x);
                static_for<0, 4>([&](auto P) {
                    constexpr int k1 = i + 8 * P.value;
                    if constexpr (k1 != 0) x[k1] = cmul(x[k1], tw[i & 1][P.value]);
                });
            }
            if constexpr (i > 0)
                static_for<0, 4>([&](auto P) {
                    constexpr int k1 = (i - 1) + 8 * P.value;
                    lds_st8(col + 34 * k1, x[k1]);
                });
            K3_SB();
        });
    }
    // B -> C stays inside the half-wave that owns sub-transform sb
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- pass C: radix-32 over the 32 consecutive points of row t, read as 16-byte pairs: rows q, q+4, q+8, q+12 of
    // pairs hold the inputs of head groups bitrev3(2q) and bitrev3(2q+1)
    {
        const v2f *const rowp = lds + 34 * t;
        static_for<0, 4>([&](auto Q) {
            static_for<0, 4>([&](auto P) {
                constexpr int j = Q.value + 4 * P.value;
                const v4f v = *reinterpret_cast<const v4f *>(rowp + 2 * j);
                x[bitrev<32>(2 * j)] = v2f{v.x, v.y};
                x[bitrev<32>(2 * j + 1)] = v2f{v.z, v.w};
            });
        });
        K3_SB();
        static_for<0, 4>([&](auto Q) {
            dit_head4<bitrev<8>(2 * Q.value), 32, +1>(x);
            dit_head4<bitrev<8>(2 * Q.value + 1), 32, +1>(x);
            K3_SB();
        });
        dit_single<8, 32, +1>(x);
        static_for<0, 8>([&](auto Ii) { dit_tail<Ii.value, 32, +1>(x); });
    }
}

#ifndef CSDR_SPEC_WAVES
#define CSDR_SPEC_WAVES 1
#endif
template <int LOG2N>
__global__ __launch_bounds__(SpecCfg<LOG2N>::T) __attribute__((amdgpu_waves_per_eu(CSDR_SPEC_WAVES)))
void spectrum_kernel(SpectrumArgs a)
{
    using Cfg = SpecCfg<LOG2N>;
    constexpr int N = Cfg::N, T = Cfg::T, R0 = Cfg::R0, G = Cfg::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    v2f *tw2 = lds + Cfg::LDS_DATA;
    // With nparts > 1 the frames of a channel are cut into nparts groups, one workgroup each: the
    // running sum is the linear map sum <- alpha_f sum + p_f, so a group accumulates its frames from
    // zero into `part` and spectrum_combine_kernel folds the groups in order.
    const int t = threadIdx.x, ch = blockIdx.x / a.nparts, part = blockIdx.x % a.nparts;
    const int f0 = (int)((long)a.nframes * part / a.nparts), f1 = (int)((long)a.nframes * (part + 1) / a.nparts);
    for (int i = t; i < 1024; i += T) tw2[i] = reinterpret_cast<const v2f *>(a.tw2)[i];
    v2f w1[G];
#pragma unroll
    for (int e = 0; e < G; e++) w1[e] = reinterpret_cast<const v2f *>(a.tw1)[Cfg::col(t, e)];
    const v2f *in = reinterpret_cast<const v2f *>(a.in) + (long)ch * a.in_stride;
    float *sum = a.sum + (long)ch * N, *pwr = a.pwr + (long)ch * N, *ave = a.ave + (long)ch * N;
    int ave_count = a.counters[2 * ch], total = a.counters[2 * ch + 1];
    total += f0;                                              // counters at this group's first frame
    ave_count = ave_count + f0 < a.ave_size ? ave_count + f0 : (ave_count > a.ave_size ? ave_count : a.ave_size);
    int over = 0;
    // every thread owns the same 32 bins in every frame: the running sum and mean stay in registers
    // for the whole call, only the last frame's bels are written
    int tt = t;
    asm volatile("" : "+v"(tt));              // keep the scattered addresses out of LICM's hands
    const int k0 = tt >> 5, k1 = tt & 31;
    float sm[32];                             // the mean is sum / count: recomputed, not carried
    static_for<0, 32>([&](auto Rr) {
        constexpr int r = Rr.value, k2 = r;
        const int j = ((k0 + R0 * (k1 + 32 * k2)) + N / 2) & (N - 1);     // display order, fft.cpp:564-589
        sm[r] = a.nparts == 1 ? sum[j] : 0.f;
    });
    // the samples of frame f+1 are fetched while frame f goes through its transform (one workgroup has only two
    // waves and three share a CU: nothing else would hide the HBM latency of a frame's 32 loads per thread)
    // (not at N = 16384: a workgroup is 512 threads there, two waves per SIMD and so at most 256 registers per wave;
    // with the prefetched frame on top of the points and the running sums the kernel spilled 80 of them -- 324 bytes
    // of scratch per lane -- and ran at 1.5 TB/s; the second wave of the SIMD hides the loads instead: round 4)
    constexpr bool PREFETCH = LOG2N < 14;
    v2f nxt[PREFETCH ? 32 : 1];
    auto fetch = [&](int f) {
        if constexpr (PREFETCH) {
            const v2f *src = in + (long)f * N;
#pragma unroll
            for (int e = 0; e < G; e++)
#pragma unroll
                for (int n1 = 0; n1 < R0; n1++) nxt[e * R0 + n1] = src[1024 * n1 + Cfg::col(t, e)];
        }
    };
    if (f0 < f1) fetch(f0);
    for (int f = f0; f < f1; f++) {
        v2f x[32];
#pragma unroll
        for (int e = 0; e < G; e++)
#pragma unroll
            for (int n1 = 0; n1 < R0; n1++) {
                const int i = 1024 * n1 + Cfg::col(t, e);
                v2f s;
                if constexpr (PREFETCH) s = nxt[e * R0 + n1]; else s = in[(long)f * N + i];
                const float w = a.win[i];
                if (s.x > refc::FFT_OVER_LIMIT_F) over = 1;                     // OVER_LIMIT, fft.cpp:30,275
                x[e * R0 + n1] = v2f{w * s.y, w * s.x};           // I/Q swapped, fft.cpp:280-281
            }
        if (f + 1 < f1) fetch(f + 1);
        const float prev_count = (float)ave_count;
        total++;                                                  // CpxFFT counters, fft.cpp:515-517
        if (ave_count < a.ave_size) ave_count++;
        fft_fwd_passes<LOG2N>(x, lds, tw2, w1);        // (its own barrier keeps it behind the previous frame's pass C)
        const float inv_prev = 1.0f / prev_count;      // (one division per frame instead of one per bin, as in spectrum16_kernel)
        static_for<0, 32>([&](auto Rr) {
            constexpr int r = Rr.value;
            const float p = x[r].x * x[r].x + x[r].y * x[r].y;
            if (total <= a.ave_size) sm[r] = sm[r] + p;
            else sm[r] = sm[r] - sm[r] * inv_prev + p;            // minus the previous mean (fft.cpp:570-574)
        });
    }
    if (a.nparts > 1) {
        float *dst = a.part + ((long)ch * a.nparts + part) * N;
        static_for<0, 32>([&](auto Rr) {
            constexpr int r = Rr.value, k2 = r;
            dst[((k0 + R0 * (k1 + 32 * k2)) + N / 2) & (N - 1)] = sm[r];
        });
    } else if (a.nframes > 0) {
        static_for<0, 32>([&](auto Rr) {
            constexpr int r = Rr.value, k2 = r;
            const int j = ((k0 + R0 * (k1 + 32 * k2)) + N / 2) & (N - 1);
            const float m = sm[r] / (float)ave_count;
            sum[j] = sm[r]; pwr[j] = m;
            ave[j] = (float)((double)log10f(m + a.kc) + a.kb);
            if constexpr ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        });
    }
    if (t == 0 && a.nparts == 1) { a.counters[2 * ch] = ave_count; a.counters[2 * ch + 1] = total; }
    if (over) a.overload[ch] = 1;
}

// ---- the 4096-point display spectrum (BASELINE config C1) as 256 threads x 16 points -----------------------------
// spectrum_kernel<12> holds 32 points, the prefetched next frame and 32 running sums per thread: 300 registers, ONE
// wave per SIMD, its vector unit 42 % busy with nothing to overlap a wave's own LDS and memory waits.  Sixteen points
// per thread fit two waves per SIMD.  N = 16 x 16 x 16, three radix-16 DIT passes in registers:
//   A  thread t: samples 256 a + t (one coalesced 8-byte load per point), DFT over a, twiddle W_N^{t ka}   -> (ka, t)
//   B  thread (ka, c): points (ka, 16 b + c), DFT over b, twiddle W_256^{c kb}, back to the same 16 places
//   C  thread (ka, kb): its 16 consecutive points, DFT over c                    -> bin ka + 16 kb + 256 kc
// B -> C stays inside the quarter-wave that owns ka: no workgroup barrier.  Rows of 16 points are 18 apart in LDS
// (pass C reads them as 16-byte pieces without bank conflicts).
__device__ __forceinline__ int pad16(int pos) { return pos + ((pos >> 4) << 1); }
constexpr int SPEC16_LDS = (4096 + 2 * 256) * 8;
#ifndef CSDR_SPEC16_WAVES
#define CSDR_SPEC16_WAVES 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CSDR_SPEC16_WAVES)))
void spectrum16_kernel(SpectrumArgs a)
{
    constexpr int N = 4096, T = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x, ch = blockIdx.x / a.nparts, part = blockIdx.x % a.nparts;
    const int f0 = (int)((long)a.nframes * part / a.nparts), f1 = (int)((long)a.nframes * (part + 1) / a.nparts);
    const v2f *tw1 = reinterpret_cast<const v2f *>(a.tw1);           // W_N^n, n < 1024
    const v2f wA = tw1[t], wB = tw1[16 * (t & 15)];                  // W_N^t and W_256^c
    const v2f *in = reinterpret_cast<const v2f *>(a.in) + (long)ch * a.in_stride;
    float *sum = a.sum + (long)ch * N, *pwr = a.pwr + (long)ch * N, *ave = a.ave + (long)ch * N;
    int ave_count = a.counters[2 * ch], total = a.counters[2 * ch + 1];
    total += f0;                                              // counters at this group's first frame
    ave_count = ave_count + f0 < a.ave_size ? ave_count + f0 : (ave_count > a.ave_size ? ave_count : a.ave_size);
    int over = 0;
    int tt = t;
    asm volatile("" : "+v"(tt));              // keep the scattered addresses out of LICM's hands
    const int kbin = (tt >> 4) + 16 * (tt & 15);              // this thread's bins: kbin + 256 kc
    float sm[16], wn[16];
    static_for<0, 16>([&](auto Rr) {
        constexpr int r = Rr.value;
        sm[r] = a.nparts == 1 ? sum[((kbin + 256 * r) + N / 2) & (N - 1)] : 0.f;   // display order, fft.cpp:564-589
        wn[r] = a.win[256 * r + t];
    });
    v2f pwA[16];                              // W_N^{t ka}: resident (30 registers; pass B's are recomputed per frame)
    twiddle_powers<16>(wA, pwA);
    v2f nxt[16];
    auto fetch = [&](int f) {
        const v2f *src = in + (long)f * N + t;
#pragma unroll
        for (int q = 0; q < 16; q++) nxt[q] = src[256 * q];
    };
    if (f0 < f1) fetch(f0);
    for (int f = f0; f < f1; f++) {
        v2f x[16];
        static_for<0, 16>([&](auto Q) {
            constexpr int q = Q.value;
            const v2f s_ = nxt[q];
            if (s_.x > refc::FFT_OVER_LIMIT_F) over = 1;                        // OVER_LIMIT, fft.cpp:30,275
            x[bitrev<16>(q)] = v2f{wn[q] * s_.y, wn[q] * s_.x};    // I/Q swapped, fft.cpp:280-281
        });
        if (f + 1 < f1) fetch(f + 1);
        const float prev_count = (float)ave_count;
        total++;                                                  // CpxFFT counters, fft.cpp:515-517
        if (ave_count < a.ave_size) ave_count++;
        // ---- pass A
        dft_dit<16, +1>(x);
        static_for<1, 16>([&](auto K) { x[K.value] = cmul(x[K.value], pwA[K.value]); });
        __syncthreads();                       // the previous frame's pass C has read its rows
        static_for<0, 16>([&](auto K) { lds[pad16(256 * K.value + t)] = x[K.value]; });
        __syncthreads();
        // ---- pass B: (ka, c) = (t >> 4, t & 15)
        {
            v2f *col = lds + pad16(256 * (t >> 4)) + (t & 15);                   // point b at col[18 b]
            static_for<0, 16>([&](auto B) { x[bitrev<16>(B.value)] = lds_ld8(col + 18 * B.value); });
            dft_dit<16, +1>(x);
            v2f pw[16];
            twiddle_powers<16>(opaque(wB), pw);
            static_for<1, 16>([&](auto K) { x[K.value] = cmul(x[K.value], pw[K.value]); });
            static_for<0, 16>([&](auto K) { lds_st8(col + 18 * K.value, x[K.value]); });
        }
        // B -> C stays inside the sixteen threads that own ka
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- pass C: the 16 consecutive points of row t
        {
            const v2f *row = lds + 18 * t;
            static_for<0, 8>([&](auto J) {
                const v4f v = *reinterpret_cast<const v4f *>(row + 2 * J.value);
                x[bitrev<16>(2 * J.value)] = v2f{v.x, v.y};
                x[bitrev<16>(2 * J.value + 1)] = v2f{v.z, v.w};
            });
            dft_dit<16, +1>(x);
        }
        const float inv_prev = 1.0f / prev_count;                 // (one division per frame instead of one per bin: -10 %)
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const float p = x[r].x * x[r].x + x[r].y * x[r].y;
            if (total <= a.ave_size) sm[r] = sm[r] + p;
            else sm[r] = sm[r] - sm[r] * inv_prev + p;            // minus the previous mean (fft.cpp:570-574)
        });
    }
    if (a.nparts > 1) {
        float *dst = a.part + ((long)ch * a.nparts + part) * N;
        static_for<0, 16>([&](auto Rr) { dst[((kbin + 256 * Rr.value) + N / 2) & (N - 1)] = sm[Rr.value]; });
    } else if (a.nframes > 0) {
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const int j = ((kbin + 256 * r) + N / 2) & (N - 1);
            const float m = sm[r] / (float)ave_count;
            sum[j] = sm[r]; pwr[j] = m;
            ave[j] = (float)((double)log10f(m + a.kc) + a.kb);
        });
    }
    if (t == 0 && a.nparts == 1) { a.counters[2 * ch] = ave_count; a.counters[2 * ch + 1] = total; }
    if (over) a.overload[ch] = 1;
}

// ---- the 2048-point display spectrum the same way (round 4): 128 threads x 16 points, N = 16 x 8 x 16 ----------------
// spectrum_kernel<11> is ONE wave of 32 points per thread.  Here: pass A as above (samples 128 a + t), pass B over the
// eight points b of a COLUMN PAIR per thread (thread (ka, c pair): 2 x 8 points, twiddle W_128^{c kb}), pass C the 16
// consecutive points of row t = 8 ka + kb -> bin ka + 16 kb + 128 kc.  B -> C inside the eight threads that own ka.
constexpr int SPEC8_LDS = (2048 + 2 * 128) * 8;
__global__ __launch_bounds__(128)
void spectrum8_kernel(SpectrumArgs a)
{
    constexpr int N = 2048, T = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x, ch = blockIdx.x / a.nparts, part = blockIdx.x % a.nparts;
    const int f0 = (int)((long)a.nframes * part / a.nparts), f1 = (int)((long)a.nframes * (part + 1) / a.nparts);
    const v2f *tw1 = reinterpret_cast<const v2f *>(a.tw1);           // W_N^n, n < 1024
    const int cp = t & 7;                                             // pass B: columns 2 cp, 2 cp + 1 of ka = t >> 3
    const v2f wA = tw1[t], wB0 = tw1[16 * (2 * cp)], wB1 = tw1[16 * (2 * cp + 1)];   // W_N^t; W_128^c = W_N^{16 c}
    const v2f *in = reinterpret_cast<const v2f *>(a.in) + (long)ch * a.in_stride;
    float *sum = a.sum + (long)ch * N, *pwr = a.pwr + (long)ch * N, *ave = a.ave + (long)ch * N;
    int ave_count = a.counters[2 * ch], total = a.counters[2 * ch + 1];
    total += f0;                                              // counters at this group's first frame
    ave_count = ave_count + f0 < a.ave_size ? ave_count + f0 : (ave_count > a.ave_size ? ave_count : a.ave_size);
    int over = 0;
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int kbin = (tt >> 3) + 16 * (tt & 7);               // this thread's bins: kbin + 128 kc
    float sm[16], wn[16];
    static_for<0, 16>([&](auto Rr) {
        constexpr int r = Rr.value;
        sm[r] = a.nparts == 1 ? sum[((kbin + 128 * r) + N / 2) & (N - 1)] : 0.f;   // display order, fft.cpp:564-589
        wn[r] = a.win[128 * r + t];
    });
    v2f pwA[16];
    twiddle_powers<16>(wA, pwA);
    v2f nxt[16];
    auto fetch = [&](int f) {
        const v2f *src = in + (long)f * N + t;
#pragma unroll
        for (int q = 0; q < 16; q++) nxt[q] = src[128 * q];
    };
    if (f0 < f1) fetch(f0);
    for (int f = f0; f < f1; f++) {
        v2f x[16];
        static_for<0, 16>([&](auto Q) {
            constexpr int q = Q.value;
            const v2f s_ = nxt[q];
            if (s_.x > refc::FFT_OVER_LIMIT_F) over = 1;                        // OVER_LIMIT, fft.cpp:30,275
            x[bitrev<16>(q)] = v2f{wn[q] * s_.y, wn[q] * s_.x};    // I/Q swapped, fft.cpp:280-281
        });
        if (f + 1 < f1) fetch(f + 1);
        const float prev_count = (float)ave_count;
        total++;                                                  // CpxFFT counters, fft.cpp:515-517
        if (ave_count < a.ave_size) ave_count++;
        // ---- pass A
        dft_dit<16, +1>(x);
        static_for<1, 16>([&](auto K) { x[K.value] = cmul(x[K.value], pwA[K.value]); });
        __syncthreads();                       // the previous frame's pass C has read its rows
        static_for<0, 16>([&](auto K) { lds[pad16(128 * K.value + t)] = x[K.value]; });
        __syncthreads();
        // ---- pass B: (ka, column pair) = (t >> 3, t & 7): point (b, c) at lds[18 (8 ka + b) + c]
        {
            v2f *cell = lds + 18 * (8 * (t >> 3)) + 2 * cp;
            v2f y0[8], y1[8];
            static_for<0, 8>([&](auto B) {
                const v4f v = *reinterpret_cast<const v4f *>(cell + 18 * B.value);
                y0[bitrev<8>(B.value)] = v2f{v.x, v.y}; y1[bitrev<8>(B.value)] = v2f{v.z, v.w};
            });
            dft_dit<8, +1>(y0);
            dft_dit<8, +1>(y1);
            v2f p0[8], p1[8];
            twiddle_powers<8>(opaque(wB0), p0);
            twiddle_powers<8>(opaque(wB1), p1);
            static_for<0, 8>([&](auto K) {
                constexpr int k = K.value;
                if constexpr (k != 0) { y0[k] = cmul(y0[k], p0[k]); y1[k] = cmul(y1[k], p1[k]); }
                *reinterpret_cast<v4f *>(cell + 18 * k) = v4f{y0[k].x, y0[k].y, y1[k].x, y1[k].y};
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- pass C: the 16 consecutive points of row t
        {
            const v2f *row = lds + 18 * t;
            static_for<0, 8>([&](auto J) {
                const v4f v = *reinterpret_cast<const v4f *>(row + 2 * J.value);
                x[bitrev<16>(2 * J.value)] = v2f{v.x, v.y};
                x[bitrev<16>(2 * J.value + 1)] = v2f{v.z, v.w};
            });
            dft_dit<16, +1>(x);
        }
        const float inv_prev = 1.0f / prev_count;
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const float p = x[r].x * x[r].x + x[r].y * x[r].y;
            if (total <= a.ave_size) sm[r] = sm[r] + p;
            else sm[r] = sm[r] - sm[r] * inv_prev + p;            // minus the previous mean (fft.cpp:570-574)
        });
    }
    if (a.nparts > 1) {
        float *dst = a.part + ((long)ch * a.nparts + part) * N;
        static_for<0, 16>([&](auto Rr) { dst[((kbin + 128 * Rr.value) + N / 2) & (N - 1)] = sm[Rr.value]; });
    } else if (a.nframes > 0) {
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const int j = ((kbin + 128 * r) + N / 2) & (N - 1);
            const float m = sm[r] / (float)ave_count;
            sum[j] = sm[r]; pwr[j] = m;
            ave[j] = (float)((double)log10f(m + a.kc) + a.kb);
        });
    }
    if (t == 0 && a.nparts == 1) { a.counters[2 * ch] = ave_count; a.counters[2 * ch + 1] = total; }
    if (over) a.overload[ch] = 1;
}

// ---- the 8192-point display spectrum with sixteen points per thread (round 4): 512 threads, N = 16 x 32 x 16 -------------
// spectrum_kernel<13> is 256 threads x 32 points at 300 registers (2.7 TB/s).  The middle pass has 32 points per column,
// two threads' worth: the column (ka, c) is split by the PARITY of b over a pair of neighbouring lanes (h = t & 1) --
// each does a 16-point DIT over its b = 2 j + h, the odd one applies W_32^{kj}, and the radix-2 butterfly that joins the
// halves takes the partner's sixteen values over the DPP quad permute (1,0,3,2): no LDS, no barrier.
//   n = 512 a + 16 b + c     k = ka + 16 kb + 512 kc     kb = kj + 16 h        LDS cell of (ka, x, c): 18 (32 ka + x) + c
//   A  thread t           : samples 512 a + t, DFT over a, twiddle W_N^{t ka}
//   B  thread (ka, c, h)  : DFT over b as above, twiddle W_512^{c kb}, back to rows kb of the same columns
//   C  thread t = 32 ka + kb : its 16 consecutive points, DFT over c                   -> bin ka + 16 kb + 512 kc
// B -> C inside the half-wave that owns ka.  One workgroup per CU would fit twice (74 KB of LDS); registers allow one.
constexpr int SPEC32_LDS = (8192 + 2 * 512) * 8;
__device__ __forceinline__ v2f pair_swap(v2f v)           // the value of lane t ^ 1
{
    return v2f{__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.x), 0xB1, 0xf, 0xf, false)),
               __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.y), 0xB1, 0xf, 0xf, false))};
}
__global__ __launch_bounds__(512)
void spectrum32_kernel(SpectrumArgs a)
{
    constexpr int N = 8192, T = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x, ch = blockIdx.x / a.nparts, part = blockIdx.x % a.nparts;
    const int f0 = (int)((long)a.nframes * part / a.nparts), f1 = (int)((long)a.nframes * (part + 1) / a.nparts);
    const v2f *tw1 = reinterpret_cast<const v2f *>(a.tw1);           // W_N^n, n < 1024
    const int hB = t & 1, cB = (t >> 1) & 15, kaB = t >> 5;          // pass B: (ka, c, h)
    const v2f wA = tw1[t];                                            // W_N^t
    const v2f wB = tw1[16 * cB];                                      // W_512^c = W_N^{16 c}
    v2f wH;                                                           // W_512^{16 c h} = W_32^{c h} = W_N^{256 c h}, c h < 16:
    {                                                                 // the table holds an OCTANT of W_N at N = 8192 (W_N^1024 = e^{j pi/4})
        const int m = 256 * cB * hB;                                  // < 4096: up to three eighth turns beyond the table entry
        const v2f v = tw1[m & 1023];
        const int o = (m >> 10) & 3;
        constexpr float r = 0.70710678118654752440f;
        if (o == 0) wH = v;
        else if (o == 1) wH = v2f{r * (v.x - v.y), r * (v.x + v.y)};
        else if (o == 2) wH = v2f{-v.y, v.x};
        else wH = v2f{-r * (v.x + v.y), r * (v.x - v.y)};
    }
    const v2f *in = reinterpret_cast<const v2f *>(a.in) + (long)ch * a.in_stride;
    float *sum = a.sum + (long)ch * N, *pwr = a.pwr + (long)ch * N, *ave = a.ave + (long)ch * N;
    int ave_count = a.counters[2 * ch], total = a.counters[2 * ch + 1];
    total += f0;                                              // counters at this group's first frame
    ave_count = ave_count + f0 < a.ave_size ? ave_count + f0 : (ave_count > a.ave_size ? ave_count : a.ave_size);
    int over = 0;
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int kbin = (tt >> 5) + 16 * (tt & 31);              // pass C row t = 32 ka + kb: bins kbin + 512 kc
    float sm[16], wn[16];
    static_for<0, 16>([&](auto Rr) {
        constexpr int r = Rr.value;
        sm[r] = a.nparts == 1 ? sum[((kbin + 512 * r) + N / 2) & (N - 1)] : 0.f;   // display order, fft.cpp:564-589
        wn[r] = a.win[512 * r + t];
    });
    v2f nxt[16];
    auto fetch = [&](int f) {
        const v2f *src = in + (long)f * N + t;
#pragma unroll
        for (int q = 0; q < 16; q++) nxt[q] = src[512 * q];
    };
    if (f0 < f1) fetch(f0);
    for (int f = f0; f < f1; f++) {
        v2f x[16];
        static_for<0, 16>([&](auto Q) {
            constexpr int q = Q.value;
            const v2f s_ = nxt[q];
            if (s_.x > refc::FFT_OVER_LIMIT_F) over = 1;                        // OVER_LIMIT, fft.cpp:30,275
            x[bitrev<16>(q)] = v2f{wn[q] * s_.y, wn[q] * s_.x};    // I/Q swapped, fft.cpp:280-281
        });
        if (f + 1 < f1) fetch(f + 1);
        const float prev_count = (float)ave_count;
        total++;                                                  // CpxFFT counters, fft.cpp:515-517
        if (ave_count < a.ave_size) ave_count++;
        // ---- pass A
        dft_dit<16, +1>(x);
        {
            v2f pw[16];
            twiddle_powers<16>(opaque(wA), pw);
            static_for<1, 16>([&](auto K) { x[K.value] = cmul(x[K.value], pw[K.value]); });
        }
        __syncthreads();                       // the previous frame's pass C has read its rows
        static_for<0, 16>([&](auto K) { lds[pad16(512 * K.value + t)] = x[K.value]; });
        __syncthreads();
        // ---- pass B: column (ka, c), the half b = 2 j + h
        {
            v2f *col = lds + 18 * (32 * kaB + hB) + cB;                          // point b = 2 j + h at col[36 j]
            static_for<0, 16>([&](auto J) { x[bitrev<16>(J.value)] = lds_ld8(col + 36 * J.value); });
            dft_dit<16, +1>(x);                                                   // Y_h[kj]
            // the odd half times W_32^{kj} (constants), then the butterfly with the partner lane's half
            static_for<0, 16>([&](auto K) {
                constexpr int k = K.value;
                const v2f w32 = v2f{(float)__builtin_cos(6.283185307179586476925286766559 * k / 32.0),
                                    (float)__builtin_sin(6.283185307179586476925286766559 * k / 32.0)};
                const v2f mine = hB ? cmul(x[k], w32) : x[k];
                const v2f other = pair_swap(mine);
                x[k] = hB ? other - mine : mine + other;          // h = 0: Y0 + w Y1 -> kb = kj;  h = 1: Y0 - w Y1 -> kb = kj + 16
            });
            v2f pw[16];
            twiddle_powers<16>(opaque(wB), pw);                   // W_512^{c kj}
            static_for<0, 16>([&](auto K) {
                constexpr int k = K.value;
                v2f v = x[k];
                if constexpr (k != 0) v = cmul(v, pw[k]);
                v = hB ? cmul(v, wH) : v;                         // times W_512^{16 c h}
                lds_st8(lds + 18 * (32 * kaB + k + 16 * hB) + cB, v);
            });
        }
        // B -> C stays inside the thirty-two threads that own ka
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- pass C: the 16 consecutive points of row t
        {
            const v2f *row = lds + 18 * t;
            static_for<0, 8>([&](auto J) {
                const v4f v = *reinterpret_cast<const v4f *>(row + 2 * J.value);
                x[bitrev<16>(2 * J.value)] = v2f{v.x, v.y};
                x[bitrev<16>(2 * J.value + 1)] = v2f{v.z, v.w};
            });
            dft_dit<16, +1>(x);
        }
        const float inv_prev = 1.0f / prev_count;
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const float p = x[r].x * x[r].x + x[r].y * x[r].y;
            if (total <= a.ave_size) sm[r] = sm[r] + p;
            else sm[r] = sm[r] - sm[r] * inv_prev + p;            // minus the previous mean (fft.cpp:570-574)
        });
    }
    if (a.nparts > 1) {
        float *dst = a.part + ((long)ch * a.nparts + part) * N;
        static_for<0, 16>([&](auto Rr) { dst[((kbin + 512 * Rr.value) + N / 2) & (N - 1)] = sm[Rr.value]; });
    } else if (a.nframes > 0) {
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value;
            const int j = ((kbin + 512 * r) + N / 2) & (N - 1);
            const float m = sm[r] / (float)ave_count;
            sum[j] = sm[r]; pwr[j] = m;
            ave[j] = (float)((double)log10f(m + a.kc) + a.kb);
        });
    }
    if (t == 0 && a.nparts == 1) { a.counters[2 * ch] = ave_count; a.counters[2 * ch + 1] = total; }
    if (over) a.overload[ch] = 1;
}

// plain transform: out[k] = sum_n in[n] e^{sign j 2 pi n k / N}; sign=-1 via conjugation
template <int LOG2N>
__global__ __launch_bounds__(SpecCfg<LOG2N>::T)
void fft_plain_kernel(const v2f *in, v2f *out, const v2f *tw1g, const v2f *tw2g, int sign)
{
    using Cfg = SpecCfg<LOG2N>;
    constexpr int T = Cfg::T, R0 = Cfg::R0, G = Cfg::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    v2f *tw2 = lds + Cfg::LDS_DATA;
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += T) tw2[i] = tw2g[i];
    v2f w1[G];
#pragma unroll
    for (int e = 0; e < G; e++) w1[e] = tw1g[Cfg::col(t, e)];
    const float cj = sign > 0 ? 1.0f : -1.0f;
    v2f x[32];
#pragma unroll
    for (int e = 0; e < G; e++)
#pragma unroll
        for (int n1 = 0; n1 < R0; n1++) {
            const v2f s = in[1024 * n1 + Cfg::col(t, e)];
            x[e * R0 + n1] = v2f{s.x, cj * s.y};
        }
    __syncthreads();
    fft_fwd_passes<LOG2N>(x, lds, tw2, w1);
    const int k0 = t >> 5, k1 = t & 31;
    __syncthreads();                       // in == out allowed: every input was read before pass 1
    static_for<0, 32>([&](auto Rr) {
        constexpr int r = Rr.value, k2 = r;
        out[k0 + R0 * (k1 + 32 * k2)] = v2f{x[r].x, cj * x[r].y};
    });
}

// The weight alpha_g = product of the per-frame factors (1 - 1/count once the average is full) of every frame group,
// and the average count after the call: the same for all bins of a channel (the combine kernel used to do this walk in
// each of its channels x N threads: 0.18 of K3's 1.6 ms).  One thread per (channel, frame group): the counters at a
// group's first frame follow from those at the call's start in closed form, so the groups' walks -- the same float
// operations in the same order as one walk over all frames -- run side by side (round 4: one thread per channel took
// 40 us of the C1 call's 940 for its 512 sequential divisions).
// One workgroup per channel (thread g = frame group g, in rounds of the block size): once every thread has read the
// counters, thread 0 moves them -- what spectrum_count_kernel did in a launch of its own.
__global__ void spectrum_alpha_kernel(SpectrumArgs a)
{
    const int ch = blockIdx.x;
    const int ave0 = a.counters[2 * ch], total0 = a.counters[2 * ch + 1];
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ave_count = ave0 + a.nframes;
        a.counters[2 * ch] = ave_count < a.ave_size ? ave_count : (ave0 > a.ave_size ? ave0 : a.ave_size);
        a.counters[2 * ch + 1] = total0 + a.nframes;
    }
    for (int g = threadIdx.x; g < a.nparts; g += blockDim.x) {
        // after k frames: total0 + k, and the average count saturates at ave_size (a count above it -- the average was
        // shortened -- stays)
        auto ave_after = [&](int k) { return ave0 < a.ave_size ? (ave0 + k < a.ave_size ? ave0 + k : a.ave_size) : ave0; };
        const int f0 = (int)((long)a.nframes * g / a.nparts), f1 = (int)((long)a.nframes * (g + 1) / a.nparts);
        int ave_count = ave_after(f0), total = total0 + f0;
        float al = 1.f;                                                // product of the group's alpha_f
        for (int f = f0; f < f1; f++) {
            const float prev = (float)ave_count;
            total++;
            if (ave_count < a.ave_size) ave_count++;
            if (total > a.ave_size) al = al - al / prev;
            if (al == 0.f) break;                                      // (no averaging: the first frame already forgets everything)
        }
        a.alpha[(long)ch * a.nparts + g] = al;
        if (g == 0) a.alpha[(long)a.channels * a.nparts + ch] = (float)ave_after(a.nframes);
    }
}
// folds the frame groups of spectrum_kernel (nparts > 1) into the running sum, writes mean and bels
__global__ void spectrum_combine_kernel(SpectrumArgs a, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (j >= n) return;
    float sm = a.sum[(long)ch * n + j];
    for (int g = 0; g < a.nparts; g++)
        sm = a.alpha[(long)ch * a.nparts + g] * sm + a.part[((long)ch * a.nparts + g) * n + j];
    const float m = sm / a.alpha[(long)a.channels * a.nparts + ch];
    a.sum[(long)ch * n + j] = sm; a.pwr[(long)ch * n + j] = m;
    a.ave[(long)ch * n + j] = (float)((double)log10f(m + a.kc) + a.kb);
}
template <int LOG2N>
static hipError_t spec_launch_one(const SpectrumArgs &a, hipStream_t s)
{
    using Cfg = SpecCfg<LOG2N>;
    hipError_t e = CSDR_MAX_LDS_ONCE((&spectrum_kernel<LOG2N>), Cfg::LDS_BYTES);
    if (e != hipSuccess) return e;
    static const bool wide = !(getenv("CSDR_SPEC16") && atoi(getenv("CSDR_SPEC16")) == 0);
    if (LOG2N == 12 && wide)
        hipLaunchKernelGGL(spectrum16_kernel, dim3(a.channels * a.nparts), dim3(256), SPEC16_LDS, s, a);
    else if (LOG2N == 11 && wide)
        hipLaunchKernelGGL(spectrum8_kernel, dim3(a.channels * a.nparts), dim3(128), SPEC8_LDS, s, a);
    else if (LOG2N == 13 && wide) {
        e = CSDR_MAX_LDS_ONCE((&spectrum32_kernel), SPEC32_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(spectrum32_kernel, dim3(a.channels * a.nparts), dim3(512), SPEC32_LDS, s, a);
    } else
        hipLaunchKernelGGL(spectrum_kernel<LOG2N>, dim3(a.channels * a.nparts), dim3(Cfg::T), Cfg::LDS_BYTES, s, a);
    if (a.nparts > 1) {
        hipLaunchKernelGGL(spectrum_alpha_kernel, dim3(a.channels), dim3(64), 0, s, a);
        hipLaunchKernelGGL(spectrum_combine_kernel, dim3(Cfg::N / 256, a.channels), dim3(256), 0, s, a, (int)Cfg::N);
    }
    return hipGetLastError();
}
template <int LOG2N>
static hipError_t plain_launch_one(int sign, const float *in, float *out, const float *tw1, const float *tw2, hipStream_t s)
{
    using Cfg = SpecCfg<LOG2N>;
    hipError_t e = CSDR_MAX_LDS_ONCE((&fft_plain_kernel<LOG2N>), Cfg::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fft_plain_kernel<LOG2N>, dim3(1), dim3(Cfg::T), Cfg::LDS_BYTES, s, (const v2f *)in, (v2f *)out,
                       (const v2f *)tw1, (const v2f *)tw2, sign);
    return hipGetLastError();
}

hipError_t spectrum_launch(int log2n, const SpectrumArgs &a, hipStream_t stream)
{
    switch (log2n) {
    case 11: return spec_launch_one<11>(a, stream);
    case 12: return spec_launch_one<12>(a, stream);
    case 13: return spec_launch_one<13>(a, stream);
    case 14: return spec_launch_one<14>(a, stream);
    default: return hipErrorInvalidValue;
    }
}
hipError_t fft_plain_launch(int log2n, int sign, const float *in, float *out, const float *tw1,
                            const float *tw2, hipStream_t stream)
{
    switch (log2n) {
    case 11: return plain_launch_one<11>(sign, in, out, tw1, tw2, stream);
    case 12: return plain_launch_one<12>(sign, in, out, tw1, tw2, stream);
    case 13: return plain_launch_one<13>(sign, in, out, tw1, tw2, stream);
    case 14: return plain_launch_one<14>(sign, in, out, tw1, tw2, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace csdr
