// fastfir16_kernels.hip -- CFastFIR::ProcessData (reference dsp/fastfir.cpp:268-306) for N = 2048, the reference's own
// filter size (dsp/fastfir.h: CONV_FFT_SIZE) and the one every receiver of the chain runs, as 128 threads x 16 points.
//
// The generic kernel (fastfir_kernels.hip) holds 32 points per thread: at N = 2048 that is ONE wave per workgroup at
// 272 registers -- four waves per CU, vector unit 17 %, LDS 15 % busy, everything else latency (tools/pmc_lds_per_kernel.sh).
// Sixteen points per thread fit three times the waves.  N = 16 x 8 x 16:
//   n = 128 a + 16 b + c   (a < 16, b < 8, c < 16)        k = ka + 16 kb + 128 kc
//   F1  thread t = 16 b + c : DIF over a (the block's rows: a < 8 the old half, a >= 8 the new one), twiddle W_N^{t ka}
//   F2  thread (ka, c pair) : DIF over b, twiddle W_128^{c kb}                       in place: (ka, b, c) -> (ka, kb, c)
//   F3  thread (ka, kb)     : DIF over c, multiply by H, DIT back over kc, conjugate twiddle   -- registers only
//   I2  thread (ka, c pair) : DIT back over kb
//   I1  thread t            : conjugate twiddle, DIT back over ka, store the rows a >= 8 (the valid half)
// LDS cell of (ka, x, c): 18 (8 ka + x) + c -- rows of 16 points, 18 apart (16-byte row reads without bank conflicts).
// F2 -> F3 -> I2 stay inside the wave that owns ka (eight per wave): two workgroup barriers per block, as in the
// generic kernel.  H comes in this kernel's own order (fastfir16_bin_of): slot kc * 128 + t3 for thread t3 = 8 ka + kb.
#include "fastfir_dev.hpp"
#include "fastfir_kernels.h"

namespace csdr {

namespace {
constexpr int F16_N = 2048, F16_T = 128, F16_L = 1024;
constexpr int F16_LDS_DATA = 18 * 128;                   // 128 rows of 16 (+2)
constexpr int F16_LDS_BYTES = (F16_LDS_DATA + 256) * 8;  // + the 8 x 16 twiddles of F2 / I2, [kb][c] and [c][kb]
__device__ __forceinline__ int row18(int row, int c) { return 18 * row + c; }
}  // namespace

__global__ __launch_bounds__(F16_T)
void fastfir16_kernel(FastFirArgs a)
{
    constexpr int N = F16_N, T = F16_T, L = F16_L;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    v2f *twB = lds + F16_LDS_DATA;                        // twB[16 kb + c] = W_128^{c kb}
    const int t = threadIdx.x;

    int wg = blockIdx.x, ch, run;
    if ((a.channels & 7) == 0) {                          // the runs of a channel (one H) on one XCD's L2
        const int xcd = wg & 7, slot = wg >> 3;
        ch = (slot / a.runs) * 8 + xcd;
        run = slot % a.runs;
    } else {
        ch = wg / a.runs;
        run = wg % a.runs;
    }
    const int b0 = run * a.blocks_per_run;
    int b1 = b0 + a.blocks_per_run;
    if (b1 > a.nblocks) b1 = a.nblocks;
    if (ch >= a.channels || b0 >= b1) return;             // uniform per workgroup

    const v2f *tw1 = reinterpret_cast<const v2f *>(a.tw1);    // W_N^n, n < 1024
    {
        const int m = 16 * (t & 15) * (t >> 4);           // W_128^{c kb} = W_N^{16 c kb}; W_N^1024 = -1
        v2f v = tw1[m & 1023];
        if (m & 1024) v = -v;
        twB[t] = v;
        // ... and transposed for I3, whose lanes differ in kb: [kb][c] there is a stride of 32 banks (PMC: a quarter of
        // this kernel's LDS cycles were bank conflicts), [c][kb] is eight neighbouring cells
        twB[128 + 8 * (t & 15) + (t >> 4)] = v;
    }
    v2f pw[16];                                           // W_N^{t ka}: resident
    twiddle_powers<16>(tw1[t], pw);

    const v2f *in = reinterpret_cast<const v2f *>(a.in) + (long)ch * a.in_stride;
    v2f *out = reinterpret_cast<v2f *>(a.out) + (long)ch * a.out_stride;
    const v2f *hist = reinterpret_cast<const v2f *>(a.hist) + (long)ch * L;
    const v2f *H = reinterpret_cast<const v2f *>(a.h) + (long)ch * a.h_stride * 2;   // h_stride counts float4

    v2f carry[8], nxt[8];                                 // rows a of the old / the new half: sample 128 a + t
    {
        const v2f *src = b0 == 0 ? hist : in + (long)(b0 - 1) * L;
#pragma unroll
        for (int q = 0; q < 8; q++) carry[q] = src[128 * q + t];
#pragma unroll
        for (int q = 0; q < 8; q++) nxt[q] = in[(long)b0 * L + 128 * q + t];
    }
    const int ka2 = t >> 3, cp = t & 7;                   // F2 / I2: (ka, columns 2 cp and 2 cp + 1)
    const int kb3 = t & 7;                                // F3: thread t = 8 ka + kb owns row t
    __syncthreads();                                      // twB

    for (int b = b0; b < b1; b++) {
        v2f x[16];
#pragma unroll
        for (int q = 0; q < 8; q++) { x[q] = carry[q]; x[8 + q] = nxt[q]; carry[q] = nxt[q]; }
        if (b + 1 < b1) {
#pragma unroll
            for (int q = 0; q < 8; q++) nxt[q] = in[(long)(b + 1) * L + 128 * q + t];
        }
        // ---------------- F1: DIF over a, twiddle, column t of every ka ----------------
        dft_dif<16, +1>(x);
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value, ka = bitrev<16>(r);
            if constexpr (ka != 0) x[r] = cmul(x[r], pw[ka]);
            // (no barrier in front: these are the cells this thread itself read in I1 of the previous block)
            lds[row18(8 * ka + (t >> 4), t & 15)] = x[r];
        });
        __syncthreads();
        // ---------------- F2: DIF over b for the column pair, twiddle, in place ----------------
        {
            v2f y0[8], y1[8];
            v2f *cell = lds + row18(8 * ka2, 2 * cp);     // (ka, b, 2 cp) at cell + 18 b
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const v4f v = *reinterpret_cast<const v4f *>(cell + 18 * q);
                y0[q] = v2f{v.x, v.y}; y1[q] = v2f{v.z, v.w};
            }
            dft_dif<8, +1>(y0);
            dft_dif<8, +1>(y1);
            static_for<0, 8>([&](auto Rr) {
                constexpr int r = Rr.value, kb = bitrev<8>(r);
                if constexpr (kb != 0) {
                    const v4f w = *reinterpret_cast<const v4f *>(twB + 16 * kb + 2 * cp);
                    y0[r] = cmul(y0[r], v2f{w.x, w.y});
                    y1[r] = cmul(y1[r], v2f{w.z, w.w});
                }
                *reinterpret_cast<v4f *>(cell + 18 * kb) = v4f{y0[r].x, y0[r].y, y1[r].x, y1[r].y};
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---------------- F3 + H + I3: row t, registers only ----------------
        {
            v2f hv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) hv[q] = H[128 * q + t];      // slot kc * 128 + t (L2)
            v2f *row = lds + 18 * t;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const v4f v = *reinterpret_cast<const v4f *>(row + 2 * q);
                x[2 * q] = v2f{v.x, v.y}; x[2 * q + 1] = v2f{v.z, v.w};
            }
            dft_dif<16, +1>(x);
            static_for<0, 16>([&](auto Rr) { x[Rr.value] = cmul(x[Rr.value], hv[bitrev<16>(Rr.value)]); });
            dft_dit<16, -1>(x);
            static_for<0, 16>([&](auto Cc) {
                constexpr int c = Cc.value;
                if constexpr (c != 0) x[c] = cmul_conj(x[c], twB[128 + 8 * c + kb3]);
            });
#pragma unroll
            for (int q = 0; q < 8; q++)                   // two cells per store, like the loads: no two rows of a pass in one bank
                *reinterpret_cast<v4f *>(row + 2 * q) = v4f{x[2 * q].x, x[2 * q].y, x[2 * q + 1].x, x[2 * q + 1].y};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---------------- I2: DIT back over kb for the column pair, in place ----------------
        {
            v2f y0[8], y1[8];
            v2f *cell = lds + row18(8 * ka2, 2 * cp);
            static_for<0, 8>([&](auto Rr) {
                constexpr int r = Rr.value, kb = bitrev<8>(r);
                const v4f v = *reinterpret_cast<const v4f *>(cell + 18 * kb);
                y0[r] = v2f{v.x, v.y}; y1[r] = v2f{v.z, v.w};
            });
            dft_dit<8, -1>(y0);
            dft_dit<8, -1>(y1);
#pragma unroll
            for (int q = 0; q < 8; q++) *reinterpret_cast<v4f *>(cell + 18 * q) = v4f{y0[q].x, y0[q].y, y1[q].x, y1[q].y};
        }
        __syncthreads();
        // ---------------- I1: conjugate twiddle, DIT back over ka, the valid half out ----------------
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value, ka = bitrev<16>(r);
            x[r] = lds[row18(8 * ka + (t >> 4), t & 15)];
            if constexpr (ka != 0) x[r] = cmul_conj(x[r], pw[ka]);
        });
        dft_dit<16, -1>(x);
#pragma unroll
        for (int q = 0; q < 8; q++) out[(long)b * L + 128 * q + t] = x[8 + q];
    }
    // the tail of this call's input is the overlap of the next call (fastfir.cpp:280-300), in the other half of the
    // ping-pong history
    if (b1 == a.nblocks) {
        v2f *hn = reinterpret_cast<v2f *>(a.hist_next) + (long)ch * L;
#pragma unroll
        for (int q = 0; q < 8; q++) hn[128 * q + t] = carry[q];
    }
}

hipError_t fastfir16_launch(const FastFirArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(fastfir16_kernel, dim3(a.channels * a.runs), dim3(F16_T), F16_LDS_BYTES, stream, a);
    return hipGetLastError();
}

// natural-order spectrum bin of H slot i (= kc * 128 + t3, thread t3 = 8 ka + kb): k = ka + 16 kb + 128 kc
int fastfir16_bin_of(int slot)
{
    const int kc = slot >> 7, t3 = slot & 127;
    return (t3 >> 3) + 16 * (t3 & 7) + 128 * kc;
}

// (N = 4096 had a kernel of this shape here in round 4 -- 256 threads x 16 points, 0.378 of the roofline; since round 5 the
// pipelined build of fastfir2_kernels.hip takes that size, two blocks per workgroup: 13 % faster.)

}  // namespace csdr
