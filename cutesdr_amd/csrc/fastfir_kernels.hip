// fastfir_kernels.hip -- batched overlap-save FFT FIR for gfx950 (K1 in DESIGN.md).
//
// Replaces the per-block work of CFastFIR::ProcessData (reference dsp/fastfir.cpp:268-306:
// FwdFFT -> CpxMpy (:312-321) -> RevFFT -> keep samples P-1..N-1) for many channels and many
// blocks per launch.  One workgroup of N/32 threads owns one N-point block at a time:
//
//   pass F1  radix-(N/1024) DIF on samples read straight from HBM (16 B/lane coalesced)
//   pass F2  radix-32 DIF inside each 1024-point sub-transform          (LDS exchange)
//   pass F3  radix-32 DIF, multiply by H[k], radix-32 DIT inverse        (registers only)
//   pass I2  radix-32 DIT inverse                                        (LDS exchange)
//   pass I3  radix-(N/1024) DIT inverse, store the valid upper half to HBM
//
// The forward transform is decimation-in-frequency (digit-reversed spectrum), the inverse is
// decimation-in-time, so no reordering pass exists; H[k] is stored by the host in the register
// order of pass F3 (see csdr_fastfir_hperm()).  Two of the four LDS exchanges stay inside a
// half-wave (a 1024-point sub-transform lives in 32 lanes) and need no workgroup barrier.
// A workgroup walks consecutive blocks of one channel and carries the overlapping half of the
// input in registers, so every input sample is read from HBM once.
//
// HBM roofline accounting: 8 B read + 8 B written per output sample.
#include "fft_core.hpp"
#include "fastfir_kernels.h"

namespace csdr {

template <int LOG2N>
struct FastFirCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 32;          // threads per workgroup
    static constexpr int R0 = N / 1024;       // radix of the outer pass
    static constexpr int G = 32 / R0;         // adjacent columns handled per thread in F1/I3
    static constexpr int LDS_DATA = N + 2 * (N / 32);               // padded v2f elements
    static constexpr int LDS_BYTES = (LDS_DATA + 1024) * 8;         // + 32x32 twiddle table
};

// DBG=true builds a diagnostic twin that copies the (unpadded) LDS image to a.dbg after pass
// a.dbg_stage (1 = F1, 2 = F2, 3 = F3+H+I1, 4 = I2) of its first block and exits.
template <int LOG2N, bool DBG>
__device__ __forceinline__ void dbg_dump(const FastFirArgs &a, const v2f *lds, int stage)
{
    if constexpr (DBG) {
        if (a.dbg_stage == stage) {
            __syncthreads();
            for (int i = threadIdx.x; i < (1 << LOG2N); i += blockDim.x) a.dbg[i] = lds[lds_pad(i)];
        }
    }
}

template <int LOG2N, bool DBG = false>
__global__ __launch_bounds__(FastFirCfg<LOG2N>::T)
void fastfir_os_kernel(FastFirArgs a)
{
    using Cfg = FastFirCfg<LOG2N>;
    constexpr int N = Cfg::N, T = Cfg::T, R0 = Cfg::R0, G = Cfg::G, L = N / 2;
    constexpr int HALF = R0 / 2;              // rows of the old / new half in F1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    v2f *tw2 = lds + Cfg::LDS_DATA;           // tw2[k1*32 + n2] = W_1024^{n2*k1}

    const int t = threadIdx.x;

    // ---- which (channel, run of blocks) this workgroup owns; XCD-aware so that the runs of a
    //      channel (which share H) sit on one XCD's L2 when the channel count allows it
    int wg = blockIdx.x, ch, run;
    if ((a.channels & 7) == 0) {
        int xcd = wg & 7, slot = wg >> 3;
        ch = (slot / a.runs) * 8 + xcd;
        run = slot % a.runs;
    } else {
        ch = wg / a.runs;
        run = wg % a.runs;
    }
    const int b0 = run * a.blocks_per_run;
    int b1 = b0 + a.blocks_per_run;
    if (b1 > a.nblocks) b1 = a.nblocks;
    if (ch >= a.channels || b0 >= b1) return;          // uniform per workgroup

    for (int i = t; i < 1024; i += T) tw2[i] = a.tw2[i];

    const v2f *in = a.in + (long)ch * a.in_stride;
    const v2f *hist = a.hist + (long)ch * L;
    v2f *out = a.out + (long)ch * a.out_stride;
    const v4f *H = a.h + (long)ch * a.h_stride;

    // outer-pass base twiddles W_N^{n2}, n2 = G*t+e: constant over the block loop
    v2f w1[G];
#pragma unroll
    for (int e = 0; e < G; e++) w1[e] = a.tw1[G * t + e];

    // x[e*R0 + n1] <-> sample 1024*n1 + G*t + e of the block
    v2f x[32];
    v2f carry[16];       // the half that becomes the old half of the next block

    // old half of the first block: previous call's tail (b0 == 0) or the input itself
    {
        const v2f *src = (b0 == 0) ? hist : (in + (long)(b0 - 1) * L);
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int e = 0; e < G; e++) carry[e * HALF + n1] = src[1024 * n1 + G * t + e];
    }

    const int sb = t >> 5, sn = t & 31;       // sub-transform and column of passes F2 / I2

    for (int b = b0; b < b1; b++) {
        // ---------------- F1: load, radix-R0 DIF, twiddle, scatter to LDS ----------------
        {
            const v2f *src = in + (long)b * L;
#pragma unroll
            for (int e = 0; e < G; e++)
#pragma unroll
                for (int n1 = 0; n1 < HALF; n1++) {
                    x[e * R0 + n1] = carry[e * HALF + n1];
                    v2f nv = src[1024 * n1 + G * t + e];
                    x[e * R0 + HALF + n1] = nv;
                    carry[e * HALF + n1] = nv;
                }
        }
#pragma unroll
        for (int e = 0; e < G; e++) {
            v2f y[R0];
#pragma unroll
            for (int i = 0; i < R0; i++) y[i] = x[e * R0 + i];
            dft_dif<R0, +1>(y);
            v2f pw[R0];
            twiddle_powers<R0>(opaque(w1[e]), pw);     // recomputed per block: no registers to keep it
            static_for<0, R0>([&](auto Rr) {
                constexpr int r = Rr.value, k0 = bitrev<R0>(r);
                if constexpr (k0 != 0) y[r] = cmul(y[r], pw[k0]);
            });
#pragma unroll
            for (int i = 0; i < R0; i++) x[e * R0 + i] = y[i];
        }
        __syncthreads();                       // previous block's I3 reads are done
        static_for<0, R0>([&](auto Rr) {
            constexpr int r = Rr.value, k0 = bitrev<R0>(r);
            const int base = lds_pad(1024 * k0 + G * t);
            if constexpr (G % 2 == 0) {
#pragma unroll
                for (int e = 0; e < G; e += 2) {
                    v4f v = {x[e * R0 + r].x, x[e * R0 + r].y, x[(e + 1) * R0 + r].x, x[(e + 1) * R0 + r].y};
                    *reinterpret_cast<v4f *>(&lds[base + e]) = v;
                }
            } else {
#pragma unroll
                for (int e = 0; e < G; e++) lds[base + e] = x[e * R0 + r];
            }
        });
        __syncthreads();
        dbg_dump<LOG2N, DBG>(a, lds, 1);
        if (DBG && a.dbg_stage == 1) return;

        // ---------------- F2: radix-32 DIF inside sub-transform sb, column sn -------------
        {
            const int base = lds_pad(1024 * sb) + sn;
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) x[n1] = lds[base + 34 * n1];
            dft_dif<32, +1>(x);
            static_for<1, 32>([&](auto Rr) {
                constexpr int r = Rr.value, k1 = bitrev<32>(r);
                x[r] = cmul(x[r], tw2[k1 * 32 + sn]);
            });
            static_for<0, 32>([&](auto Rr) {
                constexpr int r = Rr.value, k1 = bitrev<32>(r);
                lds[base + 34 * k1] = x[r];
            });
        }
        // F2 -> F3 stays inside the half-wave that owns sub-transform sb
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dbg_dump<LOG2N, DBG>(a, lds, 2);
        if (DBG && a.dbg_stage == 2) return;

        // ---------------- F3 + H + I1: rows 32t..32t+31, registers only -------------------
        {
            const v4f *row = reinterpret_cast<const v4f *>(&lds[34 * t]);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                v4f v = row[j];
                x[2 * j] = v2f{v.x, v.y};
                x[2 * j + 1] = v2f{v.z, v.w};
            }
            dft_dif<32, +1>(x);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                v4f hv = H[j * T + t];
                x[2 * j] = cmul(x[2 * j], v2f{hv.x, hv.y});
                x[2 * j + 1] = cmul(x[2 * j + 1], v2f{hv.z, hv.w});
            }
            dft_dit<32, -1>(x);
            v4f *wrow = reinterpret_cast<v4f *>(&lds[34 * t]);
#pragma unroll
            for (int j = 0; j < 16; j++)
                wrow[j] = v4f{x[2 * j].x, x[2 * j].y, x[2 * j + 1].x, x[2 * j + 1].y};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        dbg_dump<LOG2N, DBG>(a, lds, 3);
        if (DBG && a.dbg_stage == 3) return;

        // ---------------- I2: conj twiddle, radix-32 DIT inverse ---------------------------
        {
            const int base = lds_pad(1024 * sb) + sn;
            static_for<0, 32>([&](auto Rr) {
                constexpr int r = Rr.value, k1 = bitrev<32>(r);
                x[r] = lds[base + 34 * k1];
                if constexpr (k1 != 0) x[r] = cmul_conj(x[r], tw2[k1 * 32 + sn]);
            });
            dft_dit<32, -1>(x);
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) lds[base + 34 * n1] = x[n1];
        }
        __syncthreads();
        dbg_dump<LOG2N, DBG>(a, lds, 4);
        if (DBG && a.dbg_stage == 4) return;

        // ---------------- I3: conj twiddle, radix-R0 DIT inverse, store valid half ---------
        static_for<0, R0>([&](auto Rr) {
            constexpr int r = Rr.value, k0 = bitrev<R0>(r);
            const int base = lds_pad(1024 * k0 + G * t);
            if constexpr (G % 2 == 0) {
#pragma unroll
                for (int e = 0; e < G; e += 2) {
                    v4f v = *reinterpret_cast<const v4f *>(&lds[base + e]);
                    x[e * R0 + r] = v2f{v.x, v.y};
                    x[(e + 1) * R0 + r] = v2f{v.z, v.w};
                }
            } else {
#pragma unroll
                for (int e = 0; e < G; e++) x[e * R0 + r] = lds[base + e];
            }
        });
        {
            v2f *dst = out + (long)b * L;
            v2f y[G][R0];
#pragma unroll
            for (int e = 0; e < G; e++) {
#pragma unroll
                for (int i = 0; i < R0; i++) y[e][i] = x[e * R0 + i];
                v2f pw[R0];
                twiddle_powers<R0>(opaque(w1[e]), pw);
                static_for<0, R0>([&](auto Rr) {
                    constexpr int r = Rr.value, k0 = bitrev<R0>(r);
                    if constexpr (k0 != 0) y[e][r] = cmul_conj(y[e][r], pw[k0]);
                });
                dft_dit<R0, -1>(y[e]);
            }
            // sample 1024*n1 + G*t + e, n1 >= R0/2  ->  output offset 1024*(n1-R0/2) + G*t + e
#pragma unroll
            for (int n1 = HALF; n1 < R0; n1++) {
                v2f *p = dst + 1024 * (n1 - HALF) + G * t;
                if constexpr (G % 2 == 0) {
#pragma unroll
                    for (int e = 0; e < G; e += 2)
                        *reinterpret_cast<v4f *>(p + e) = v4f{y[e][n1].x, y[e][n1].y, y[e + 1][n1].x, y[e + 1][n1].y};
                } else {
#pragma unroll
                    for (int e = 0; e < G; e++) p[e] = y[e][n1];
                }
            }
        }
    }

    // the tail of this call's input is the overlap of the next call (fastfir.cpp:280-300);
    // written to the other half of the ping-pong history so no workgroup can still be reading it
    if (b1 == a.nblocks) {
        v2f *hnext = a.hist_next + (long)ch * L;
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int e = 0; e < G; e++) hnext[1024 * n1 + G * t + e] = carry[e * HALF + n1];
    }
}

template <int LOG2N>
static hipError_t launch_one(const FastFirArgs &a, hipStream_t stream)
{
    using Cfg = FastFirCfg<LOG2N>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastfir_os_kernel<LOG2N, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(a.channels * a.runs), block(Cfg::T);
    if (a.dbg_stage > 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastfir_os_kernel<LOG2N, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fastfir_os_kernel<LOG2N, true>), dim3(1), block, Cfg::LDS_BYTES, stream, a);
    } else {
        hipLaunchKernelGGL((fastfir_os_kernel<LOG2N, false>), grid, block, Cfg::LDS_BYTES, stream, a);
    }
    return hipGetLastError();
}

hipError_t fastfir_launch(int log2n, const FastFirArgs &a, hipStream_t stream)
{
    switch (log2n) {
    case 11: return launch_one<11>(a, stream);
    case 12: return launch_one<12>(a, stream);
    case 13: return launch_one<13>(a, stream);
    case 14: return launch_one<14>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// Host mirror of the kernel's index algebra: which natural-order spectrum bin k the thread t /
// register r of pass F3 holds.  pos = 32 t + bitrev5(r) = 1024 k0 + 32 k1 + k2 and
// k = k0 + R0 (k1 + 32 k2).
int fastfir_bin_of(int log2n, int t, int r)
{
    const int R0 = (1 << log2n) / 1024;
    const int k2 = bitrev<32>(r);
    const int k0 = t >> 5, k1 = t & 31;
    return k0 + R0 * (k1 + 32 * k2);
}

}  // namespace csdr
