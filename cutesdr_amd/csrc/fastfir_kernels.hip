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
#include "launch_once.hpp"
#include "fastfir_dev.hpp"
#include "fastfir_kernels.h"

namespace csdr {

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

    const rsrc_t r_in = make_rsrc(a.in + (long)ch * a.in_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc_t r_hist = make_rsrc(a.hist + (long)ch * L, L * 8u);
    const rsrc_t r_out = make_rsrc(a.out + (long)ch * a.out_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc_t r_h = make_rsrc(a.h + (long)ch * a.h_stride, N * 8u);
    const int voff = t * (G * 8);             // this thread's G adjacent columns, in bytes
    // one hop-half of samples: rows n1 = 0..HALF-1 of 1024 samples, columns G*t .. G*t+G-1
    auto load_half = [&](rsrc_t r, int soff, v2f (&dst)[16]) {
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int e = 0; e < G; e += 2) {
                v4f v = buf_load16(r, voff + e * 8, soff + n1 * 8192);
                dst[e * HALF + n1] = v2f{v.x, v.y};
                dst[(e + 1) * HALF + n1] = v2f{v.z, v.w};
            }
    };

    // outer-pass base twiddles W_N^{n2}, n2 = G*t+e: constant over the block loop
    v2f w1[G];
#pragma unroll
    for (int e = 0; e < G; e++) w1[e] = a.tw1[G * t + e];

    // x[e*R0 + n1] <-> sample 1024*n1 + G*t + e of the block
    v2f x[32];
    v2f carry[16];       // new half of the previous block = old half of this one
    v2f nxt[16];         // new half of this block, fetched while the previous block finished

    // old half of the first block: previous call's tail (b0 == 0) or the input itself
    if (b0 == 0) load_half(r_hist, 0, carry);
    else load_half(r_in, (b0 - 1) * (L * 8), carry);
    load_half(r_in, b0 * (L * 8), nxt);

    const int sb = t >> 5, sn = t & 31;       // sub-transform and column of passes F2 / I2

    for (int b = b0; b < b1; b++) {
        // ---------------- F1: load, radix-R0 DIF, twiddle, scatter to LDS ----------------
#pragma unroll
        for (int e = 0; e < G; e++)
#pragma unroll
            for (int n1 = 0; n1 < HALF; n1++) {
                x[e * R0 + n1] = carry[e * HALF + n1];
                x[e * R0 + HALF + n1] = nxt[e * HALF + n1];
                carry[e * HALF + n1] = nxt[e * HALF + n1];
            }
        // next block's new samples: issued as soon as nxt is free, in flight during the whole block
        if (b + 1 < b1) load_half(r_in, (b + 1) * (L * 8), nxt);
        __builtin_amdgcn_sched_barrier(0);
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
        // no barrier here: pass F1 writes exactly the LDS cells this thread itself read in pass I3 of
        // the previous block (same columns, all rows), so program order is enough
        {
            v4f wv[16];                        // (R0 rows) x (G/2 column pairs) = 16 float4
#pragma unroll
            for (int r = 0; r < R0; r++)
#pragma unroll
                for (int e = 0; e < G; e += 2)
                    wv[r * (G / 2) + e / 2] = store_operand(x[e * R0 + r], x[(e + 1) * R0 + r]);
            CSDR_STORE_GROUP_BEGIN();
            static_for<0, R0>([&](auto Rr) {
                constexpr int r = Rr.value, k0 = bitrev<R0>(r);
                const int base = lds_pad(1024 * k0 + G * t);
#pragma unroll
                for (int e = 0; e < G; e += 2)
                    *reinterpret_cast<v4f *>(&lds[base + e]) = wv[r * (G / 2) + e / 2];
            });
            CSDR_STORE_GROUP_END();
        }
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
            // 8-byte LDS stores: no tuple assembly, not subject to the wide-store hazard, so the
            // scheduler may interleave them with the last butterfly stage
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
            // H[k] comes from L2; issue the loads before the LDS reads and the forward butterflies
            // so that their latency is covered (pinned: the scheduler would sink them to the use)
            v4f hv[16];
#pragma unroll
            for (int j = 0; j < 16; j++) hv[j] = buf_load16(r_h, t * 16, j * (T * 16));
            __builtin_amdgcn_sched_barrier(0);
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
                x[2 * j] = cmul(x[2 * j], v2f{hv[j].x, hv[j].y});
                x[2 * j + 1] = cmul(x[2 * j + 1], v2f{hv[j].z, hv[j].w});
            }
            dft_dit<32, -1>(x);
            v4f *wrow = reinterpret_cast<v4f *>(&lds[34 * t]);
            v4f wv[16];
#pragma unroll
            for (int j = 0; j < 16; j++) wv[j] = store_operand(x[2 * j], x[2 * j + 1]);
            CSDR_STORE_GROUP_BEGIN();
#pragma unroll
            for (int j = 0; j < 16; j++) wrow[j] = wv[j];
            CSDR_STORE_GROUP_END();
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
        // next block's new samples: issued now, consumed at the top of the next iteration
        static_for<0, R0>([&](auto Rr) {
            constexpr int r = Rr.value, k0 = bitrev<R0>(r);
            const int base = lds_pad(1024 * k0 + G * t);
#pragma unroll
            for (int e = 0; e < G; e += 2) {
                v4f v = *reinterpret_cast<const v4f *>(&lds[base + e]);
                x[e * R0 + r] = v2f{v.x, v.y};
                x[(e + 1) * R0 + r] = v2f{v.z, v.w};
            }
        });
        {
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
            v4f sv[8];                         // (R0/2 rows) x (G/2 column pairs) = 8 float4
#pragma unroll
            for (int n1 = HALF; n1 < R0; n1++)
#pragma unroll
                for (int e = 0; e < G; e += 2)
                    sv[(n1 - HALF) * (G / 2) + e / 2] = store_operand(y[e][n1], y[e + 1][n1]);
            CSDR_STORE_GROUP_BEGIN();
#pragma unroll
            for (int n1 = HALF; n1 < R0; n1++)
#pragma unroll
                for (int e = 0; e < G; e += 2)
                    buf_store16(r_out, voff + e * 8, b * (L * 8) + (n1 - HALF) * 8192,
                                sv[(n1 - HALF) * (G / 2) + e / 2]);
            CSDR_STORE_GROUP_END();
        }
    }

    // the tail of this call's input is the overlap of the next call (fastfir.cpp:280-300);
    // written to the other half of the ping-pong history so no workgroup can still be reading it
    if (b1 == a.nblocks) {
        const rsrc_t r_hn = make_rsrc(a.hist_next + (long)ch * L, L * 8u);
        v4f sv[8];
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int e = 0; e < G; e += 2)
                sv[n1 * (G / 2) + e / 2] = store_operand(carry[e * HALF + n1], carry[(e + 1) * HALF + n1]);
        CSDR_STORE_GROUP_BEGIN();
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int e = 0; e < G; e += 2) buf_store16(r_hn, voff + e * 8, n1 * 8192, sv[n1 * (G / 2) + e / 2]);
        CSDR_STORE_GROUP_END();
    }
}

template <int LOG2N>
static hipError_t launch_one(const FastFirArgs &a, hipStream_t stream)
{
    using Cfg = FastFirCfg<LOG2N>;
    {   // once per device (launch_once.hpp)
        hipError_t e = CSDR_MAX_LDS_ONCE((&fastfir_os_kernel<LOG2N, false>), Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    dim3 grid(a.channels * a.runs), block(Cfg::T);
    if (a.dbg_stage > 0) {
        hipError_t e = CSDR_MAX_LDS_ONCE((&fastfir_os_kernel<LOG2N, true>), Cfg::LDS_BYTES);
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
