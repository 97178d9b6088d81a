// fastfir16k_kernels.hip -- 16-wave variant of the 16384-point overlap-save kernel (K1b).
//
// Same five passes, LDS image and spectrum order conventions as fastfir_os_kernel<14>
// (fastfir_kernels.hip), but 1024 threads per workgroup with 16 points per thread, so that a CU
// holds 16 waves (4 per SIMD) instead of 8 and LDS traffic, VALU work and memory latency of
// different waves overlap.  The radix-32 transforms are shared by a lane PAIR (l, l+32): each
// lane runs a radix-16 in registers and the remaining radix-2 stage crosses the pair with
// v_permlane32_swap (no LDS).  A 1024-point sub-transform now occupies exactly one wave, so the
// F2->F3 and I1->I2 exchanges stay wave-local as before; 3 workgroup barriers per block.
//
// Reference: CFastFIR::ProcessData / CpxMpy (dsp/fastfir.cpp:268-321), CFft::FwdFFT/RevFFT
// (dsp/fft.cpp:416-426).
#include "fft_core.hpp"
#include "fastfir_kernels.h"
#include <cmath>

namespace csdr {

typedef __amdgpu_buffer_rsrc_t rsrc16_t;
typedef unsigned int v4u16 __attribute__((ext_vector_type(4)));
typedef unsigned int v2u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rsrc16_t mk_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v2f ld8(rsrc16_t r, int voff, int soff)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ v4f ld16(rsrc16_t r, int voff, int soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void st8(rsrc16_t r, int voff, int soff, v2f v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u16, v), r, voff, soff, 0);
}

// Exchange across the lane pair (l, l+32).  v_permlane32_swap_b32 vdst, vsrc swaps lanes 32..63 of
// vdst with lanes 0..31 of vsrc.  Applied to two registers (p, q) that hold (E, O) of the partial
// transforms -- E in both registers of the low lane half, O in both of the high half -- it leaves
// every lane with one complete (E[k], O[k]) pair: the low half that of register p's index, the
// high half that of register q's index.  Four complex pairs per statement; the leading two wait
// states cover the VALU -> permlane hazard (what hipcc inserts for its own permlane code).
__device__ __forceinline__ void pairswap4(v2f &p0, v2f &q0, v2f &p1, v2f &q1, v2f &p2, v2f &q2, v2f &p3, v2f &q3)
{
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                 "v_permlane32_swap_b32 %4, %6\n\tv_permlane32_swap_b32 %5, %7\n\t"
                 "v_permlane32_swap_b32 %8, %10\n\tv_permlane32_swap_b32 %9, %11\n\t"
                 "v_permlane32_swap_b32 %12, %14\n\tv_permlane32_swap_b32 %13, %15"
                 : "+v"(p0.x), "+v"(p0.y), "+v"(q0.x), "+v"(q0.y), "+v"(p1.x), "+v"(p1.y), "+v"(q1.x), "+v"(q1.y),
                   "+v"(p2.x), "+v"(p2.y), "+v"(q2.x), "+v"(q2.y), "+v"(p3.x), "+v"(p3.y), "+v"(q3.x), "+v"(q3.y));
}

// e + w*o and e - w*o with three packed instructions (FMA form of the radix-2 butterfly)
__device__ __forceinline__ void bfly_fma(v2f &e, v2f &o, v2f w)
{
#if CSDR_PK_ASM
    v2f t, a;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(o), "v"(w), "v"(e));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "v"(o), "v"(w), "v"(t));
#else
    v2f a = e + cmul_c(o, w);
#endif
    o = e * 2.0f - a;
    e = a;
}

// Final radix-2 stage of a 32-point decimation-in-time DFT that is shared by a lane pair.
// On entry z[r] holds (h ? O : E)[bitrev4(r)] (the two 16-point transforms, one per lane half).
// PARITY = false: registers (r, r+1), r = bitrev4(i): lane half h finishes bin k = i + 8h:
//                 z[r] = X[k], z[r+1] = X[k+16]                       (twiddles tw[h*8 + i])
// PARITY = true:  registers (r, r+8), r < 8:        lane half h finishes bin k = bitrev4(r) + h:
//                 z[r] = X[k], z[r+8] = X[k+16]                       (twiddles tw[h*8 + r])
template <bool PARITY>
__device__ __forceinline__ void pair_stage(v2f (&z)[16], const v2f *tw)
{
    v2f w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = tw[i];
    if constexpr (PARITY) {
        pairswap4(z[0], z[8], z[1], z[9], z[2], z[10], z[3], z[11]);
        pairswap4(z[4], z[12], z[5], z[13], z[6], z[14], z[7], z[15]);
#pragma unroll
        for (int r = 0; r < 8; r++) bfly_fma(z[r], z[r + 8], w[r]);
    } else {
        // i = 0..7 <-> register pair (bitrev4(i), bitrev4(i)+1) = (0,1) (8,9) (4,5) (12,13) (2,3) (10,11) (6,7) (14,15)
        pairswap4(z[0], z[1], z[8], z[9], z[4], z[5], z[12], z[13]);
        pairswap4(z[2], z[3], z[10], z[11], z[6], z[7], z[14], z[15]);
        static_for<0, 8>([&](auto I) {
            constexpr int i = I.value, r = bitrev<16>(i);
            bfly_fma(z[r], z[r + 1], w[i]);
        });
    }
}

__device__ __forceinline__ int pad33(int pos) { return pos + (pos >> 5); }

constexpr int K16_N = 16384, K16_T = 1024, K16_L = 8192;
constexpr int K16_LDS_DATA = K16_N + K16_N / 32;          // one pad element per 32: 8-byte accesses only
constexpr int K16_TW = 1024 + 48;                          // 32x32 inter-pass table + 3 pair-stage sets
constexpr int K16_LDS_BYTES = (K16_LDS_DATA + K16_TW) * 8;

template <bool DBG>
__device__ __forceinline__ bool k16_dbg(const FastFirArgs &a, const v2f *lds, int stage)
{
    if constexpr (DBG) {
        if (a.dbg_stage == stage) {
            __syncthreads();
            for (int i = threadIdx.x; i < K16_N; i += K16_T) a.dbg[i] = lds[pad33(i)];
            return true;
        }
    }
    return false;
}

template <bool DBG>
__global__ __launch_bounds__(K16_T)
void fastfir_os16k_kernel(FastFirArgs a)
{
    constexpr int L = K16_L, T = K16_T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    v2f *tw2 = lds + K16_LDS_DATA;               // tw2[k1*32+n] = W_1024^{n k1}
    const v2f *ptw = tw2 + 1024;                 // pair-stage twiddles, 3 sets x 2 lane halves x 8
    const int t = threadIdx.x;
    const int sb = t >> 6, lane = t & 63, h = lane >> 5, sn = lane & 31;

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
    if (ch >= a.channels || b0 >= b1) return;

    for (int i = t; i < K16_TW; i += T) tw2[i] = a.tw2[i];
    const rsrc16_t r_in = mk_rsrc(a.in + (long)ch * a.in_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc16_t r_hist = mk_rsrc(a.hist + (long)ch * L, L * 8u);
    const rsrc16_t r_out = mk_rsrc(a.out + (long)ch * a.out_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc16_t r_h = mk_rsrc(a.h + (long)ch * a.h_stride, K16_N * 8u);
    const rsrc16_t r_hn = mk_rsrc(a.hist_next + (long)ch * L, L * 8u);
    const v2f w1 = a.tw1[t];                     // W_N^{n2}, n2 = t: this thread's column

    const int colbase = pad33(1024 * sb) + sn;             // column sn of sub-transform sb: + 33*n1
    const int rowbase = 33 * (32 * sb + sn);               // row 32 sb + sn: + n
    const v2f *twA = ptw + 0 + 8 * h;            // W32^{+(i+8h)}
    const v2f *twB = ptw + 16 + 8 * h;           // W32^{+(bitrev4(r)+h)}, r < 8
    const v2f *twC = ptw + 32 + 8 * h;           // W32^{-(i+8h)}

    for (int b = b0; b < b1; b++) {
        v2f x[16];
        // ---------------- F1: column t, radix-16 DIF, twiddle, scatter ------------------------
        if (b == 0) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) x[n1] = ld8(r_hist, t * 8, n1 * 8192);
        } else {                                 // the old half comes back from L2
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) x[n1] = ld8(r_in, t * 8, (b - 1) * (L * 8) + n1 * 8192);
        }
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) x[8 + n1] = ld8(r_in, t * 8, b * (L * 8) + n1 * 8192);
        if (b == a.nblocks - 1) {                // this call's tail is the next call's overlap
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) st8(r_hn, t * 8, n1 * 8192, x[8 + n1]);
        }
        dft_dif<16, +1>(x);
        {
            v2f pw[16];
            twiddle_powers<16>(opaque(w1), pw);
            static_for<1, 16>([&](auto Rr) {
                constexpr int r = Rr.value, k0 = bitrev<16>(r);
                x[r] = cmul(x[r], pw[k0]);
            });
        }
        // no barrier here: pass F1 writes exactly the LDS cells this thread itself read in pass I3 of
        // the previous block (same column, all rows), so program order is enough
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value, k0 = bitrev<16>(r);
            lds[pad33(1024 * k0 + t)] = x[r];
        });
        __syncthreads();
        if (k16_dbg<DBG>(a, lds, 1)) return;

        // ---------------- F2: 32-point DIT over column sn; lane half h owns n1 = 2m+h ------------
#pragma unroll
        for (int m = 0; m < 16; m++) x[m] = lds[colbase + 33 * (2 * m + h)];
        dft_dif<16, +1>(x);
        pair_stage<false>(x, twA);               // x[bitrev4(i)] = bin i+8h, x[bitrev4(i)+1] = bin i+8h+16
        static_for<0, 8>([&](auto I) {
            constexpr int i = I.value, r = bitrev<16>(i);
            const int k1 = i + 8 * h;
            x[r] = cmul(x[r], tw2[k1 * 32 + sn]);          // k1 = 0: the table holds 1+0j
            x[r + 1] = cmul(x[r + 1], tw2[(k1 + 16) * 32 + sn]);
            lds[colbase + 33 * k1] = x[r];
            lds[colbase + 33 * (k1 + 16)] = x[r + 1];
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k16_dbg<DBG>(a, lds, 2)) return;

        // ---------------- F3 + H + I1 on row 32 sb + sn; lane half h owns n = 2m+h ----------------
        {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = lds[rowbase + 2 * m + h];
            dft_dif<16, +1>(x);
            pair_stage<true>(x, twB);            // x[r] = bin bitrev4(r)+h, x[r+8] = that + 16 (r < 8)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const v4f hv = ld16(r_h, t * 16, j * (T * 16));      // L2 hit; 4 waves per SIMD cover it
                x[2 * j] = cmul(x[2 * j], v2f{hv.x, hv.y});
                x[2 * j + 1] = cmul(x[2 * j + 1], v2f{hv.z, hv.w});
            }
            // inverse DIT: lane half h holds the bins of parity h; input index m = (k2 - h)/2:
            // register r < 8 is m = bitrev3(r), register r+8 is m = bitrev3(r) + 8
            v2f u[16];
            static_for<0, 8>([&](auto Rr) {
                constexpr int r = Rr.value, m = bitrev<8>(r);
                u[m] = x[r];
                u[m + 8] = x[r + 8];
            });
            dft_dif<16, -1>(u);
            pair_stage<false>(u, twC);           // u[bitrev4(i)] = sample i+8h, u[bitrev4(i)+1] = sample i+8h+16
            static_for<0, 8>([&](auto I) {
                constexpr int i = I.value, r = bitrev<16>(i);
                lds[rowbase + i + 8 * h] = u[r];
                lds[rowbase + i + 8 * h + 16] = u[r + 1];
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k16_dbg<DBG>(a, lds, 3)) return;

        // ---------------- I2: 32-point inverse DIT over column sn ------------------------------------
        static_for<0, 16>([&](auto M) {
            constexpr int m = M.value;
            const int k1 = 2 * m + h;
            x[m] = cmul_conj(lds[colbase + 33 * k1], tw2[k1 * 32 + sn]);
        });
        dft_dif<16, -1>(x);
        pair_stage<false>(x, twC);
        static_for<0, 8>([&](auto I) {
            constexpr int i = I.value, r = bitrev<16>(i);
            lds[colbase + 33 * (i + 8 * h)] = x[r];
            lds[colbase + 33 * (i + 8 * h + 16)] = x[r + 1];
        });
        __syncthreads();
        if (k16_dbg<DBG>(a, lds, 4)) return;

        // ---------------- I3: column t, conj twiddle, radix-16 DIT inverse, store valid half ---------
        static_for<0, 16>([&](auto Rr) {
            constexpr int r = Rr.value, k0 = bitrev<16>(r);
            x[r] = lds[pad33(1024 * k0 + t)];
        });
        {
            v2f pw[16];
            twiddle_powers<16>(opaque(w1), pw);
            static_for<1, 16>([&](auto Rr) {
                constexpr int r = Rr.value, k0 = bitrev<16>(r);
                x[r] = cmul_conj(x[r], pw[k0]);
            });
        }
        dft_dit<16, -1>(x);
#pragma unroll
        for (int n1 = 8; n1 < 16; n1++) st8(r_out, t * 8, b * (L * 8) + (n1 - 8) * 8192, x[n1]);
    }
}

hipError_t fastfir16k_launch(const FastFirArgs &a, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastfir_os16k_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, K16_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fastfir_os16k_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, K16_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (a.dbg_stage > 0)
        hipLaunchKernelGGL(fastfir_os16k_kernel<true>, dim3(1), dim3(K16_T), K16_LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL(fastfir_os16k_kernel<false>, dim3(a.channels * a.runs), dim3(K16_T), K16_LDS_BYTES, stream, a);
    return hipGetLastError();
}

// spectrum bin held by register r of thread t after pass F3 of this variant
int fastfir16k_bin_of(int t, int r)
{
    const int sb = t >> 6, lane = t & 63, h = lane >> 5, sn = lane & 31;
    const int k2 = bitrev<16>(r & 7) + h + ((r & 8) ? 16 : 0);
    return sb + 16 * (sn + 32 * k2);
}

// the 48 pair-stage twiddles appended to the inter-pass table: sets A, B, C x lane half x 8
void fastfir16k_pair_twiddles(float *out96)
{
    for (int set = 0; set < 3; set++)
        for (int h = 0; h < 2; h++)
            for (int i = 0; i < 8; i++) {
                int k = (set == 1) ? bitrev<16>(i) + h : i + 8 * h;
                double ang = 2.0 * 3.14159265358979323846 * k / 32.0 * (set == 2 ? -1.0 : 1.0);
                out96[2 * (set * 16 + h * 8 + i)] = (float)cos(ang);
                out96[2 * (set * 16 + h * 8 + i) + 1] = (float)sin(ang);
            }
}

}  // namespace csdr
