// fastfir2_kernels.hip -- batched overlap-save FFT FIR for gfx950, software-pipelined build (K1): N = 16384 (the
// headline configuration), and since round 5 N = 8192, 4096 and 2048 -- the same passes with an outer pass of radix 8 / 4 /
// 2 over four / eight / sixteen columns per thread (the instruction stream of the 16384-point instantiation is unchanged,
// checked in the ISA), 4096 points as two and 2048 points as four blocks side by side per workgroup (K1Cfg below).
//
// Same algorithm, LDS image, spectrum order and HBM traffic as fastfir_os_kernel
// (fastfir_kernels.hip; reference dsp/fastfir.cpp:268-321, dsp/fft.cpp:416-426): passes
// F1 (radix-R0 from HBM) | F2 (radix-32) | F3 + H + I1 (registers) | I2 (radix-32) | I3 (radix-R0,
// store the valid half).  What differs is the ORDER of the instruction stream.  The first build
// left the compiler a free hand and got, per pass, "all LDS reads | all butterflies | all LDS
// writes": with two waves per SIMD running in lockstep between the workgroup barriers, the LDS
// pipe and the VALU took turns (13.8k VALU + 9.2k LDS cycles per block against 24.8k measured).
// Here every pass is cut into groups of four points (fft_core.hpp: head / tail4 / head4 / tail):
//   * the points of a group are fetched in the order the first two stages need them, so the
//     butterflies start after four reads instead of thirty-two;
//   * a group's results are written while the next group's butterflies issue (one group behind);
//   * the 8-byte LDS accesses are relaxed atomics: hipcc neither merges them into ds_read2_b64 /
//     ds_write2_b64 (half the LDS rate of ds_read_b64) nor reorders them, and still counts them
//     in its s_waitcnt bookkeeping;
//   * sched_barrier(0) between groups pins that order against the machine scheduler.
// The outer-pass twiddle powers are computed once per block (pass I3) and reused by pass F1 of the
// next block.
#define CSDR_FMA_BFLY 1          // FMA-form decimation-in-time butterflies (fft_core.hpp)
#define CSDR_PLAIN_CONST_FMA 1   // ... written without asm where the twiddle is a compile-time constant
#include "launch_once.hpp"
#include "fastfir_dev.hpp"
#include "fastfir_kernels.h"

namespace csdr {

#define CSDR_SB() __builtin_amdgcn_sched_barrier(0)

// Round-3 knobs (each A/B'd with tools/ab_many.sh).  The kernel runs at the chip's POWER limit
// (1.87-1.91 GHz in-kernel against 2.4 nominal, tools/k1_cycles.py): what shortens a launch is less energy per
// block -- fewer instructions, fewer bytes moved -- not fewer stall cycles.
#ifndef K1_LDAUX
#define K1_LDAUX 2          // cache policy of the input loads: nt (every sample is read once)
#endif
#ifndef K1_STAUX
#define K1_STAUX 2          // ... and of the output stores (written once)
#endif
#ifndef K1_HREG
#define K1_HREG 8           // float4 of this thread's share of H that stay in registers for the whole run (0..8 fit)
#endif
#ifndef K1_HREG4K
#define K1_HREG4K 4         // ... at N = 4096, whose outer pass (eight columns of four points) keeps more values live
#endif
#ifndef K1_W1_FETCH_MAX_R0
#define K1_W1_FETCH_MAX_R0 2 // outer radix up to which the base twiddles are fetched per block instead of held (2: N = 2048 only;
                             // at N = 4096 the fetch removes the kernel's 40 bytes of scratch and measures 1.5 % SLOWER)
#endif
#ifndef K1_HREG2K
#define K1_HREG2K 8         // ... at N = 2048 (the sixteen base twiddles are fetched where they are used: 0 / 4 / 8 resident measured 0.710 / 0.697 / 0.689 ms)
#endif

// Diagnostic build only (-DCSDR_K1_STAMPS, tools/k1_stamps.py): cycle shares of the passes of one block,
// summed per wave in scalar registers and written to a.dbg after the loop.  No stamp executes otherwise.
#ifdef CSDR_K1_STAMPS
#define CSDR_STAMP(i)                                                                         \
    do {                                                                                      \
        CSDR_SB();                                                                            \
        unsigned long long now_;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");         \
        CSDR_SB();                                                                            \
        acc_[i] += now_ - last_;                                                              \
        last_ = now_;                                                                         \
    } while (0)
#else
#define CSDR_STAMP(i) do { } while (0)
#endif

// Priority ladder.  The two waves of a SIMD (w and w+4) run the same code between two workgroup barriers, and
// the SIMD issues by priority, then age: left alone, the older wave takes every slot it can use, reaches the
// barrier thousands of cycles early and waits while the younger one runs alone at a single wave's issue
// rate (in-kernel stamps: 8.3k of 21.7k cycles per block spent waiting).  Lowering the priority step by step
// through the interval (3, 2, 1, 0) makes whichever wave is behind the preferred one: the pair stays within
// one segment of each other and the waits fall to 2k cycles.
#define CSDR_PRIO(p) __builtin_amdgcn_s_setprio(p)

// Timing ablations (tools/altlib.py NAME -DK1_ABLATE -DABL_...; the results are garbage, never shipped): H from a
// constant instead of L2, pass twiddles from a constant instead of LDS, no workgroup barriers, no output stores.
#ifdef K1_ABLATE
__device__ __forceinline__ void keep_alive(v4f v) { asm volatile("" ::"v"(v)); }
#endif

// (N = 2048 likewise: FOUR blocks of one wave each, and a block's barriers are wave barriers.)
// N = 4096 runs TWO blocks side by side in one workgroup: a block of 4096 points is 128 threads and 43 KB of LDS, three
// workgroups -- six waves -- per CU, against the eight (two per SIMD) the schedule below is made for.  Two "virtual
// workgroups" of 128 threads, each with its own LDS image and its own run of blocks, sharing the twiddle table and the
// (then merely coincident) barriers: 256 threads, 78 KB, two per CU, eight waves -- the shape of the 8192-point launch.
template <int LOG2N>
struct K1Cfg {
    using Base = FastFirCfg<LOG2N>;
    static constexpr int VW = LOG2N == 12 ? 2 : (LOG2N == 11 ? 4 : 1);      // virtual workgroups per workgroup
    static constexpr int TV = Base::T;                                      // threads of one block
    static constexpr int T = VW * TV;
    static constexpr int LDS_BYTES = (VW * Base::LDS_DATA + 1024) * 8;
};

template <int LOG2N>
__global__ __launch_bounds__(K1Cfg<LOG2N>::T) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fastfir_os2_kernel(FastFirArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_FF);
    using Cfg = FastFirCfg<LOG2N>;
    constexpr int N = Cfg::N, T = Cfg::T, R0 = Cfg::R0, G = Cfg::G, L = N / 2;
    constexpr int HALF = R0 / 2;
    constexpr int VW = K1Cfg<LOG2N>::VW;
    constexpr int HREG = LOG2N == 12 ? K1_HREG4K : (LOG2N == 11 ? K1_HREG2K : K1_HREG);   // resident float4 of H
    static_assert(R0 == 16 || R0 == 8 || R0 == 4 || R0 == 2, "the grouped outer pass is written for N = 2048 ... 16384");
    // a block of 2048 points is ONE wave: its two "workgroup" barriers are wave barriers (the four blocks of a workgroup
    // then run free of each other)
    auto block_barrier = [] {
        if constexpr (LOG2N == 11) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = VW == 1 ? (int)threadIdx.x : (int)threadIdx.x % T, vw = VW == 1 ? 0 : (int)threadIdx.x / T;
    v2f *lds = reinterpret_cast<v2f *>(smem_raw) + vw * Cfg::LDS_DATA;
    v2f *tw2 = reinterpret_cast<v2f *>(smem_raw) + VW * Cfg::LDS_DATA;     // tw2[k1*32 + n2] = W_1024^{n2*k1}
    if constexpr (VW > 1) {                   // (a virtual workgroup that has nothing to do leaves below: the table is whole first)
        for (int i = threadIdx.x; i < 1024; i += VW * T) tw2[i] = a.tw2[i];
        __syncthreads();
    }

    int wg = blockIdx.x * VW + vw, ch, run;
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
    // VW == 2 (N = 4096): the two virtual workgroups of a real one share its s_barrier, and one of them may END here -- or
    // walk one block fewer in the last run -- while the other keeps arriving at barriers.  That is defined on this target:
    // the gfx9 barrier counts the waves of the workgroup that have not terminated (an s_endpgm wave leaves the count), so
    // the survivor's barriers complete with its own waves.  The HIP model does not promise it, the build pins the target:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fastfir_os2_kernel<12> relies on gfx950's s_barrier ignoring terminated waves (two virtual workgroups per workgroup)"
#endif
    if (ch >= a.channels || b0 >= b1) return;          // uniform per (virtual) workgroup

    if constexpr (VW == 1)
        for (int i = t; i < 1024; i += T) tw2[i] = a.tw2[i];

    const rsrc_t r_in = make_rsrc(a.in + (long)ch * a.in_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc_t r_hist = make_rsrc(a.hist + (long)ch * L, L * 8u);
    const rsrc_t r_out = make_rsrc(a.out + (long)ch * a.out_stride, (unsigned)a.nblocks * L * 8u);
    const rsrc_t r_h = make_rsrc(a.h + (long)ch * a.h_stride, N * 8u);
    // A thread's G columns of the outer pass are G / 2 PAIRS, pair pp = columns PSTEP pp + 2t, + 1 (PSTEP = 2 T: every
    // load, store and 16-byte LDS access of a wave covers 64 adjacent pairs -- with G adjacent columns per thread the
    // four 16-byte accesses of the 4096-point kernel each touched a quarter of every line: 1.8 ms instead of 0.8)
    constexpr int PSTEP = 2 * T, PSTEP_LDS = PSTEP + 2 * (PSTEP / 32), PSTEP_B = PSTEP * 8;
    const int voff = t * 16;
    // one hop-half of samples: rows n1 = 0..HALF-1 of 1024 samples, this thread's G / 2 column pairs
    auto load_half = [&](rsrc_t r, int soff, v2f (&dst)[16]) {
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int pp = 0; pp < G / 2; pp++) {
                v4f v = buf_load16_aux<K1_LDAUX>(r, voff + pp * PSTEP_B, soff + n1 * 8192);
                dst[(2 * pp) * HALF + n1] = v2f{v.x, v.y};
                dst[(2 * pp + 1) * HALF + n1] = v2f{v.z, v.w};
            }
    };

    // outer-pass twiddles W_N^{n2 k0}, n2 = 2t+e, k0 = 1..15: pw[e][k0].  Rebuilt at the top of every
    // I3 and kept for F1 of the next block only: live across the whole loop they would not fit beside H
    // (N = 2048: sixteen base twiddles and no powers -- they are fetched where the outer passes use them, 8 KB of table
    // that stays in the vector cache, instead of thirty-two registers held for the whole run)
    constexpr bool W1_RESIDENT = R0 > K1_W1_FETCH_MAX_R0;    // the base twiddles stay in registers for the whole run
    auto w_pair = [&](int pp) { return *reinterpret_cast<const v4f *>(a.tw1 + PSTEP * pp + 2 * t); };
    v2f w1[R0 > 2 ? G : 1];
    auto load_w1 = [&] {
        if constexpr (R0 > 2) {
#pragma unroll
            for (int pp = 0; pp < G / 2; pp++) { const v4f v = w_pair(pp); w1[2 * pp] = v2f{v.x, v.y}; w1[2 * pp + 1] = v2f{v.z, v.w}; }
        }
    };
    load_w1();
    v2f pw[G][R0];
    if constexpr (R0 > 2) {
#pragma unroll
        for (int e = 0; e < G; e++) twiddle_powers<R0>(opaque(w1[e]), pw[e]);
    }

    // H: float4 j of this thread (fastfir2_bin_of) multiplies in F3's tail group j / 2.  The first K1_HREG of the
    // sixteen stay in registers for the whole run -- all the registers the kernel has to spare: a 1 KB fetch from
    // L2 costs about as much energy as four packed instructions -- the rest is fetched from L2 for every block
    v4f hv[16];
#pragma unroll
    for (int j = 0; j < HREG; j++) hv[j] = buf_load16(r_h, t * 16, j * (T * 16));
#ifdef K1_ABLATE
    v4f habl = {1.0f, 0.0f, 1.0f, 0.0f};
    asm volatile("" : "+v"(habl));
    v2f twabl = {0.8f, 0.6f};
    asm volatile("" : "+v"(twabl));
#endif
    v2f x[32];           // phase B: the 32 points of this thread
    // The two halves of a block's input, [e * HALF + n1] = column PSTEP (e / 2) + 2t + (e & 1), row n1.  The new
    // half of one block is the old half of the next: the block loop is unrolled by two and the buffers
    // swap roles, so nothing is copied; the samples after next are fetched into the old half's registers
    // as soon as the first butterfly stage has read them.
    v2f hp[16], hq[16];

    if (b0 == 0) load_half(r_hist, 0, hp);
    else load_half(r_in, (b0 - 1) * (L * 8), hp);
    load_half(r_in, b0 * (L * 8), hq);

    const int sb = t >> 5, sn = t & 31;       // sub-transform and column of passes F2 / I2
    v2f *const col = lds + lds_pad(1024 * sb) + sn;         // F2 / I2: point n1 at col[34 * n1]
    const v2f *const twc = tw2 + sn;                        // twiddle k1 at twc[32 * k1]
    v2f *const rowp = lds + 34 * t;                         // F3: this thread's 32 consecutive points
    v2f *const outer = lds + lds_pad(2 * t);                // F1 / I3: row k0, pair pp at outer[OUTER_ROW k0 + PSTEP_LDS pp]
    constexpr int OUTER_ROW = 1024 + 2 * (1024 / 32);       // padded elements between rows of the outer pass

#ifdef K1_CYC          // diagnostic build (tools/k1_cycles.py): shader cycles and real time of the whole block loop
    const unsigned long long cyc0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef CSDR_K1_STAMPS
    unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    // one block: [oldh | newh] in, the valid half out; block b+1's new half is left in oldh
    auto one_block = [&](const int b, v2f (&oldh)[16], v2f (&newh)[16]) {
        // ================= F1: radix-16 forward transform of [oldh | newh], twiddle, scatter to LDS =================
        // (a decimation-in-time network like every transform of this kernel -- FMA butterflies; its
        // bit-reversed input order costs nothing, the samples sit in registers: position bitrev(n1) <- row n1)
        CSDR_SB();
        CSDR_PRIO(1);
        {
            v2f y[G][R0];
            static_for<0, HALF>([&](auto N1) {
                static_for<0, G>([&](auto E) {
                    constexpr int e = E.value, n1 = N1.value, po = bitrev<R0>(n1), pn = bitrev<R0>(HALF + n1);
                    y[e][po] = oldh[e * HALF + n1]; y[e][pn] = newh[e * HALF + n1];
                });
            });
            if constexpr (R0 == 2) {
                static_for<0, G>([&](auto E) { bfly_dit<0, +1>(y[E.value][0], y[E.value][1]); });
            } else {
                static_for<0, (R0 >= 8 ? R0 / 4 : 1)>([&](auto Gg) {
                    static_for<0, G>([&](auto E) { dit_head4<Gg.value, R0, +1>(y[E.value]); });
                });
            }
            CSDR_SB();
            // block b+1's new half: into the registers of the old half, which the butterflies above have read
            // (unconditional, so that the block stays one straight line of code: after the last block of the
            // call the last block is fetched again, after the last of a run the samples are simply not used)
            load_half(r_in, (b + 1 < a.nblocks ? b + 1 : a.nblocks - 1) * (L * 8), oldh);
            CSDR_SB();
            CSDR_PRIO(0);
            v4f wv[16];                                    // (R0 rows) x (G / 2 column pairs): row k0, pair pp at [k0 * (G / 2) + pp]
            if constexpr (R0 == 16) {
                static_for<0, R0 / 4 + 1>([&](auto Ii) {
                    constexpr int i = Ii.value;                // tail group i finishes rows k0 = i, i+4, i+8, i+12
                    if constexpr (i < R0 / 4) {
                        dit_tail<i, R0, +1>(y[0]);
                        dit_tail<i, R0, +1>(y[1]);
                        static_for<0, 4>([&](auto P) {
                            constexpr int k0 = i + 4 * P.value;
                            if constexpr (k0 != 0) {
                                y[0][k0] = cmul(y[0][k0], pw[0][k0]);
                                y[1][k0] = cmul(y[1][k0], pw[1][k0]);
                            }
                            wv[k0] = store_operand(y[0][k0], y[1][k0]);
                        });
                    }
                    if constexpr (i > 0) {                 // rows of the previous group: written while this one computes
                        CSDR_STORE_GROUP_BEGIN();
                        static_for<0, 4>([&](auto P) {
                            constexpr int k0 = (i - 1) + 4 * P.value;
                            *reinterpret_cast<v4f *>(outer + OUTER_ROW * k0) = wv[k0];
                        });
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            } else if constexpr (R0 == 2) {
                // N = 2048: the one butterfly above was the whole outer transform; group i = rows 0, 1 of the pairs 2i, 2i + 1
                v4f wq[8];
                static_for<0, 8>([&](auto PP) { wq[PP.value] = w_pair(PP.value); });
                static_for<0, 5>([&](auto Ii) {
                    constexpr int i = Ii.value;
                    if constexpr (i < 4) {
                        static_for<0, 2>([&](auto Q) {
                            constexpr int pp = 2 * i + Q.value;
                            y[2 * pp][1] = cmul(y[2 * pp][1], v2f{wq[pp].x, wq[pp].y});
                            y[2 * pp + 1][1] = cmul(y[2 * pp + 1][1], v2f{wq[pp].z, wq[pp].w});
                            wv[pp] = store_operand(y[2 * pp][0], y[2 * pp + 1][0]);
                            wv[8 + pp] = store_operand(y[2 * pp][1], y[2 * pp + 1][1]);
                        });
                    }
                    if constexpr (i > 0) {
                        CSDR_STORE_GROUP_BEGIN();
                        static_for<0, 2>([&](auto Q) {
                            constexpr int pp = 2 * (i - 1) + Q.value;
                            *reinterpret_cast<v4f *>(outer + PSTEP_LDS * pp) = wv[pp];
                            *reinterpret_cast<v4f *>(outer + OUTER_ROW + PSTEP_LDS * pp) = wv[8 + pp];
                        });
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            } else {
                // N = 8192: the last stage (8) in four groups, group i finishes rows k0 = i, i + 4 of the G = 4 columns;
                // N = 4096: the head group was the whole radix-4 transform, group i is row k0 = i of the G = 8 columns.
                // Either way four float4 per group, stored one group behind, as above.
                constexpr int NG = 4, RPG = R0 / NG;           // rows per group
                static_for<0, NG + 1>([&](auto Ii) {
                    constexpr int i = Ii.value;
                    if constexpr (i < NG) {
                        if constexpr (R0 == 8)
                            static_for<0, G>([&](auto E) { bfly_dit<4 * i, +1>(y[E.value][i], y[E.value][i + 4]); });
                        static_for<0, RPG>([&](auto P) {
                            constexpr int k0 = i + NG * P.value;
                            static_for<0, G>([&](auto E) {
                                if constexpr (k0 != 0) y[E.value][k0] = cmul(y[E.value][k0], pw[E.value][k0]);
                            });
                            static_for<0, G / 2>([&](auto PP) {
                                wv[k0 * (G / 2) + PP.value] = store_operand(y[2 * PP.value][k0], y[2 * PP.value + 1][k0]);
                            });
                        });
                    }
                    if constexpr (i > 0) {
                        CSDR_STORE_GROUP_BEGIN();
                        static_for<0, RPG>([&](auto P) {
                            constexpr int k0 = (i - 1) + NG * P.value;
                            static_for<0, G / 2>([&](auto PP) {
                                *reinterpret_cast<v4f *>(outer + OUTER_ROW * k0 + PSTEP_LDS * PP.value) = wv[k0 * (G / 2) + PP.value];
                            });
                        });
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            }
        }
        CSDR_STAMP(0);                                 // F1 (and the loop-carried moves)
#ifdef ABL_BAR
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
        block_barrier();
#endif
        CSDR_STAMP(1);                                 // barrier after F1

        // ================= F2: radix-32 forward inside sub-transform sb, column sn =================
        CSDR_PRIO(3);
        {
            // the four points of head group g (network positions 4g..4g+3 <- rows bitrev(4g+q)); three groups
            // ahead of the butterflies (lgkmcnt counts to 15)
            auto fetch = [&](auto Gg) {
                static_for<0, 4>([&](auto Q) {
                    constexpr int p = 4 * Gg.value + Q.value;
                    x[p] = lds_ld8(col + 34 * bitrev<32>(p));
                });
            };
            static_for<0, 3>(fetch);
            CSDR_SB();
            static_for<0, 8>([&](auto Gg) {
                if constexpr (Gg.value + 3 < 8) fetch(std::integral_constant<int, Gg.value + 3>{});
                dit_head4<Gg.value, 32, +1>(x);
                if constexpr ((Gg.value & 1) == 1) CSDR_SB();
            });
            CSDR_STAMP(7);                             // F2 heads
            dit_single<8, 32, +1>(x);
            CSDR_SB();
            // tail group i finishes k1 = i, i+8, i+16, i+24: twiddle, store (one group behind).  H[k] comes from
            // L2 (what is not resident: K1_HREG), at most two loads per tail group (a burst of sixteen held the wave for
            // ~500 cycles of issue alone), in flight from here to the multiply in F3
            v2f tw[2][4];
#ifdef ABL_TW
            static_for<1, 4>([&](auto P) { tw[0][P.value] = twabl; });
#else
            static_for<1, 4>([&](auto P) { tw[0][P.value] = lds_ld8(twc + 32 * (8 * P.value)); });
#endif
            CSDR_SB();
            CSDR_STAMP(8);                             // F2 middle stage
            static_for<0, 9>([&](auto Ii) {
                constexpr int i = Ii.value;
                if constexpr (i < 7)                   // twiddles of the next group
#ifdef ABL_TW
                    static_for<0, 4>([&](auto P) { tw[(i + 1) & 1][P.value] = twabl; });
#else
                    static_for<0, 4>([&](auto P) { tw[(i + 1) & 1][P.value] = lds_ld8(twc + 32 * (i + 1 + 8 * P.value)); });
#endif
                if constexpr (i < 8) {
#ifdef ABL_H
                    hv[2 * i] = habl; hv[2 * i + 1] = habl;
#else
                    if constexpr (2 * i >= HREG) hv[2 * i] = buf_load16(r_h, t * 16, (2 * i) * (T * 16));
                    if constexpr (2 * i + 1 >= HREG) hv[2 * i + 1] = buf_load16(r_h, t * 16, (2 * i + 1) * (T * 16));
#endif
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
                CSDR_SB();
            });
        }
        CSDR_STAMP(2);                                 // F2
        // F2 -> F3 stays inside the half-wave that owns sub-transform sb
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ================= F3 + H + I1: points 32t..32t+31, registers only =================
        CSDR_PRIO(2);
        {
            // forward: network position p <- point m2 = bitrev(p).  Rows q, q+4, q+8, q+12 of 16-byte pairs hold
            // the points {2q, 2q+1} + 8 {0,1,2,3}: exactly the inputs of head groups bitrev3(2q) and bitrev3(2q+1)
            v2f y[32];
            static_for<0, 4>([&](auto Q) {
                static_for<0, 4>([&](auto P) {
                    constexpr int j = Q.value + 4 * P.value;
                    const v4f v = *reinterpret_cast<const v4f *>(rowp + 2 * j);
                    x[bitrev<32>(2 * j)] = v2f{v.x, v.y};
                    x[bitrev<32>(2 * j + 1)] = v2f{v.z, v.w};
                });
            });
            CSDR_SB();
            static_for<0, 4>([&](auto Q) {
                dit_head4<bitrev<8>(2 * Q.value), 32, +1>(x);
                dit_head4<bitrev<8>(2 * Q.value + 1), 32, +1>(x);
                CSDR_SB();
            });
            dit_single<8, 32, +1>(x);
            CSDR_SB();
            // tail group i finishes the bins k2 = i, i+8, i+16, i+24 -- the four inputs (network positions
            // 4g..4g+3, g = bitrev3(i), position 4g + 2 q1 + q0 <- k2 = i + 8 q1 + 16 q0) of the inverse's head
            // group g: multiply by H (folded into that group's first butterflies) and go straight on
            static_for<0, 8>([&](auto Ii) {
                constexpr int i = Ii.value, g = bitrev<8>(i);
                dit_tail<i, 32, +1>(x);
                // times H: the products of the odd inputs ride in the FMA butterflies of the inverse's first stage
                y[4 * g] = x[i]; y[4 * g + 1] = x[i + 16]; y[4 * g + 2] = x[i + 8]; y[4 * g + 3] = x[i + 24];
                dit_head4_tw<g, 32, -1>(y, v2f{hv[2 * i].x, hv[2 * i].y}, v2f{hv[2 * i].z, hv[2 * i].w},
                                        v2f{hv[2 * i + 1].x, hv[2 * i + 1].y}, v2f{hv[2 * i + 1].z, hv[2 * i + 1].w});
                if constexpr ((i & 1) == 1) CSDR_SB();
            });
#pragma unroll
            for (int i = 0; i < 32; i++) x[i] = y[i];
            CSDR_PRIO(1);
            dit_single<8, 32, -1>(x);
            CSDR_SB();
            v4f wv[16];
            static_for<0, 5>([&](auto Q) {
                constexpr int q = Q.value;
                if constexpr (q < 4) {
                    dit_tail<2 * q, 32, -1>(x);
                    dit_tail<2 * q + 1, 32, -1>(x);
                    static_for<0, 4>([&](auto P) {
                        constexpr int j = q + 4 * P.value;
                        wv[j] = store_operand(x[2 * j], x[2 * j + 1]);
                    });
                }
                if constexpr (q > 0) {
                    CSDR_STORE_GROUP_BEGIN();
                    static_for<0, 4>([&](auto P) {
                        constexpr int j = (q - 1) + 4 * P.value;
                        *reinterpret_cast<v4f *>(rowp + 2 * j) = wv[j];
                    });
                    CSDR_STORE_GROUP_END();
                } else {
                    CSDR_SB();
                }
            });
        }
        CSDR_STAMP(3);                                 // F3 + H + I1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ================= I2: conj twiddle, radix-32 DIT inverse =================
        CSDR_PRIO(0);
        {
            v2f tw[32];
            // points and twiddles of head group g, two groups ahead of the butterflies
            auto fetch = [&](auto Gg) {
                static_for<0, 4>([&](auto Q) {
                    constexpr int r = 4 * Gg.value + Q.value;
                    x[r] = lds_ld8(col + 34 * bitrev<32>(r));
                });
                static_for<0, 4>([&](auto Q) {
                    constexpr int r = 4 * Gg.value + Q.value;
#ifdef ABL_TW
                    if constexpr (r != 0) tw[r] = twabl;
#else
                    if constexpr (r != 0) tw[r] = lds_ld8(twc + 32 * bitrev<32>(r));
#endif
                });
            };
            static_for<0, 2>(fetch);
            CSDR_SB();
            static_for<0, 8>([&](auto Gg) {
                constexpr int g = Gg.value;
                if constexpr (g + 2 < 8) fetch(std::integral_constant<int, g + 2>{});
                dit_head4_conjtw<g, 32, -1, g == 0>(x, tw[4 * g], tw[4 * g + 1], tw[4 * g + 2], tw[4 * g + 3]);
                if constexpr ((g & 1) == 1) CSDR_SB();
            });
            dit_single<8, 32, -1>(x);
            CSDR_SB();
            static_for<0, 9>([&](auto I) {
                constexpr int i = I.value;
                if constexpr (i < 8) dit_tail<i, 32, -1>(x);
                if constexpr (i > 0)
                    static_for<0, 4>([&](auto P) {
                        constexpr int n1 = (i - 1) + 8 * P.value;
                        lds_st8(col + 34 * n1, x[n1]);
                    });
                CSDR_SB();
            });
        }
        CSDR_STAMP(4);                                 // I2
#ifdef ABL_BAR
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
        block_barrier();
#endif
        CSDR_STAMP(5);                                 // barrier after I2

        // ================= I3: conj twiddle, radix-16 DIT inverse, store the valid half =================
        CSDR_PRIO(3);
        {
            v2f y[G][R0];
            static_for<0, R0>([&](auto Rr) {
                constexpr int r = Rr.value, k0 = bitrev<R0>(r);
                static_for<0, G / 2>([&](auto PP) {
                    const v4f v = *reinterpret_cast<const v4f *>(outer + OUTER_ROW * k0 + PSTEP_LDS * PP.value);
                    y[2 * PP.value][r] = v2f{v.x, v.y};
                    y[2 * PP.value + 1][r] = v2f{v.z, v.w};
                });
            });
            if constexpr (R0 > 2) {
                if constexpr (!W1_RESIDENT) load_w1();
#pragma unroll
                for (int e = 0; e < G; e++) twiddle_powers<R0>(opaque(w1[e]), pw[e]);     // while the reads are in flight
                CSDR_SB();
                static_for<0, (R0 >= 8 ? R0 / 4 : 1)>([&](auto Gg) {
                    constexpr int g = Gg.value;
                    static_for<0, G>([&](auto E) {
                        constexpr int e = E.value;
                        dit_head4_conjtw<g, R0, -1, g == 0>(y[e], pw[e][bitrev<R0>(4 * g)], pw[e][bitrev<R0>(4 * g + 1)],
                                                            pw[e][bitrev<R0>(4 * g + 2)], pw[e][bitrev<R0>(4 * g + 3)]);
                    });
                    if constexpr ((g & 1) == 1) CSDR_SB();
                });
            }
            // sample 1024*n1 + column, n1 >= HALF  ->  output offset 1024*(n1-HALF) + column
            CSDR_PRIO(2);
            v4f sv[8];                                     // (HALF rows) x (G / 2 column pairs)
            if constexpr (R0 == 16) {
                static_for<0, R0 / 4 + 1>([&](auto I) {
                    constexpr int i = I.value;
                    if constexpr (i < R0 / 4) {
                        dit_tail_upper<i, R0, -1>(y[0]);      // only rows 8..15 of the inverse transform are kept
                        dit_tail_upper<i, R0, -1>(y[1]);
                        sv[2 * i] = store_operand(y[0][i + 8], y[1][i + 8]);
                        sv[2 * i + 1] = store_operand(y[0][i + 12], y[1][i + 12]);
                    }
                    if constexpr (i > 0) {
                        CSDR_STORE_GROUP_BEGIN();
#ifdef ABL_GST
                        keep_alive(sv[2 * (i - 1)]); keep_alive(sv[2 * (i - 1) + 1]);
#else
                        buf_store16_aux<K1_STAUX>(r_out, voff, b * (L * 8) + (i - 1) * 8192, sv[2 * (i - 1)]);
                        buf_store16_aux<K1_STAUX>(r_out, voff, b * (L * 8) + (i - 1 + 4) * 8192, sv[2 * (i - 1) + 1]);
#endif
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            } else if constexpr (R0 == 2) {
                // N = 2048: the kept half is the butterfly's difference output, row 1 = y0 - conj(w) y1; group i = pairs 2i, 2i + 1
                v4f wq[8];
                static_for<0, 8>([&](auto PP) { wq[PP.value] = w_pair(PP.value); });
                static_for<0, 5>([&](auto I) {
                    constexpr int i = I.value;
                    if constexpr (i < 4) {
                        static_for<0, 2>([&](auto Q) {
                            constexpr int pp = 2 * i + Q.value;
                            const v2f d0 = y[2 * pp][0] - cmul_conj(y[2 * pp][1], v2f{wq[pp].x, wq[pp].y});
                            const v2f d1 = y[2 * pp + 1][0] - cmul_conj(y[2 * pp + 1][1], v2f{wq[pp].z, wq[pp].w});
                            sv[pp] = store_operand(d0, d1);
                        });
                    }
                    if constexpr (i > 0) {
                        CSDR_STORE_GROUP_BEGIN();
                        static_for<0, 2>([&](auto Q) {
                            constexpr int pp = 2 * (i - 1) + Q.value;
                            buf_store16_aux<K1_STAUX>(r_out, voff + pp * PSTEP_B, b * (L * 8), sv[pp]);
                        });
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            } else {
                // only the upper half of the inverse transform is kept (fastfir.cpp:291-300).  N = 8192: the last stage's
                // difference outputs, rows 4..7, group i = row 4 + i; N = 4096: rows 2, 3 of the head group's radix-4
                // transform, group i = row 2 + i
                constexpr int NG = HALF;
                static_for<0, NG + 1>([&](auto I) {
                    constexpr int i = I.value;
                    if constexpr (i < NG) {
                        if constexpr (R0 == 8)
                            static_for<0, G>([&](auto E) { bfly_dit_lower<4 * i, -1>(y[E.value][i], y[E.value][i + 4]); });
                        static_for<0, G / 2>([&](auto PP) {
                            sv[i * (G / 2) + PP.value] = store_operand(y[2 * PP.value][HALF + i], y[2 * PP.value + 1][HALF + i]);
                        });
                    }
                    if constexpr (i > 0) {
                        CSDR_STORE_GROUP_BEGIN();
                        static_for<0, G / 2>([&](auto PP) {
                            buf_store16_aux<K1_STAUX>(r_out, voff + PP.value * PSTEP_B, b * (L * 8) + (i - 1) * 8192,
                                                      sv[(i - 1) * (G / 2) + PP.value]);
                        });
                        CSDR_STORE_GROUP_END();
                    } else {
                        CSDR_SB();
                    }
                });
            }
        }
        CSDR_STAMP(6);                                 // I3
    };
    // Everything fetched so far (both input halves, the resident part of H) is waited for HERE, once: left to the
    // compiler the wait sits at the loop header ("vmcnt(7) ... vmcnt(0)" in front of F1's first butterflies), where
    // on the back edge the eight youngest vector-memory operations are the output stores of the block just finished.
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0); lgkmcnt / expcnt left alone (gfx9 encoding)
    int b = b0;
    for (; b + 1 < b1; b += 2) {
        one_block(b, hp, hq);
        one_block(b + 1, hq, hp);
    }
    const bool odd_tail = b < b1;                // uniform per workgroup: a run with an odd number of blocks
    if (odd_tail) one_block(b, hp, hq);

#ifdef K1_CYC
    if (a.dbg && t == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dbg) + (long)blockIdx.x * 2;
        o[0] = __builtin_amdgcn_s_memtime() - cyc0_;
        o[1] = __builtin_amdgcn_s_memrealtime() - rt0_;
    }
#endif
#ifdef CSDR_K1_STAMPS
    if (a.dbg && (t & 63) == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.dbg) + ((long)blockIdx.x * (T / 64) + (t >> 6)) * 16;
        for (int i = 0; i < 16; i++) o[i] = acc_[i];
    }
#endif
    // the tail of this call's input is the overlap of the next call (fastfir.cpp:280-300);
    // written to the other half of the ping-pong history so no workgroup can still be reading it
    if (b1 == a.nblocks) {
        const rsrc_t r_hn = make_rsrc(a.hist_next + (long)ch * L, L * 8u);
        v4f sv[8];
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)      // the last new half: in hp after a pair of blocks, in hq after a single one
#pragma unroll
            for (int pp = 0; pp < G / 2; pp++)
                sv[n1 * (G / 2) + pp] = odd_tail ? store_operand(hq[(2 * pp) * HALF + n1], hq[(2 * pp + 1) * HALF + n1])
                                                 : store_operand(hp[(2 * pp) * HALF + n1], hp[(2 * pp + 1) * HALF + n1]);
        CSDR_STORE_GROUP_BEGIN();
#pragma unroll
        for (int n1 = 0; n1 < HALF; n1++)
#pragma unroll
            for (int pp = 0; pp < G / 2; pp++) buf_store16(r_hn, voff + pp * PSTEP_B, n1 * 8192, sv[n1 * (G / 2) + pp]);
        CSDR_STORE_GROUP_END();
    }
}

template <int LOG2N>
static hipError_t launch2_one(const FastFirArgs &a, hipStream_t stream)
{
    using Cfg = K1Cfg<LOG2N>;
    // once per device and size (the attribute belongs to the device, and a process may drive several): the per-launch
    // call cost the per-datagram host form microseconds
    hipError_t e = CSDR_MAX_LDS_ONCE(&fastfir_os2_kernel<LOG2N>, Cfg::LDS_BYTES);
    if (e != hipSuccess) return e;
#ifdef CSDR_WG_TRACE
    FastFirArgs b = a;
    b.trace = wgtrace_next();
    hipLaunchKernelGGL((fastfir_os2_kernel<LOG2N>), dim3((a.channels * a.runs + Cfg::VW - 1) / Cfg::VW), dim3(Cfg::T),
                       Cfg::LDS_BYTES, stream, b);
    return hipGetLastError();
#endif
    hipLaunchKernelGGL((fastfir_os2_kernel<LOG2N>), dim3((a.channels * a.runs + Cfg::VW - 1) / Cfg::VW), dim3(Cfg::T),
                       Cfg::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t fastfir2_launch(int log2n, const FastFirArgs &a, hipStream_t stream)
{
    switch (log2n) {
    case 11: return launch2_one<11>(a, stream);
    case 12: return launch2_one<12>(a, stream);
    case 13: return launch2_one<13>(a, stream);
    case 14: return launch2_one<14>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// Host mirror of the kernel's index algebra: thread t of pass F3 owns k0 = t >> 5 (sub-transform) and k1 = t & 31
// (its row), and consumes H in the order its tail groups finish bins: float4 j = 2 i + h of thread t, half e,
// multiplies k2 = i + 8 h + 16 e; natural bin k = k0 + R0 (k1 + 32 k2), R0 = N / 1024 sub-transforms.
int fastfir2_bin_of(int log2n, int t, int j, int e)
{
    const int R0 = (1 << log2n) / 1024;
    const int i = j >> 1, h = j & 1;
    const int k2 = i + 8 * h + 16 * e;
    return (t >> 5) + R0 * ((t & 31) + 32 * k2);
}

}  // namespace csdr
