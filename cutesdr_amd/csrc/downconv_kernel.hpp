// downconv_kernel.hpp -- NCO mixer + decimate-by-2^n cascade for gfx950 (K2 in DESIGN.md).
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
#pragma once
#include "fft_core.hpp"
#include "downconv_kernels.h"
#include "../../include/csdr_hb_taps.h"

namespace csdr {

constexpr int DC_T = 64;                  // threads per workgroup
constexpr int DC_ROW = 2 * DC_T;           // samples per row (16 B per lane)
constexpr int DC_TILE = 512;              // input samples per tile
constexpr int DC_ANCHOR_ROWS = 16;
typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));   // a 16-byte load from an 8-byte aligned address

// ---- the cascade as a compile-time plan ----------------------------------------------------------------
// SetDataRate's stage sequences are few in practice (CIC-3s, then 11-tap half bands, then one to three longer
// ones: tools/list_dc_plans.py), and a kernel that knows its sequence needs no stage dispatch, no layout words
// in registers, no lane predicates in the passes of a complete tile.  DcPlanT<kinds...> instantiates the kernel
// for one sequence (cutesdr_amd/_build.py compiles downconv_plan.hip once per listed plan); DcPlanDyn is the
// same kernel with everything taken at run time, for every other sequence.
template <int... K> struct DcPlanT {
    static constexpr int NS = sizeof...(K);
    static constexpr int KIND[sizeof...(K) + 1] = {K..., 0};
};
struct DcPlanDyn {
    static constexpr int NS = -1;
    static constexpr int KIND[1] = {0};
};
constexpr int dc_hist_of(int kind) { return kind == 3 ? 2 : kind - 1; }
// pair coefficient q and centre coefficient of a stage kind, as dc_host.hpp's dc_make_plan stores them (fp32)
constexpr float dc_pair_coef(int kind, int q) { return kind == 3 ? (q == 0 ? 0.125f : 0.375f) : (float)csdr_hb_even[(kind - 11) / 4][q]; }
constexpr float dc_centre_coef(int kind) { return kind == 3 ? 0.f : 0.5f; }

// LDS image: region s = [even half | odd half] of xe_s = [history | stage-s input], region ns = tile outputs
struct DcLayout {
    int roff[DC_MAX_STAGES + 2];
    int ooff[DC_MAX_STAGES + 1];
    int slots;
};
constexpr DcLayout dc_layout_of(const int *kind, int ns)
{
    DcLayout l{};
    int o = 0;
    for (int s = 0; s <= ns; s++) {
        l.roff[s] = o;
        if (s < ns) {
            const int half = (dc_hist_of(kind[s]) / 2 + (DC_TILE >> (s + 1)) + 2) & ~1;    // + slack for the CIC's O[j+1]
            l.ooff[s] = half;
            o += 2 * half;
        } else {
            l.ooff[s] = 0;
            o += ((DC_TILE >> s) + 1) & ~1;
        }
    }
    l.slots = o;
    return l;
}

// The workgroup is ONE wave and the LDS unit executes a wave's instructions in issue order: a ds_read issued after a
// ds_write of the same wave sees the written data, whichever lane wrote it.  The hand-over between the steps of a
// tile therefore needs neither an s_waitcnt (it would make every step wait out the LDS write latency) nor an
// s_barrier; only the compiler must keep the program order of the LDS accesses.  (A wavefront-scope fence would
// do that too, but it also orders -- and so drains -- the global loads prefetched for the next tile.)
#ifdef CSDR_DC_HARD_BARRIER
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
#endif
static_assert(DC_T == 64, "lds_barrier() relies on a single-wave workgroup");

// the raw samples are read once: non-temporal loads keep them from displacing what the chip re-reads (K1 measured
// -1.2 % for the same change; here A/B'd with tools/bench_k2_plans.py)
__device__ __forceinline__ v4f dc_stream_load(const v4f *p)
{
#ifdef CSDR_NO_NT
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}

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
//
// The stage input xe = [history | input] is kept split by sample parity, E[m] = xe[2m] and
// O[m] = xe[2m+1], so that output j reads E[j+q] / O[j+q]: consecutive lanes touch consecutive
// LDS words (the interleaved layout made every tap read a 16-byte-stride access).  Outputs go to
// the next stage's halves (its history is even, so output j has parity j&1) or, after the last
// stage, to the linear tile-output region.
// per-stage parameters, staged in LDS once per workgroup: the tap values are read from there (wide
// uniform reads), the layout words sit packed in two registers with stage s in lane s (v_readlane)
enum { DP_KIND = 0, DP_HIST2, DP_ROFF, DP_OOFF, DP_CC, DP_C0 = 8, DP_WORDS = 24 };   // rows of 96 B, taps 16-B aligned

// The arithmetic of one output, the same in every form of the stage (run-time plan, compiled plan, one or several
// outputs per lane): first product rounded on its own, then acc + (a + b) * c as one fused multiply-add per tap pair.
// Left to the compiler, "o * cc + (a + b) * c0" contracts into a fused multiply-add around EITHER product, depending on
// the code around it -- one-ulp differences between the forms; dc_first() keeps the first product out of that choice.
__device__ __forceinline__ v2f dc_first(v2f x, float c)
{
    v2f p = x * c;
    asm volatile("" : "+v"(p));
    return p;
}
__device__ __forceinline__ v2f dc_mac(v2f acc, v2f a, v2f b, float c) { return acc + (a + b) * c; }
__device__ __forceinline__ v2f dc_cic(v2f e0, v2f o0, v2f e1, v2f o1, float c0, float c1)    // downconvert.cpp:453-454
{
    return dc_mac(dc_first(o0 + e1, c1), e0, o1, c0);
}

template <int L>
__device__ __forceinline__ void dc_stage(const v2f *E, const v2f *O, v2f *yE, v2f *yO, v2f *ylin, int nout,
                                         const int *prm, int t)
{
    constexpr int NP = (L == 3) ? 2 : (L + 1) / 4;
    constexpr int H = (L - 1) / 2;              // centre tap (odd index for every L = 4k+3)
    constexpr int UN = 1;                       // outputs per thread and pass: all reads before any write
    float c[NP];
#pragma unroll
    for (int q = 0; q < NP; q += 4) {
        const v4f v = *reinterpret_cast<const v4f *>(prm + DP_C0 + q);
        c[q] = v.x;
        if (q + 1 < NP) c[q + 1] = v.y;
        if (q + 2 < NP) c[q + 2] = v.z;
        if (q + 3 < NP) c[q + 3] = v.w;
    }
    const float cc = __int_as_float(prm[DP_CC]);
    for (int j0 = t; j0 < nout; j0 += UN * DC_T) {
        v2f acc[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int j = j0 + u * DC_T;
            if (j < nout) {
                if (L == 3) {
                    acc[u] = dc_cic(E[j], O[j], E[j + 1], O[j + 1], c[0], c[1]);
                } else {
                    acc[u] = dc_first(O[j + (H - 1) / 2], cc);
#pragma unroll
                    for (int q = 0; q < NP; q++) acc[u] = dc_mac(acc[u], E[j + q], E[j + H - q], c[q]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int j = j0 + u * DC_T;
            if (j < nout) {
                if (ylin) ylin[j] = acc[u];
                else if (j & 1) yO[j >> 1] = acc[u];
                else yE[j >> 1] = acc[u];
            }
        }
    }
}

// The same stage for a complete tile of a compile-time plan: the output count is a constant, whole passes carry
// no lane predicate and every LDS offset is an immediate.
// A lane computes G ADJACENT outputs from one register window (round 3): outputs j .. j+G-1 read E[j .. j+H+G-1]
// and G consecutive odd-half samples, as 16-byte LDS reads -- (H+G)/2 + G/2 wide reads instead of (H+2)*G narrow
// ones.  The down-converter keeps the LDS pipe busy three quarters of its time (SQ_LDS_IDX_ACTIVE 74 %, DESIGN K2),
// and the first two stages are two thirds of that traffic: 56 -> 40 B per output of an 11-tap stage at G = 2.  The
// arithmetic per output -- centre product, then the pairs in tap order -- is what dc_stage does, so the words are too.
#ifndef DC_STAGE_G4_MIN
#define DC_STAGE_G4_MIN (1 << 30)         // four outputs per lane: measured no better than two (more registers), off
#endif
#ifndef DC_STAGE_G2_MIN
#define DC_STAGE_G2_MIN 2
#endif
#ifndef DC_STAGE_G2_MAXL
#define DC_STAGE_G2_MAXL 19
#endif
template <int L, int NOUT>
__device__ __forceinline__ void dc_stage_full(const v2f *E, const v2f *O, v2f *yE, v2f *yO, bool last, int t)
{
    constexpr int NP = (L == 3) ? 2 : (L + 1) / 4;
    constexpr int H = (L - 1) / 2;
    float c[NP];                                 // the taps are literals here: no table read, no registers
#pragma unroll
    for (int q = 0; q < NP; q++) c[q] = dc_pair_coef(L, q);
    constexpr float cc = dc_centre_coef(L);
    // (long half bands sit at the end of a cascade, a few outputs per tile: their window would cost more registers --
    // spills at 128 -- than the LDS bytes it saves)
    constexpr int G = NOUT >= DC_STAGE_G4_MIN ? 4 : ((NOUT >= DC_STAGE_G2_MIN && L <= DC_STAGE_G2_MAXL) ? 2 : 1);
    if constexpr (G == 1) {
        constexpr int PASSES = NOUT >= DC_T ? NOUT / DC_T : 1;
        v2f r[PASSES];
        if (NOUT >= DC_T || t < NOUT) {
#pragma unroll
            for (int k = 0; k < PASSES; k++) {
                const v2f *e = E + t + k * DC_T, *o = O + t + k * DC_T;
                if (L == 3) r[k] = dc_cic(e[0], o[0], e[1], o[1], c[0], c[1]);
                else {
                    v2f acc = dc_first(o[(H - 1) / 2], cc);
#pragma unroll
                    for (int q = 0; q < NP; q++) acc = dc_mac(acc, e[q], e[H - q], c[q]);
                    r[k] = acc;
                }
            }
            v2f *y = last ? yE + t : ((t & 1) ? yO : yE) + (t >> 1);
#pragma unroll
            for (int k = 0; k < PASSES; k++) y[k * (last ? DC_T : DC_T / 2)] = r[k];
        }
    } else {
        constexpr int PASSES = NOUT >= G * DC_T ? NOUT / (G * DC_T) : 1;
        constexpr int NWIN = (L == 3 ? 1 : H) + G;                    // E[j .. j+NWIN-1]
        constexpr int NB = (NWIN + 1) / 2;
        constexpr int C0 = (L == 3) ? 0 : (H - 1) / 2;                // first odd-half sample: O[j + C0]
        constexpr int OB0 = C0 & ~1, NOB = (C0 - OB0 + (L == 3 ? G + 1 : G) + 1) / 2;
        // Alignment and extent of the 16-byte window reads.  A region starts at an even slot (dc_layout_of rounds every
        // half to an even length), j is a multiple of G and OB0 is even: E + j + 2i and O + j + OB0 + 2i are 16-byte
        // aligned whatever the history's parity (only the OUTPUT halves, which start behind hist/2 slots, are 8-byte
        // aligned: they are stored as v2f unless `last`).  An odd window is read one sample long; that sample and the
        // whole last window lie inside the half, whose length is (hist/2 + NOUT + 2) & ~1:
        constexpr int HALF = (dc_hist_of(L) / 2 + NOUT + 2) & ~1;
        static_assert((NOUT >= G ? NOUT - G : 0) + 2 * NB <= HALF, "even-half window reads stay inside the stage's LDS half");
        static_assert((NOUT >= G ? NOUT - G : 0) + OB0 + 2 * NOB <= HALF, "odd-half window reads stay inside the stage's LDS half");
        v2f r[PASSES][G];
        if (NOUT >= G * DC_T || G * t < NOUT) {
#pragma unroll
            for (int k = 0; k < PASSES; k++) {
                const int j = G * t + G * DC_T * k;
                v2f w[2 * NB], o[2 * NOB];
#pragma unroll
                for (int i = 0; i < NB; i++) {
                    const v4f v = *reinterpret_cast<const v4f *>(E + j + 2 * i);
                    w[2 * i] = v2f{v.x, v.y}; w[2 * i + 1] = v2f{v.z, v.w};
                }
#pragma unroll
                for (int i = 0; i < NOB; i++) {
                    const v4f v = *reinterpret_cast<const v4f *>(O + j + OB0 + 2 * i);
                    o[2 * i] = v2f{v.x, v.y}; o[2 * i + 1] = v2f{v.z, v.w};
                }
#pragma unroll
                for (int u = 0; u < G; u++) {
                    if (L == 3) {
                        r[k][u] = dc_cic(w[u], o[u], w[u + 1], o[u + 1], c[0], c[1]);
                    } else {
                        v2f acc = dc_first(o[C0 - OB0 + u], cc);
#pragma unroll
                        for (int q = 0; q < NP; q++) acc = dc_mac(acc, w[u + q], w[u + H - q], c[q]);
                        r[k][u] = acc;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < PASSES; k++) {
                const int j = G * t + G * DC_T * k;
                if (last) {
#pragma unroll
                    for (int u = 0; u < G; u += 2)
                        *reinterpret_cast<v4f *>(yE + j + u) = v4f{r[k][u].x, r[k][u].y, r[k][u + 1].x, r[k][u + 1].y};
                } else if constexpr (G == 4) {                        // (the halves start behind an odd history: 8-byte aligned)
                    yE[(j >> 1)] = r[k][0]; yE[(j >> 1) + 1] = r[k][2];
                    yO[(j >> 1)] = r[k][1]; yO[(j >> 1) + 1] = r[k][3];
                } else {
                    yE[j >> 1] = r[k][0];
                    yO[j >> 1] = r[k][1];
                }
            }
        }
    }
}

// value of lane + 1 (lane 63: lane 0's): the DPP wavefront rotate, no LDS
__device__ __forceinline__ float dc_rotl1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, false));   // wave_rol:1
}

// BLK: the noise blanker's decision applied in this kernel's loads (DcArgs::nb_mask; a kernel of its own so that the
// plain kernel keeps its registers: the blanked form carries the mask words and one more sample per tile)
#ifndef CSDR_DC_WAVES_PER_EU
#define CSDR_DC_WAVES_PER_EU 4
#endif
template <class P, bool BLK = false>
__global__ __launch_bounds__(DC_T) __attribute__((amdgpu_waves_per_eu(CSDR_DC_WAVES_PER_EU, CSDR_DC_WAVES_PER_EU)))
void downconv_kernel(DcArgs a)
{
    constexpr bool FIX = P::NS >= 0;                      // compile-time plan
    constexpr DcLayout LY = dc_layout_of(P::KIND, FIX ? P::NS : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *lds = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x;
    const int ns = FIX ? P::NS : a.nstages;
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_DC);
    const int wg = blockIdx.x;
    const int ci = wg / a.nseg, seg = wg % a.nseg;
    if (ci >= a.nchan) return;
    const int ch = a.chan_list ? a.chan_list[ci] : ci;
    const DcChan cs = a.chan[ch];

    // LDS regions R_s = [hist_s | stage-s input] at a.roff[s] (host computed); R_ns = tile outputs
    const int *roff = FIX ? LY.roff : a.roff;
    __shared__ __attribute__((aligned(16))) int ptab[DC_MAX_STAGES + 1][DP_WORDS];
    for (int i = t; i < (ns + 1) * DP_WORDS; i += DC_T) {
        const int s = i / DP_WORDS, k = i % DP_WORDS;
        int v = 0;
        if (k == DP_ROFF) v = roff[s];
        else if (k == DP_OOFF) v = a.ooff[s];
        else if (s < ns) {
            if (k == DP_KIND) v = a.kind[s];
            else if (k == DP_HIST2) v = a.st[s].hist / 2;
            else if (k == DP_CC) v = __float_as_int(a.st[s].ccoef);
            else if (k >= DP_C0 && k - DP_C0 < DC_MAX_PAIRS) v = __float_as_int(a.st[s].c[k - DP_C0]);
        }
        ptab[s][k] = v;
    }
    for (int s = 0; s < ns; s++)
        for (int i = t; i < a.st[s].hist; i += DC_T)
            lds[roff[s] + ((i & 1) ? a.ooff[s] : 0) + (i >> 1)] = v2f{0.f, 0.f};
    lds_barrier();
    // lane s: kind | hist/2 << 8 | odd-half offset << 16, and the region offset, of stage s
    int pv0, pv1;
    {
        const int *row = ptab[(t & 63) <= ns ? (t & 63) : ns];
        pv0 = row[DP_KIND] | (row[DP_HIST2] << 8) | (row[DP_OOFF] << 16);
        pv1 = row[DP_ROFF];
    }
    const v2f step1 = phasor_of(cs.inc);                 // one sample of NCO rotation
    v2f p0 = {1.f, 0.f}, p1 = {1.f, 0.f};
    int anchor_rows = 0;                                 // rows until the phasors are re-anchored

    const long in_row = a.in_rows ? a.in_rows[ch] : ch;
    const v2f *in = a.in + in_row * a.in_stride;
    const unsigned char *pk = a.wire.pk ? a.wire.pk + in_row * a.wire.chan_stride : nullptr;   // datagram input
    // blanker in front: delay D1 = delay_n + 1 (0 when the row's blanker is off: its mask is all zero), mask row, history
    int D1 = 0;
    const unsigned *mrow = nullptr;
    const v2f *bhist = nullptr;
    if constexpr (BLK) {
        const NbChan nb = a.nb_state[in_row];
        D1 = nb.on ? nb.delay_n + 1 : 0;
        mrow = a.nb_mask + in_row * a.nb_mask_stride;
        bhist = a.nb_hist + in_row * NB_HIST;
    }
    // one raw input sample of the stream the blanker saw: this call's, or the history's for j < 0 (BLK only)
    auto raw_at = [&](long j) -> v2f {
        if (j < 0) return bhist[NB_HIST + j];
        if (pk) { const wf2 w = wire_sample(pk, a.wire.pkt_len, j); return v2f{w.x, w.y}; }
        return in[j];
    };
    // samples idx, idx + 1 (idx even) as the blanker hands them over: delayed by D1, zero under the mask
    auto blanked_pair = [&](long idx) -> v4f {
        v2f s0 = raw_at(idx - D1), s1 = raw_at(idx + 1 - D1);
        const unsigned m = mrow[idx >> 5] >> (idx & 31);
        if (m & 1u) s0 = v2f{0.f, 0.f};
        if (m & 2u) s1 = v2f{0.f, 0.f};
        return v4f{s0.x, s0.y, s1.x, s1.y};
    };
    v2f *out = a.out + (long)ch * a.out_stride;
    const v2f *hist = a.hist + (long)ch * a.hist_stride;
    v2f *hist_next = a.hist_next + (long)ch * a.hist_stride;
    const long seg_start = (long)seg * a.seg_len;
    long seg_end = seg_start + a.seg_len;
    if (seg_end > a.n_in) seg_end = a.n_in;
    const v2f rowstep = phasor_of(cs.inc * (unsigned long long)DC_ROW);
    const float a_inf = a.amp[DC_AMP_N - 1], inv_a_inf = 1.0f / a_inf;
    // Re-anchoring the phasors (exact phase from the 64-bit accumulator, then one complex multiply per row) on an
    // ABSOLUTE grid -- every DC_ANCHOR_ROWS rows counted from the channel's first sample, not from wherever this
    // workgroup's segment or this call happens to start: a tile that starts between two grid points anchors at the
    // grid point behind it and walks the rows in between with the same multiplies a continuous run would have made.
    // The mixed samples, hence the output WORDS, then do not depend on how the stream is cut into calls and segments
    // (for calls that are whole tiles: a short tile breaks the row cadence and anchors where it stands, as before).
    auto anchor = [&](long at, bool full_tile) {
        int back = 0;                                    // whole tiles between the grid point and `at`
        if (full_tile) {
            const unsigned long long ab = cs.age + (unsigned long long)at;
            if ((ab & (unsigned long long)(DC_TILE - 1)) == 0) back = (int)((ab / DC_TILE) & (unsigned long long)(DC_ANCHOR_ROWS / (DC_TILE / DC_ROW) - 1));
        }
        p0 = phasor_of(cs.phase + cs.inc * (unsigned long long)(at - (long)back * DC_TILE + 2 * t + 1)) * a_inf;
        p1 = cmul(p0, step1);
        for (int r = 0; r < back * (DC_TILE / DC_ROW); r++) { p0 = cmul(p0, rowstep); p1 = cmul(p1, rowstep); }
        anchor_rows = full_tile ? DC_ANCHOR_ROWS - back * (DC_TILE / DC_ROW) : 0;
    };

    // The raw input of a tile is fetched into registers one tile ahead, while the previous tile goes
    // through the cascade: nothing waits on HBM latency except the very first tile.
    constexpr int NR = DC_TILE / DC_ROW;
    v4f raw[NR];
    // BLK: the tile's mask words (a lane's two samples of a row share one) and, for datagram input behind an ODD delay,
    // the raw words of the one sample behind the tile's last pair (see the steady loop)
    unsigned mkw = 0u;                                   // lane l & 15 holds word l & 15 of the tile's sixteen mask words
    unsigned xw[3] = {0u, 0u, 0u};
    // BLK: the prefetch of a COMPLETE tile at p whose delayed samples all lie in this call's input (the steady loop's
    // condition); other tiles are read where they are consumed (blanked_pair)
    auto fetch_blk = [&](long p, auto FMT) {
        if constexpr (BLK) {
            constexpr int fmt = FMT.value;                   // 0: float rows, otherwise the datagram length
            if (!(p - D1 - 1 >= 0 && p + DC_TILE <= seg_end)) return;
            const bool odd = D1 & 1;
            mkw = mrow[(p >> 5) + (t & 15)];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const long i = p + r * DC_ROW + 2 * t;
                if constexpr (fmt != 0) {
                    const wf4 w = wire_pair_fetch(pk, fmt, i - D1 - (odd ? 1 : 0));               // an aligned pair
                    raw[r] = v4f{w.x, w.y, w.z, w.w};
                } else {
                    raw[r] = *reinterpret_cast<const f4a8 *>(in + i - D1);                        // 8-byte aligned, 16 bytes
                }
            }
            if constexpr (fmt != 0) {
                if (odd) {                                      // the sample behind the tile's last aligned pair
                    const unsigned j = (unsigned)(p + DC_TILE - 1 - D1);
                    if constexpr (fmt == 1444) {
                        const unsigned q = j / 240u, jj = j - q * 240u;
                        const unsigned short *h = reinterpret_cast<const unsigned short *>(pk + (q * 1444u + 4u + 6u * jj));
                        xw[0] = h[0]; xw[1] = h[1]; xw[2] = h[2];
                    } else {
                        xw[0] = *reinterpret_cast<const unsigned *>(pk + ((j >> 8) * 1028u + 4u + 4u * (j & 255u)));
                    }
                }
            }
        }
    };
    auto fetch_blk_any = [&](long p) {
        if (!pk) fetch_blk(p, std::integral_constant<int, 0>{});
        else if (a.wire.pkt_len == 1444) fetch_blk(p, std::integral_constant<int, 1444>{});
        else fetch_blk(p, std::integral_constant<int, 1028>{});
    };
    auto tile_len = [&](long p) {
        const long lim = ((p < seg_start) ? seg_start : seg_end) - p;
        return (int)(lim < DC_TILE ? lim : DC_TILE);
    };
    auto fetch = [&](long p) {
        if (p >= seg_end || (p < seg_start && seg == 0)) return;      // past the end / history-fed warm-up
        if constexpr (BLK) { fetch_blk_any(p); return; }
        const int m = tile_len(p);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int i = r * DC_ROW + 2 * t;
            if (i < m) {
                if (pk) {
                    const wf4 w = wire_pair_fetch(pk, a.wire.pkt_len, p + i);          // raw words; p + i is even
                    raw[r] = v4f{w.x, w.y, w.z, w.w};
                } else {
                    raw[r] = dc_stream_load(reinterpret_cast<const v4f *>(in + p + i));
                }
            }
        }
    };

    // datagram input: the prefetched words are decoded where they are consumed
    auto unwire = [&](v4f r) -> v4f {
        if (!pk) return r;
        const wf4 w = wire_pair_decode(wf4{r.x, r.y, r.z, r.w}, a.wire.pkt_len);
        return v4f{w.x, w.y, w.z, w.w};
    };
    // pos: index of the tile's first sample in this call's input (negative inside the warm-up)
    long pos = seg_start - a.W;
#ifdef DC_PROFILE
    unsigned long long tk[20] = {0}, tlast = __builtin_readcyclecounter();
#define DC_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define DC_TICK(k)
#endif
    fetch(pos);
    while (pos < seg_end) {
        if constexpr (FIX && P::NS > 0) {
            // ---- steady state of a compile-time plan: complete tiles mixed from this call's input, the NCO
            // amplitude settled, nothing to save for the next call.  No tile length, no path selection, no lane
            // predicate; the other tiles (history-fed warm-up, the last W samples, a segment's odd end) take the
            // general body below, which keeps the same one-tile-ahead prefetch protocol.
            const long hi = a.W ? (seg_end < (long)a.n_in - a.W ? seg_end : (long)a.n_in - a.W) : seg_end;
            // one copy of the loop per input format (float rows, 16-bit and 24-bit datagrams): the datagram length is a
            // constant inside each, so the decode carries no format test and the sample-index division is a multiply
            auto steady = [&](auto FMT) {
            constexpr int fmt = FMT.value;               // 0: float rows, otherwise the datagram length
            while (true) {
                const bool w = pos < seg_start;
                const long lim = w ? seg_start : hi;
                if (!(pos + DC_TILE <= lim && (seg > 0 || pos >= 0) && cs.age + (unsigned long long)pos >= DC_AMP_N)) break;
                if constexpr (BLK) { if (pos - D1 - 1 < 0) break; }     // the delayed samples reach into the history: general tile
                if (anchor_rows <= 0) anchor(pos, true);
                anchor_rows -= NR;
                {
                    v2f *e = lds + LY.roff[0] + dc_hist_of(P::KIND[0]) / 2 + t, *o = e + LY.ooff[0];
                    bool any_blank = false;
                    if constexpr (BLK) any_blank = __any(mkw != 0u);
                    v4f dv[NR];
#pragma unroll
                    for (int row = 0; row < NR; row++) {
                        dv[row] = raw[row];
                        if constexpr (fmt != 0) {
                            const wf4 d = wire_pair_decode(wf4{dv[row].x, dv[row].y, dv[row].z, dv[row].w}, fmt);
                            dv[row] = v4f{d.x, d.y, d.z, d.w};
                        }
                    }
                    if constexpr (BLK && fmt != 0) {
                        // datagrams behind an ODD delay: the lane fetched the aligned pair (e, e+1) and wants (e+1, e+2);
                        // e+2 is the first sample of the next lane's pair (lane 63: of the next row's lane 0, and behind
                        // the tile's last pair the one sample fetched into xw) -- a wavefront rotate, no LDS
                        if (D1 & 1) {
                            v2f xs;
                            if constexpr (fmt == 1444) {
                                const int vi = (int)((xw[0] << 8) | ((xw[1] & 0xffu) << 24)), vq = (int)(((xw[1] >> 8) << 8) | (xw[2] << 16));
                                xs = v2f{(float)vi * (1.0f / 65536.0f), (float)vq * (1.0f / 65536.0f)};
                            } else {
                                xs = v2f{(float)(short)(xw[0] & 0xffffu), (float)(short)(xw[0] >> 16)};
                            }
#pragma unroll
                            for (int row = 0; row < NR; row++) {
                                const float ax = dc_rotl1(dv[row].x), ay = dc_rotl1(dv[row].y);
                                const float bx = row + 1 < NR ? dc_rotl1(dv[row + 1 < NR ? row + 1 : row].x) : xs.x;
                                const float by = row + 1 < NR ? dc_rotl1(dv[row + 1 < NR ? row + 1 : row].y) : xs.y;
                                const bool lastlane = t == DC_T - 1;
                                raw[row] = v4f{dv[row].z, dv[row].w, lastlane ? bx : ax, lastlane ? by : ay};
                            }
#pragma unroll
                            for (int row = 0; row < NR; row++) dv[row] = raw[row];
                        }
                    }
#pragma unroll
                    for (int row = 0; row < NR; row++) {
                        v4f v = dv[row];
                        if constexpr (BLK) {
                            if (any_blank) {                                   // (rare: a tile with an impulse in it)
                                // word 4 row + (t >> 4) of the tile's sixteen; tiles start at multiples of 512: (i & 31) = 2t & 31
                                const unsigned m = (unsigned)__shfl((int)mkw, 4 * row + (t >> 4)) >> ((2 * t) & 31);
                                if (m & 1u) { v.x = 0.f; v.y = 0.f; }
                                if (m & 2u) { v.z = 0.f; v.w = 0.f; }
                            }
                        }
                        e[row * DC_T] = cmul(v2f{v.x, v.y}, p0);
                        o[row * DC_T] = cmul(v2f{v.z, v.w}, p1);
                        p0 = cmul(p0, rowstep);
                        p1 = cmul(p1, rowstep);
                    }
                }
                lds_barrier();
                if constexpr (BLK) {
                    fetch_blk(pos + DC_TILE, FMT);
                } else if (pos + 2 * DC_TILE <= lim) {          // the next tile is complete too: all rows, no bounds
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        const long i = pos + DC_TILE + r * DC_ROW + 2 * t;
                        if constexpr (fmt != 0) {
                            const wf4 wv = wire_pair_fetch(pk, fmt, i);
                            raw[r] = v4f{wv.x, wv.y, wv.z, wv.w};
                        } else {
                            raw[r] = dc_stream_load(reinterpret_cast<const v4f *>(in + i));
                        }
                    }
                } else {
                    fetch(pos + DC_TILE);
                }
                static_for<0, P::NS>([&](auto S) {
                    constexpr int s = S.value, L = P::KIND[s];
                    constexpr bool last = s + 1 == P::NS;
                    const v2f *E = lds + LY.roff[s], *O = E + LY.ooff[s];
                    v2f *yE = lds + LY.roff[s + 1] + (last ? 0 : dc_hist_of(P::KIND[s + 1]) / 2), *yO = yE + LY.ooff[s + 1];
                    dc_stage_full<L, ((DC_TILE / 2) >> s)>(E, O, yE, yO, last, t);
                    lds_barrier();
                });
                {
                    v2f keep[P::NS];
                    const int i = t & 31, odd = t >> 5;
                    static_for<0, P::NS>([&](auto S) {
                        constexpr int sg = S.value;
                        if (i < dc_hist_of(P::KIND[sg]) / 2) keep[sg] = lds[LY.roff[sg] + (odd ? LY.ooff[sg] : 0) + i + (DC_TILE >> (sg + 1))];
                    });
                    lds_barrier();
                    static_for<0, P::NS>([&](auto S) {
                        constexpr int sg = S.value;
                        if (i < dc_hist_of(P::KIND[sg]) / 2) lds[LY.roff[sg] + (odd ? LY.ooff[sg] : 0) + i] = keep[sg];
                    });
                }
                if (!w) {
                    constexpr int LEN = DC_TILE >> P::NS;
                    const v2f *y = lds + LY.roff[P::NS];
                    const long obase = pos >> P::NS;
#pragma unroll
                    for (int j = 0; j < (LEN + DC_T - 1) / DC_T; j++)
                        if (LEN >= DC_T || t < LEN) out[obase + j * DC_T + t] = y[j * DC_T + t];
                }
                lds_barrier();
                pos += DC_TILE;
            }
            };
            if (!pk) steady(std::integral_constant<int, 0>{});
            else if (a.wire.pkt_len == 1444) steady(std::integral_constant<int, 1444>{});
            else steady(std::integral_constant<int, 1028>{});
            if (pos >= seg_end) break;
        }
        const bool warm = pos < seg_start;
        const int n = tile_len(pos);
        // stage-0 input: parity halves behind their histories (no decimation: the linear output region)
        const int h0 = ns > 0 ? (FIX ? dc_hist_of(P::KIND[0]) : a.st[0].hist) / 2 : 0;
        v2f *r0 = lds + roff[0] + h0, *r0o = lds + roff[0] + (ns > 0 ? (FIX ? LY.ooff[0] : a.ooff[0]) : 0) + h0;
        // ---------------- stage-0 input: mix with the NCO (or take the mixed history) -----------
        if (warm && seg == 0) {
            for (int i = t; i < n; i += DC_T) {
                const v2f v = hist[pos + a.W + i];
                if (ns == 0) r0[i] = v;
                else if (i & 1) r0o[i >> 1] = v;
                else r0[i >> 1] = v;
            }
        } else {
            // the phasors run on from tile to tile and are re-anchored every DC_ANCHOR_ROWS rows
            // (and after a short tile, which breaks the row cadence)
            // p0/p1 carry the steady-state amplitude a_inf
            if (anchor_rows <= 0 || n != DC_TILE) anchor(pos, n == DC_TILE);
            anchor_rows -= NR;
            // full tile, amplitude settled, no history to save: the lean path
            const bool lean = ns > 0 && n == DC_TILE && cs.age + (unsigned long long)pos >= DC_AMP_N &&
                              (warm || a.W == 0 || pos + n <= a.n_in - a.W);
            if (lean) {
                v2f *e = r0 + t, *o = r0o + t;
#pragma unroll
                for (int row = 0; row < NR; row++) {
                    v4f v;
                    if constexpr (BLK) v = blanked_pair(pos + row * DC_ROW + 2 * t); else v = unwire(raw[row]);
                    e[row * DC_T] = cmul(v2f{v.x, v.y}, p0);
                    o[row * DC_T] = cmul(v2f{v.z, v.w}, p1);
                    p0 = cmul(p0, rowstep);
                    p1 = cmul(p1, rowstep);
                }
            } else
#pragma unroll
            for (int row = 0; row < NR; row++) {
                const int i = row * DC_ROW + 2 * t;
                const long gi = pos + i;                       // sample index within the call
                if (i < n) {
                    v4f v;
                    if constexpr (BLK) v = blanked_pair(gi); else v = unwire(raw[row]);
                    v2f x0 = cmul(v2f{v.x, v.y}, p0), x1 = cmul(v2f{v.z, v.w}, p1);
                    const unsigned long long age = cs.age + (unsigned long long)gi;
                    if (age + 1 < DC_AMP_N) {                   // start-up envelope (the phasors carry a_inf)
                        x0 *= a.amp[age] * inv_a_inf; x1 *= a.amp[age + 1] * inv_a_inf;
                    }
                    if (ns == 0) {
                        *reinterpret_cast<v4f *>(&r0[i]) = v4f{x0.x, x0.y, x1.x, x1.y};
                    } else {
                        r0[i >> 1] = x0; r0o[i >> 1] = x1;
                    }
                    // the last W mixed samples of the call are the next call's warm-up
                    const long hj = gi - (a.n_in - a.W);
                    if (!warm && hj >= 0 && a.W > 0)
                        *reinterpret_cast<v4f *>(&hist_next[hj]) = v4f{x0.x, x0.y, x1.x, x1.y};
                }
                p0 = cmul(p0, rowstep);
                p1 = cmul(p1, rowstep);
            }
        }
        DC_TICK(0);
        lds_barrier();
        DC_TICK(1);
        fetch(pos + n);                                       // next tile's input, in flight during the cascade
        // ---------------- the cascade, LDS -> LDS ---------------------------------------------
        int len = n;
        if constexpr (FIX) {
            static_for<0, (FIX ? P::NS : 0)>([&](auto S) {
                constexpr int s = S.value, L = P::KIND[s];
                constexpr bool last = s + 1 == P::NS;
                const v2f *E = lds + LY.roff[s], *O = E + LY.ooff[s];
                v2f *yE = lds + LY.roff[s + 1] + (last ? 0 : dc_hist_of(P::KIND[s + 1]) / 2), *yO = yE + LY.ooff[s + 1];
                if (n == DC_TILE) dc_stage_full<L, ((DC_TILE / 2) >> s)>(E, O, yE, yO, last, t);
                else dc_stage<L>(E, O, yE, yO, last ? yE : nullptr, len >> 1, ptab[s], t);
                lds_barrier();
                len >>= 1;
            });
        } else
        for (int s = 0; s < ns; s++) {
            const int d0 = __builtin_amdgcn_readlane(pv0, s), dn = __builtin_amdgcn_readlane(pv0, s + 1);
            const v2f *E = lds + __builtin_amdgcn_readlane(pv1, s), *O = E + (d0 >> 16);
            const bool last = s + 1 == ns;
            const int hn2 = (dn >> 8) & 0xff;                   // 0 behind the last stage
            v2f *yE = lds + __builtin_amdgcn_readlane(pv1, s + 1) + hn2;
            v2f *yO = yE + (dn >> 16);
            v2f *ylin = last ? yE : nullptr;
            const int nout = len >> 1;
#define DC_CASE(LL) case LL: dc_stage<LL>(E, O, yE, yO, ylin, nout, ptab[s], t); break;
            switch (d0 & 0xff) {
            DC_CASE(3) DC_CASE(11) DC_CASE(15) DC_CASE(19) DC_CASE(23) DC_CASE(27) DC_CASE(31)
            DC_CASE(35) DC_CASE(39) DC_CASE(43) DC_CASE(47)
            default: dc_stage<51>(E, O, yE, yO, ylin, nout, ptab[s], t); break;
            }
#undef DC_CASE
            DC_TICK(8 + s);
            lds_barrier();
            DC_TICK(3);
            len = nout;
        }
        // slide every stage's history: the last hist_s inputs of stage s (hist_s/2 per parity half) move to the
        // front of its region (lanes 0-31 the even half, 32-63 the odd one); all reads, a barrier, then the
        // writes, because a short tile overlaps source and target
        if constexpr (FIX) {
            constexpr int NSF = FIX ? P::NS : 0;
            v2f keep[NSF ? NSF : 1];
            const int i = t & 31, odd = t >> 5;
            static_for<0, NSF>([&](auto S) {
                constexpr int sg = S.value;
                if (i < dc_hist_of(P::KIND[sg]) / 2) keep[sg] = lds[LY.roff[sg] + (odd ? LY.ooff[sg] : 0) + i + (n >> (sg + 1))];
            });
            lds_barrier();
            static_for<0, NSF>([&](auto S) {
                constexpr int sg = S.value;
                if (i < dc_hist_of(P::KIND[sg]) / 2) lds[LY.roff[sg] + (odd ? LY.ooff[sg] : 0) + i] = keep[sg];
            });
        } else {
            constexpr int R = (DC_MAX_STAGES * 64 + DC_T - 1) / DC_T;
            v2f keep[R];
            int dst[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int u = t + r * DC_T, odd = (u >> 5) & 1, i = u & 31;
                const int sg = __builtin_amdgcn_readfirstlane(u >> 6);
                dst[r] = -1;
                if (sg < ns) {
                    const int d = __builtin_amdgcn_readlane(pv0, sg);
                    if (i < ((d >> 8) & 0xff)) {
                        dst[r] = __builtin_amdgcn_readlane(pv1, sg) + (odd ? (d >> 16) : 0) + i;
                        keep[r] = lds[dst[r] + (n >> (sg + 1))];
                    }
                }
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < R; r++) if (dst[r] >= 0) lds[dst[r]] = keep[r];
        }
        DC_TICK(4);
        // ---------------- tile outputs -> HBM ------------------------------------------------------
        if (!warm) {
            const v2f *y = lds + roff[ns];
            const long obase = pos >> ns;
            for (int j = t; j < len; j += DC_T) out[obase + j] = y[j];
        }
        DC_TICK(5);
        lds_barrier();
        DC_TICK(6);
        pos += n;
    }

#ifdef DC_PROFILE
    if (wg == 0 && (t == 0 || t == 256))
        printf("dcprof t%d: mix %llu bar %llu stages %llu bar %llu hist %llu out %llu bar %llu\n", t, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6]);
    if (wg == 0 && (t == 0 || t == 256)) printf("dcstages t%d: %llu %llu %llu %llu %llu %llu\n", t, tk[8], tk[9], tk[10], tk[11], tk[12], tk[13]);
#endif
    // the NCO runs on: phase and age after this call's n_in samples (the host keeps the same arithmetic in its
    // mirror, so a retune uploads a consistent state)
    if (seg == 0 && t == 0) {
        DcChan nx;
        nx.phase = cs.phase + cs.inc * (unsigned long long)a.n_in;
        nx.inc = cs.inc;
        nx.age = cs.age + (unsigned long long)a.n_in;
        a.chan_next[ch] = nx;
    }
    // calls shorter than the warm-up length keep the tail of the old history in front
    if (seg == a.nseg - 1 && a.n_in < a.W)
        for (int j = t; j < a.W - a.n_in; j += DC_T) hist_next[j] = hist[j + a.n_in];
}


}  // namespace csdr
