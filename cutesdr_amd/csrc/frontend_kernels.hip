// frontend_kernels.hip -- the input-rate stages in front of the down-converter.
//
// noiseblank_kernel replaces CNoiseProc::ProcessBlanker (dsp/noiseproc.cpp:121-176).  The reference
// loop looks sequential (running sum, blank counter) but carries no true recurrence:
//   mag_i   = max(|re_i|, |im_i|)
//   S_i     = sum of the last mag_n+1 magnitudes           (moving sum, :143-147)
//   trig_i  = mag_i * ratio > S_i                          (:155-158)
//   blank_i = a trigger among the last width_n samples     (the counter reloaded by each trigger, :160-166)
//   out_i   = blank_i ? 0 : x_{i-delay_n-1}                (delay ring of delay_n+1 entries, :150-153)
// so S is S_start + prefix-sum(mag_i - mag_{i-mag_n-1}) in fp64 and the blank window is a prefix-max
// of trigger positions: one workgroup per channel walks the call in 1024-sample tiles with two block
// scans per tile.  Samples older than the call come from a per-channel history of the last NB_HIST
// raw inputs.
//
// unpack_kernel replaces the datagram conversion loops of CUdpThread::OnreadyRead
// (interface/netiobase.cpp:497-503, 521-526); spurcal_kernel the running I/Q means of
// CSdrInterface::NcoSpurCalibrate (interface/sdrinterface.cpp:829-848).
#include "launch_once.hpp"
#include <cstdlib>
#include <type_traits>
#include "frontend_kernels.h"

namespace csdr {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
// 512 threads and two rounds of workgroups: 2.60 ms against 2.79 with 256 threads and one round (256 receivers x 2^21;
// 1024 threads 3.2 ms) -- fewer streams in flight per L2 at a time, for the sample leaving the moving-sum window
#ifndef NB_THREADS
#define NB_THREADS 512
#endif
// Samples per thread and tile: four, and EIGHT in the mask form with the ring (below).  Round 4, C4 share from 24-bit
// datagrams, per launch: two streams x 4 per thread 1.40 ms (74 registers, three workgroups per CU; bound by the
// fabric: 6.6 GB at 4.8 TB/s); two streams x 8: 18 % fewer vector instructions and the same 1.40 ms (123 registers, two
// workgroups per CU); ring x 4: 1.27 ms, now bound by the vector unit (84 % busy, 305 instructions per wave and
// 256 samples -- most of them per tile and thread: datagram addresses, the two scans, the workgroup sums, the mask
// word); ring x 8: 1.07 ms (92 registers, 64 KB of ring, two workgroups per CU, one round of 512 workgroups).
#ifndef NB_PER_THREAD
#define NB_PER_THREAD 4
#endif
#ifndef NB_MASK_PREFETCH
#define NB_MASK_PREFETCH 2       // tiles the integer mask kernel fetches ahead (1: 760 us, 2 / 3 / 4: 722 / 723 / 724 us for 256 x 2^21 24-bit samples)
#endif
#ifndef NB_PER_THREAD_RING
#define NB_PER_THREAD_RING 8
#endif
constexpr int NB_T = NB_THREADS;
template <bool MASK, bool RING> struct NbTile { static constexpr int PER = MASK && RING ? NB_PER_THREAD_RING : NB_PER_THREAD, TILE = NB_T * PER; };
// (the host cuts segments before it knows which mask form runs: a multiple of both tiles)
int noiseblank_tile(bool mask) { return mask && NbTile<true, true>::TILE > NbTile<true, false>::TILE ? NbTile<true, true>::TILE : NbTile<false, false>::TILE; }
static_assert(NbTile<true, true>::TILE % NbTile<true, false>::TILE == 0 && NbTile<false, true>::TILE == NbTile<false, false>::TILE,
              "segments are whole tiles of either form");
// RING: the magnitudes of a workgroup's last NB_RING samples stay in LDS, so that the sample LEAVING the
// moving-sum window -- mag_n + 1 = 10 001 samples, 60 KB of datagrams, behind the new one -- is not fetched and decoded
// a second time.  That second stream was half of the mask form's traffic (counters: 2 x 3.23 GB at the fabric per
// launch of the C4 share, 4.8 TB/s -- the 96 windows of an XCD's resident workgroups are 5.8 MB against 4 MB of L2,
// so the Infinity Cache served it, not L2).  An instantiation of its own (RING), so that neither form carries the
// other's registers.  Taken by the host when every channel with the blanker on has one tile <= mag_n + 1 <= NB_RING - one tile
// (no barrier between a tile's ring writes and its own ring reads is needed then); else the two-stream form.
// The sample form (three streams: new, leaving, delayed; fp32 rows out) takes the ring too, four samples per thread: its
// leaving stream was a third of 24 B per sample at the fabric.  (Round 3 tried a ring there -- 44 KB per 256-thread
// workgroup, every channel -- and dropped it: 3.36 against 2.86 ms.)
constexpr int NB_RING = 4 * NbTile<true, true>::TILE;       // 16384 magnitudes: a whole number of tiles of either form
static_assert(NB_RING % NbTile<false, true>::TILE == 0, "tiles divide the ring");
int noiseblank_ring_min(bool mask) { return mask ? NbTile<true, true>::TILE : NbTile<false, true>::TILE; }
int noiseblank_ring_max(bool mask) { return NB_RING - noiseblank_ring_min(mask); }

// Wave scans on the DPP network (row_shr 1, 2, 4, 8, then row_bcast 15 into rows 1, 3 and row_bcast 31 into rows
// 2, 3; a step without a source lane reads the identity) instead of __shfl_up: a 64-bit shuffle is two
// ds_bpermute -- an LDS-pipe round trip per step, twelve steps per tile on every wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long nb_dpp64(unsigned long long v, unsigned long long ident)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)ident, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(ident >> 32), (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
#define NB_SCAN_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
__device__ __forceinline__ double wave_incl_scan_add(double v, int)
{
#define NB_STEP(C_, R_) v += __longlong_as_double((long long)nb_dpp64<C_, R_>((unsigned long long)__double_as_longlong(v), 0ull));
    NB_SCAN_STEPS(NB_STEP)
#undef NB_STEP
    return v;
}
// trigger positions inside a tile are 32-bit offsets from the tile's first sample (NB_NEVER: none; an older trigger
// is clamped to it -- the longest blank width is a few tiles)
constexpr int NB_NEVER = -(1 << 30);
__device__ __forceinline__ int wave_incl_scan_max(int v, int)
{
#define NB_STEP(C_, R_) { const int o = __builtin_amdgcn_update_dpp(NB_NEVER, v, C_, R_, 0xf, false); v = o > v ? o : v; }
    NB_SCAN_STEPS(NB_STEP)
#undef NB_STEP
    return v;
}
// the value of the lane in front (lane 0: NB_NEVER)
__device__ __forceinline__ int wave_prev_lane(int v) { return __builtin_amdgcn_update_dpp(NB_NEVER, v, 0x138, 0xf, 0xf, false); }

// MASK: the kernel's mask mode as an instantiation of its own (no third input stream, no sample output: fewer
// registers, more waves)
template <bool MASK, bool RING = false>
__global__ __launch_bounds__(NB_T)
void noiseblank_kernel(NbArgs a)
{
    __shared__ double wsum[NB_T / 64];
    __shared__ int wmax[NB_T / 64];
    extern __shared__ __attribute__((aligned(16))) float nb_ring[];    // [NB_RING] (RING)
    const int ch = blockIdx.x / a.nseg, seg = blockIdx.x % a.nseg, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const NbChan C = a.chan[ch];                        // state at the start of the call (the last segment writes chan_next)
    const f2 *in = reinterpret_cast<const f2 *>(a.in) + (long)ch * a.in_stride;
    f2 *out = reinterpret_cast<f2 *>(a.out) + (long)ch * a.out_stride;
    constexpr bool mask_mode = MASK;
    constexpr int NB_PER = NbTile<MASK, RING>::PER, NB_TILE = NbTile<MASK, RING>::TILE, NB_NP = NB_PER / 2;
    static_assert(NB_PER % 2 == 0 && 240 % NB_PER == 0 && 32 % NB_PER == 0,
                  "a thread's samples are whole sample pairs of one 24-bit datagram and a whole fraction of a mask word");
    unsigned *mrow = mask_mode ? a.mask + (long)ch * a.mask_stride : nullptr;
    const f2 *hist = reinterpret_cast<const f2 *>(a.hist) + (long)ch * NB_HIST;
    f2 *hist_next = reinterpret_cast<f2 *>(a.hist_next) + (long)ch * NB_HIST;
    const int n = a.n;
    // sample i of the call: from the float rows or decoded from the datagrams as they arrived
    const unsigned char *pk = a.wire.pk ? a.wire.pk + (long)ch * a.wire.chan_stride : nullptr;
    const int pkt_len = a.wire.pkt_len;
    auto IN = [&](long i) -> f2 { return pk ? wire_sample(pk, pkt_len, i) : in[i]; };
    auto X = [&](long i) -> f2 { return i >= 0 ? IN(i) : hist[NB_HIST + i]; };   // i >= -NB_HIST
    // this workgroup's segment [seg_a, seg_b) of the call
    const long seg_a = (long)seg * a.seg_len;
    const long seg_b = seg_a + a.seg_len < n ? seg_a + a.seg_len : n;
    const bool last_seg = seg == a.nseg - 1;

    if (C.on) {
        const int M1 = C.mag_n + 1, D1 = C.delay_n + 1, W = C.width_n;
        const double ratio = C.ratio;
        double S0 = C.sum;
        long long last = -C.since_trig;                    // index of the last trigger, relative to this call
        long first = 0;
        constexpr bool ring = RING;                         // (the host has checked NB_TILE <= M1 <= NB_RING - NB_TILE)
        auto rslot = [](long i) -> int { const int r = (int)(i % NB_RING); return r < 0 ? r + NB_RING : r; };   // (start-up paths only)
        if (ring && seg == 0) {
            // the window in front of the call: from the history, once per channel and call
            for (long k = -(long)M1 + t; k < 0; k += NB_T) { const f2 v = X(k); nb_ring[rslot(k)] = fmaxf(fabsf(v.x), fabsf(v.y)); }
            __syncthreads();
        }
        if (seg > 0) {
            // a later segment rebuilds its start state: the moving sum at its warm-up origin is a plain
            // reduction over the mag_n+1 samples before it, and width_n samples of warm-up (outputs
            // dropped) recover the blank window that may reach into the segment
            first = seg_a - (long)((W + NB_TILE - 1) / NB_TILE) * NB_TILE;         // >= 0: seg_len >= 4 warm-ups
            double part = 0.0;
            for (long k = first - M1 + t; k < first; k += NB_T) {
                const f2 v = X(k);
                const float m = fmaxf(fabsf(v.x), fabsf(v.y));
                part += (double)m;
                if (ring) nb_ring[rslot(k)] = m;            // ... which is the window the first tile needs in the ring
            }
            part = wave_incl_scan_add(part, lane);
            if (lane == 63) wsum[w] = part;
            __syncthreads();
            S0 = 0.0;
            for (int q = 0; q < NB_T / 64; q++) S0 += wsum[q];
            last = -(1LL << 40);
            __syncthreads();
        }
        // the three loads of a tile (new sample, the one leaving the window, the delayed one) are issued
        // one tile ahead, so that they are in flight while the current tile goes through its scans
        // pw: the prefetched tile -- three streams x NB_PER samples as fp32 pairs (slot s at 2*NB_PER*s), or, for a tile
        // inside 24-bit datagrams, the raw words (new stream: 2 sample pairs = 6 words at 0; leaving and delayed stream:
        // 3 pairs = 9 words at 6 and 15 -- with four samples per thread; NB_NP pairs, then twice NB_NP + 1 in general),
        // decoded where they are consumed so that the fetch does not wait for itself
        unsigned pw[6 * NB_PER];
        bool praw = false;                                  // uniform: what the last fetch left in pw
        bool praw16 = false;                                // ... raw 16-bit datagram words (ring form)
        auto put = [&](int slot, int k, f2 v) { pw[2 * NB_PER * slot + 2 * k] = __float_as_uint(v.x); pw[2 * NB_PER * slot + 2 * k + 1] = __float_as_uint(v.y); };
        auto fetch = [&](long b0) {
            const bool inside = (ring || b0 - M1 >= 0) && (mask_mode || b0 - D1 >= 0) && b0 + NB_TILE <= seg_b;
            praw = false; praw16 = false;
            if (pk && pkt_len == 1444 && inside) {
                // 24-bit datagrams: a sample pair (even index) is 12 bytes at a 4-byte aligned offset and never straddles a
                // datagram; a thread's four samples of a stream are two pairs (even start) or parts of three (odd start)
                const unsigned i0 = (unsigned)(b0 + (long)t * NB_PER);
                auto pair_words = [&](unsigned e, unsigned *dst) {   // e even
                    const unsigned q = e / 240u, j = e - q * 240u;
                    const unsigned *wp = reinterpret_cast<const unsigned *>(pk + (q * 1444u + 4u + 6u * j));
                    dst[0] = wp[0]; dst[1] = wp[1]; dst[2] = wp[2];
                };
                const unsigned eo = (i0 - (unsigned)M1) & ~1u, ed = (i0 - (unsigned)D1) & ~1u;
                if constexpr (RING && MASK) {
                    // (a thread's samples start at a multiple of NB_PER, which divides 240: they lie in ONE datagram, 6 bytes
                    // apart -- one division instead of one per pair)
                    const unsigned q = i0 / 240u, j = i0 - q * 240u;
                    const unsigned *wp = reinterpret_cast<const unsigned *>(pk + (q * 1444u + 4u + 6u * j));
#pragma unroll
                    for (int k = 0; k < 3 * NB_NP; k++) pw[k] = wp[k];
                } else {
#pragma unroll
                    for (int p = 0; p < NB_NP; p++) pair_words(i0 + 2u * p, pw + 3 * p);
                }
                if (!ring) {
#pragma unroll
                    for (int p = 0; p < NB_NP; p++) pair_words(eo + 2u * p, pw + 3 * NB_NP + 3 * p);
                    if (M1 & 1) pair_words(eo + 2u * NB_NP, pw + 6 * NB_NP);    // (uniform) an odd start touches one pair more
                }
                if (!mask_mode) {
#pragma unroll
                    for (int p = 0; p <= NB_NP; p++) pair_words(ed + 2u * p, pw + 6 * NB_NP + 3 + 3 * p);
                }
                praw = true;
                return;
            }
            // 16-bit datagrams, ring form (no second or third stream to fetch): a sample is one word, 256 to a datagram, a
            // thread's samples contiguous inside one -- plain word loads, decoded where they are consumed (the per-sample
            // path below cost the 16-bit chain a third more in this kernel: 1.43 against 1.07 ms)
            if (RING && mask_mode && pk && pkt_len == 1028 && inside) {
                static_assert(256 % NbTile<true, true>::PER == 0, "a thread's samples lie inside one 16-bit datagram");
                const unsigned i0 = (unsigned)(b0 + (long)t * NB_PER);
                const unsigned *wp = reinterpret_cast<const unsigned *>(pk + ((i0 >> 8) * 1028u + 4u + 4u * (i0 & 255u)));
#pragma unroll
                for (int k = 0; k < NB_PER; k++) pw[k] = wp[k];
                praw16 = true;
                return;
            }
            // a tile whose three streams lie inside this call's float rows (all but the first tiles of a call and a
            // segment's last one): plain loads, no per-sample source selection or bounds
            if (!pk && inside) {
                // a thread's four samples of each stream are 32 contiguous bytes: two 16-byte loads (the leaving and the
                // delayed stream are only 8-byte aligned, which a global dwordx4 load accepts)
                const f2 *p = in + b0 + (long)t * NB_PER;
                static_assert(NB_PER % 2 == 0, "wide loads take sample pairs");
                auto ld = [&](const f2 *q, int slot) {
#pragma unroll
                    for (int k = 0; k < NB_PER; k += 2) {
                        const f4u v = *reinterpret_cast<const f4u *>(q + k);
                        put(slot, k, f2{v.x, v.y}); put(slot, k + 1, f2{v.z, v.w});
                    }
                };
                ld(p, 0);
                if (!ring) ld(p - M1, 1);
                if (!mask_mode) ld(p - D1, 2);
                return;
            }
#pragma unroll
            for (int k = 0; k < NB_PER; k++) {
                const long i = b0 + (long)t * NB_PER + k;
                if (i < seg_b) { put(0, k, IN(i)); if (!ring) put(1, k, X(i - M1)); if (!mask_mode) put(2, k, X(i - D1)); }
            }
        };
        // the prefetched tile as samples: new, leaving, delayed
        auto take = [&](f2 *x, f2 *xo, f2 *xdl) {
            if (praw16) {
#pragma unroll
                for (int k = 0; k < NB_PER; k++)
                    x[k] = f2{(float)(short)(pw[k] & 0xffffu), (float)(short)(pw[k] >> 16)};      // (I low half, Q high half: wire_format.hpp)
            } else if (praw) {
                auto pair = [&](const unsigned *w, f2 &a, f2 &b) {
                    const wf4 v = wire_pair_decode(wf4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), 0.f}, 1444);
                    a = f2{v.x, v.y}; b = f2{v.z, v.w};
                };
#pragma unroll
                for (int p = 0; p < NB_NP; p++) pair(pw + 3 * p, x[2 * p], x[2 * p + 1]);
                auto many = [&](const unsigned *w, bool odd, f2 *dst) {   // odd is uniform: each side decodes what it needs
                    f2 skip;
                    if (odd) {
                        pair(w, skip, dst[0]);
#pragma unroll
                        for (int p = 1; p < NB_NP; p++) pair(w + 3 * p, dst[2 * p - 1], dst[2 * p]);
                        pair(w + 3 * NB_NP, dst[NB_PER - 1], skip);
                    } else {
#pragma unroll
                        for (int p = 0; p < NB_NP; p++) pair(w + 3 * p, dst[2 * p], dst[2 * p + 1]);
                    }
                };
                if (!ring) many(pw + 3 * NB_NP, M1 & 1, xo);
                if (!mask_mode) many(pw + 6 * NB_NP + 3, D1 & 1, xdl);
            } else {
#pragma unroll
                for (int k = 0; k < NB_PER; k++) {
                    x[k] = f2{__uint_as_float(pw[2 * k]), __uint_as_float(pw[2 * k + 1])};
                    if (!ring) xo[k] = f2{__uint_as_float(pw[2 * NB_PER + 2 * k]), __uint_as_float(pw[2 * NB_PER + 2 * k + 1])};
                    if (!mask_mode) xdl[k] = f2{__uint_as_float(pw[4 * NB_PER + 2 * k]), __uint_as_float(pw[4 * NB_PER + 2 * k + 1])};
                }
            }
        };
        fetch(first);
        int rb = ring ? rslot(first) : 0;                   // ring slot of the tile's first sample (tiles divide the ring)
        for (long base = first; base < seg_b; base += NB_TILE) {
            f2 xd[NB_PER], xn[NB_PER], xl[NB_PER], xt[NB_PER];
            float mag[NB_PER];
            double d[NB_PER], run = 0.0;
            take(xn, xl, xt);
            float far[NB_PER];                              // ring form: the magnitudes leaving the window
            if constexpr (RING) {
                {
                    // a thread's NB_PER window-leaving magnitudes start at an arbitrary (uniform) offset from a 16-byte
                    // boundary of the ring: NB_PER / 4 + 1 aligned reads, picked apart by that offset
                    int f0 = rb - M1 + t * NB_PER;          // > -NB_RING
                    f0 = f0 < 0 ? f0 + NB_RING : f0;
                    const float4 *src = reinterpret_cast<const float4 *>(nb_ring);
                    float q[NB_PER + 4];
#pragma unroll
                    for (int v = 0; v <= NB_PER / 4; v++) {
                        int fv = (f0 >> 2) + v;             // (NB_RING is a multiple of four: no 16-byte read straddles the wrap)
                        fv = fv >= NB_RING / 4 ? fv - NB_RING / 4 : fv;
                        const float4 w4 = src[fv];
                        q[4 * v] = w4.x; q[4 * v + 1] = w4.y; q[4 * v + 2] = w4.z; q[4 * v + 3] = w4.w;
                    }
                    switch ((unsigned)(-M1) & 3u) {         // = f0 & 3: base and t * NB_PER are multiples of four
#define NB_PICK(O_) case O_: _Pragma("unroll") for (int k = 0; k < NB_PER; k++) far[k] = q[O_ + k]; break;
                        NB_PICK(0) NB_PICK(1) NB_PICK(2) default: NB_PICK(3)
#undef NB_PICK
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NB_PER; k++) {
                mag[k] = 0.f; d[k] = 0.0; xd[k] = f2{0.f, 0.f};
                if (base + (long)t * NB_PER + k < seg_b) {
                    const f2 x = xn[k];
                    xd[k] = xt[k];
                    mag[k] = fmaxf(fabsf(x.x), fabsf(x.y));
                    float mo;
                    if (ring) mo = far[k];
                    else { const f2 xo = xl[k]; mo = fmaxf(fabsf(xo.x), fabsf(xo.y)); }
                    d[k] = (double)mag[k] - (double)mo;
                }
                run += d[k];
                d[k] = run;                                // thread-local inclusive prefix
            }
            if constexpr (RING) {
                {                                           // this tile's magnitudes: read again mag_n + 1 samples from now
                    float4 *dst = reinterpret_cast<float4 *>(nb_ring) + ((rb + t * NB_PER) >> 2);
#pragma unroll
                    for (int v = 0; v < NB_PER / 4; v++) dst[v] = make_float4(mag[4 * v], mag[4 * v + 1], mag[4 * v + 2], mag[4 * v + 3]);
                }
            }
            if (base + NB_TILE < seg_b) fetch(base + NB_TILE);
            const double incl = wave_incl_scan_add(run, lane);
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            double off = S0 + (incl - run);
            for (int q = 0; q < w; q++) off += wsum[q];
            double total = 0.0;
            for (int q = 0; q < NB_T / 64; q++) total += wsum[q];
            // triggers and the position of the latest one at or before each sample, as offsets from `base`
            const long left = seg_b - base, lead = seg_a - base;
            const int nvalid = left < NB_TILE ? (int)left : NB_TILE, nskip = lead > 0 ? (int)lead : 0;
            const long long ago = last - (long long)base;   // <= -1
            const int last_rel = ago < (long long)NB_NEVER ? NB_NEVER : (int)ago;
            int lt[NB_PER], runmax = NB_NEVER;
#pragma unroll
            for (int k = 0; k < NB_PER; k++) {
                const int r = t * NB_PER + k;
                const bool trig = r < nvalid && (double)mag[k] * ratio > off + d[k];
                if (trig) runmax = r;
                lt[k] = runmax;
            }
            const int inclm = wave_incl_scan_max(runmax, lane);
            if (lane == 63) wmax[w] = inclm;
            __syncthreads();
            int before = last_rel;                          // latest trigger before this thread's samples
            for (int q = 0; q < w; q++) before = wmax[q] > before ? wmax[q] : before;
            const int upto = wave_prev_lane(inclm);
            if (upto > before) before = upto;
            int tile_last = NB_NEVER;
            for (int q = 0; q < NB_T / 64; q++) tile_last = wmax[q] > tile_last ? wmax[q] : tile_last;
            if (mask_mode) {
                // a thread's blank flags are a nibble (a byte with eight samples per thread); eight (four) neighbouring
                // lanes make a word of the mask row (tiles start at multiples of the tile length, so words never
                // straddle tiles; a warm-up tile is skipped whole)
                constexpr int LPW = 32 / NB_PER;            // lanes per mask word
                unsigned nib = 0;
#pragma unroll
                for (int k = 0; k < NB_PER; k++) {
                    const int r = t * NB_PER + k;
                    const int l = lt[k] > before ? lt[k] : before;
                    if (r < nvalid && r - l < W) nib |= 1u << k;
                }
                unsigned wv = nib << (NB_PER * (lane & (LPW - 1)));
#pragma unroll
                for (int sh = 1; sh < LPW; sh <<= 1) wv |= (unsigned)__shfl_xor((int)wv, sh);
                if ((lane & (LPW - 1)) == 0 && nskip == 0 && t * NB_PER < nvalid) mrow[(base + (long)t * NB_PER) >> 5] = wv;
            } else {
#pragma unroll
            for (int k = 0; k < NB_PER; k++) {
                const int r = t * NB_PER + k;
                if (r < nvalid && r >= nskip) {
                    const int l = lt[k] > before ? lt[k] : before;
                    out[base + r] = (r - l < W) ? f2{0.f, 0.f} : xd[k];
                }
            }
            }
            S0 += total;
            if (ring) { rb += NB_TILE; rb = rb >= NB_RING ? rb - NB_RING : rb; }
            if (tile_last > NB_NEVER) last = (long long)base + tile_last;
            __syncthreads();                               // wsum / wmax reused by the next tile
        }
        if (t == 0 && last_seg) {
            NbChan N = C;
            N.sum = S0;
            long long age = (long long)n - last;
            if (age > (1LL << 40)) age = 1LL << 40;
            N.since_trig = age;
            a.chan_next[ch] = N;
        }
    } else {
        if (t == 0 && last_seg) a.chan_next[ch] = C;
    }
    if (!C.on && mask_mode)                                 // off: nothing is blanked (and the consumer applies no delay)
        for (long wv = (seg_a >> 5) + t; wv < ((seg_b + 31) >> 5); wv += NB_T) mrow[wv] = 0u;
    if (!C.on && !mask_mode && (pk || a.out != a.in)) {
        for (long i = seg_a + t; i < seg_b; i += NB_T) out[i] = IN(i);  // off: the data passes through (:125-129)
    }
    // the last NB_HIST inputs of [history | this call] are the next call's history
    if (last_seg)
        for (long j = t; j < NB_HIST; j += NB_T) hist_next[j] = X((long)n - NB_HIST + j);
}


// =====================================================================================================================
// The mask form on DATAGRAM input in integers (round 6).  A 16- or 24-bit datagram sample is an integer multiple of 2^-8
// of the float the general kernel decodes (wire_format.hpp), so the magnitudes max(|I|, |Q|) are integers below 2^23 in
// that unit and the moving sum -- at most 32768 of them -- is an integer below 2^38: every fp64 sum of the general
// kernel (and of the reference: m_MagAveSum, noiseproc.cpp:143-147) is EXACT on such input, in any order, and equals the
// 64-bit integer sum here bit for bit.  What the integers buy: no decode to float, no float -> double conversions, a
// 32-bit subtract and add per sample instead of three half-rate fp64 operations, and -- the larger part -- the trigger
// test `mag * ratio > S` (noiseproc.cpp:155-158: one conversion, one fp64 multiply, one add, one compare per sample)
// is first asked once per THREAD in fp32 with a safety margin (largest of its eight magnitudes against the smallest of
// its eight sums): a thread without a candidate -- all but a few per launch -- skips the eight fp64 tests and the per-sample
// trigger positions.  A thread with a candidate runs the general kernel's exact fp64 test on the same numbers scaled by
// 2^8 (a power of two: the products and compares round identically), so the decisions are the general kernel's, sample
// for sample.  The host takes this kernel only while EVERY sample the blanker has seen since its last set-up came from
// datagrams (csdr_noiseproc_batch: datagram_only) -- the state it inherits (sum, history) is then integral in that unit.
// Ring form only (the window's magnitudes in LDS, as integers); eight samples per thread.
// |a - b| of two unsigned words, b wave-uniform: one instruction (the compiler has no builtin for it)
__device__ __forceinline__ unsigned nb_absdiff(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// the same for a value whose sixteen-lane row sums fit 32 bits (|v| < 2^27): the four steps inside a row are one
// v_add_u32 with a DPP operand each, only the two steps across rows carry 64 bits
__device__ __forceinline__ long long wave_incl_scan_add_i27(int v)
{
#define NB_STEP(C_, R_) v += __builtin_amdgcn_update_dpp(0, v, C_, R_, 0xf, false);
    NB_STEP(0x111, 0xf) NB_STEP(0x112, 0xf) NB_STEP(0x114, 0xf) NB_STEP(0x118, 0xf)
#undef NB_STEP
    long long w = (long long)v;
    w += (long long)nb_dpp64<0x142, 0xa>((unsigned long long)w, 0ull);
    w += (long long)nb_dpp64<0x143, 0xc>((unsigned long long)w, 0ull);
    return w;
}
__device__ __forceinline__ long long wave_incl_scan_add_i64(long long v)
{
#define NB_STEP(C_, R_) v += (long long)nb_dpp64<C_, R_>((unsigned long long)v, 0ull);
    NB_SCAN_STEPS(NB_STEP)
#undef NB_STEP
    return v;
}
template <int FMT>                                           // 1444: 24-bit datagrams, 1028: 16-bit
__global__ __launch_bounds__(NB_T)
void noiseblank_mask_int_kernel(NbArgs a)
{
    constexpr int NB_PER = NbTile<true, true>::PER, NB_TILE = NbTile<true, true>::TILE;
    static_assert(NB_PER == 8, "the packing below is written for eight samples per thread");
    static_assert(NB_T == 512, "eight waves: their sums and trigger positions cross in one row of eight lanes (three DPP steps)");
    __shared__ long long wsum[NB_T / 64];
    __shared__ int wmax[NB_T / 64];
    extern __shared__ __attribute__((aligned(16))) float nb_ring[];
    int *ring = reinterpret_cast<int *>(nb_ring);           // [NB_RING] integer magnitudes, units of 2^-8, in eight planes (below)
    const int ch = blockIdx.x / a.nseg, seg = blockIdx.x % a.nseg, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const NbChan C = a.chan[ch];
    unsigned *mrow = a.mask + (long)ch * a.mask_stride;
    const f2 *hist = reinterpret_cast<const f2 *>(a.hist) + (long)ch * NB_HIST;
    f2 *hist_next = reinterpret_cast<f2 *>(a.hist_next) + (long)ch * NB_HIST;
    const int n = a.n;
    const unsigned char *pk = a.wire.pk + (long)ch * a.wire.chan_stride;
    auto X = [&](long i) -> f2 { return i >= 0 ? wire_sample(pk, FMT, i) : hist[NB_HIST + i]; };
    // integer magnitude of sample i of [history | call] (the history holds the floats of earlier datagrams: integral)
    auto imag_at = [&](long i) -> int { const f2 v = X(i); return (int)(fmaxf(fabsf(v.x), fabsf(v.y)) * 256.0f); };
    const long seg_a = (long)seg * a.seg_len;
    const long seg_b = seg_a + a.seg_len < n ? seg_a + a.seg_len : n;
    const bool last_seg = seg == a.nseg - 1;
    if (C.on) {
        const int M1 = C.mag_n + 1, W = C.width_n;
        const double ratio = C.ratio;
        const float ratio_hi = (float)ratio * 1.000004f;    // fp32 screen: errs on the side of calling the exact test
        long long S0 = (long long)llrint(C.sum * 256.0);
        long long last = -C.since_trig;
        long first = 0;
        // The ring in EIGHT PLANES: sample s of the stream (counted so that tiles start at multiples of the tile) sits in
        // plane s mod 8 at word (s div 8) mod NB_RING / 8.  A thread's eight new magnitudes then go to the eight planes at
        // ONE word index, consecutive lanes to consecutive words (eight conflict-free 4-byte stores at constant offsets);
        // and the eight leaving the window, M1 samples back, come from the planes (k - M1) mod 8 at one of two word indices,
        // again lane-consecutive -- where eight consecutive words per thread (32 bytes between lanes) kept the LDS unit busy
        // 41 % of the kernel with three quarters of that in bank conflicts (tools/experiments/pmc_mask_int.sh)
        constexpr int PLANE = NB_RING / 8;
        static_assert((PLANE & (PLANE - 1)) == 0 && NB_TILE / 8 <= PLANE, "plane words are a power of two");
        auto rslot = [](long i) -> int { return (int)(i & 7) * PLANE + (int)((i >> 3) & (PLANE - 1)); };
        auto rbase = [](long i) -> int { const int r = (int)(i % NB_RING); return r < 0 ? r + NB_RING : r; };
        // plane and word step of the k-th sample leaving the window (wave-uniform, fixed for the launch)
        int far_plane[NB_PER], far_hi[NB_PER];
        const int far_e0 = (int)((-(long)M1) >> 3);
#pragma unroll
        for (int k = 0; k < NB_PER; k++) { far_plane[k] = (int)((k - (long)M1) & 7) * PLANE; far_hi[k] = (int)((k - (long)M1) >> 3) - far_e0; }
        if (seg == 0) {
            for (long k = -(long)M1 + t; k < 0; k += NB_T) ring[rslot(k)] = imag_at(k);
            __syncthreads();
        } else {
            first = seg_a - (long)((W + NB_TILE - 1) / NB_TILE) * NB_TILE;
            long long part = 0;
            for (long k = first - M1 + t; k < first; k += NB_T) { const int m = imag_at(k); part += m; ring[rslot(k)] = m; }
            part = wave_incl_scan_add_i64(part);
            if (lane == 63) wsum[w] = part;
            __syncthreads();
            S0 = 0;
            for (int q = 0; q < NB_T / 64; q++) S0 += wsum[q];
            last = -(1LL << 40);
            __syncthreads();
        }
        // a thread's eight samples of a tile as raw words, fetched one tile ahead: 12 words (24 bit) or 8 (16 bit); a
        // thread's samples start at a multiple of eight, which divides 240 and 256: they lie in ONE datagram
        constexpr int NW = FMT == 1444 ? 12 : 8;
        // (TWO tiles ahead: the occupancy is LDS's -- two workgroups of 64 KB, four waves per SIMD -- so the registers of a
        // second buffer are free, and one tile ahead left 48 bytes per lane in flight against an iteration of ~3 us)
        constexpr int PF = NB_MASK_PREFETCH;
        unsigned pwq[PF][NW];
        auto fetch = [&](long b0, unsigned (&pw)[NW]) {
            const unsigned i0 = (unsigned)(b0 + (long)t * NB_PER);
            if ((long)i0 + NB_PER <= seg_b) {
                const unsigned *wp;
                if constexpr (FMT == 1444) { const unsigned q = i0 / 240u, j = i0 - q * 240u; wp = reinterpret_cast<const unsigned *>(pk + (q * 1444u + 4u + 6u * j)); }
                else wp = reinterpret_cast<const unsigned *>(pk + ((i0 >> 8) * 1028u + 4u + 4u * (i0 & 255u)));
#pragma unroll
                for (int k = 0; k < NW; k++) pw[k] = wp[k];
            } else {
#pragma unroll
                for (int k = 0; k < NW; k++) pw[k] = 0u;                       // past the segment: zeros (never counted: r >= nvalid)
                // (a thread that straddles the end cannot exist: seg_b is a multiple of eight -- whole datagrams)
            }
        };
        auto iabs = [](int v) -> int { return v < 0 ? -v : v; };
        fetch(first, pwq[0]);
#pragma unroll
        for (int q = 1; q < PF; q++)
            if (first + q * NB_TILE < seg_b) fetch(first + q * NB_TILE, pwq[q]);
        int rb = rbase(first);                              // the tile's first sample, counted around the ring
        auto tile = [&](const long base, unsigned (&pw)[NW]) {
#ifdef NB_MASK_LOADONLY                                      // experiment: the kernel's loads alone (the floor of its access pattern)
            {
                unsigned x = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) x ^= pw[k];
                if (base + PF * NB_TILE < seg_b) fetch(base + PF * NB_TILE, pw);
                if (x == 0x12345679u) mrow[(base + (long)t * NB_PER) >> 5] = x;
                return;
            }
#endif
            int mag[NB_PER];
            if constexpr (FMT == 1444) {
#pragma unroll
                for (int p = 0; p < NB_PER / 2; p++) {
                    // two samples = I0 Q0 I1 Q1, three bytes each, in three words.  With the four sign bits flipped (three
                    // XORs on the packed words) a component is offset binary, b = x + 2^23, and |x| = |b - 2^23| is one
                    // v_sad_u32; the two components that straddle words are picked out by one v_perm_b32 each
                    const unsigned d0 = pw[3 * p] ^ 0x00800000u, d1 = pw[3 * p + 1] ^ 0x00008000u, d2 = pw[3 * p + 2] ^ 0x80000080u;
                    const unsigned bi0 = d0 & 0x00ffffffu;
                    const unsigned bq0 = __builtin_amdgcn_perm(d1, d0, 0x0c050403u);      // bytes d0[3] d1[0] d1[1] 0
                    const unsigned bi1 = __builtin_amdgcn_perm(d2, d1, 0x0c040302u);      // bytes d1[2] d1[3] d2[0] 0
                    const unsigned bq1 = d2 >> 8;
                    const unsigned ai0 = nb_absdiff(bi0, 0x00800000u), aq0 = nb_absdiff(bq0, 0x00800000u);
                    const unsigned ai1 = nb_absdiff(bi1, 0x00800000u), aq1 = nb_absdiff(bq1, 0x00800000u);
                    mag[2 * p] = (int)(ai0 > aq0 ? ai0 : aq0); mag[2 * p + 1] = (int)(ai1 > aq1 ? ai1 : aq1);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NB_PER; k++) {
                    const int i = (int)(short)(pw[k] & 0xffffu), q = (int)(short)(pw[k] >> 16);
                    const int ai = iabs(i), aq = iabs(q);
                    mag[k] = (ai > aq ? ai : aq) << 8;
                }
            }
            const long left = seg_b - base, lead = seg_a - base;
            const int nvalid = left < NB_TILE ? (int)left : NB_TILE, nskip = lead > 0 ? (int)lead : 0;
            const bool live = t * NB_PER < nvalid;          // (nvalid is a multiple of eight: a thread is in or out whole)
            // the magnitudes leaving the window
            int far[NB_PER];
            {
                const int p0 = ((rb >> 3) + t + far_e0) & (PLANE - 1), p1 = (p0 + 1) & (PLANE - 1);
#pragma unroll
                for (int k = 0; k < NB_PER; k++) far[k] = ring[far_plane[k] + (far_hi[k] ? p1 : p0)];
            }
            int d[NB_PER], run = 0, mmax = 0, dmin = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < NB_PER; k++) {
                run += mag[k] - far[k];                     // (a thread past the end of a call's last tile: unused, see below)
                d[k] = run;                                 // thread-local inclusive prefix (|.| < 2^26)
                mmax = mag[k] > mmax ? mag[k] : mmax;
                dmin = run < dmin ? run : dmin;
            }
            if (!live) run = 0;                            // (its fetch returned zeros: magnitudes 0 into the ring; nothing into the sums)
            {
                int *dst = ring + (rb >> 3) + t;             // (rb / 8 + t < PLANE: tiles divide the ring)
#pragma unroll
                for (int k = 0; k < NB_PER; k++) dst[k * PLANE] = mag[k];
            }
            if (base + PF * NB_TILE < seg_b) fetch(base + PF * NB_TILE, pw);
            const long long incl = wave_incl_scan_add_i27(run);
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            // the waves' sums: lane q < 8 reads wave q's, a three-step prefix inside the first row, two broadcasts (instead
            // of sixteen LDS reads per thread)
            long long ws = lane < NB_T / 64 ? wsum[lane] : 0;
            ws += (long long)nb_dpp64<0x111, 0xf>((unsigned long long)ws, 0ull);
            ws += (long long)nb_dpp64<0x112, 0xf>((unsigned long long)ws, 0ull);
            ws += (long long)nb_dpp64<0x114, 0xf>((unsigned long long)ws, 0ull);
            const int wu = __builtin_amdgcn_readfirstlane(w);
            auto lane_of = [](long long v, int l) -> long long {
                return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), l) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)v, l));
            };
            const long long total = lane_of(ws, NB_T / 64 - 1);
            const long long off = S0 + (incl - run) + (wu > 0 ? lane_of(ws, wu - 1) : 0);
            // triggers: once per thread in fp32 (largest magnitude against the smallest sum, 4e-6 of margin against the 6e-8 of
            // the three roundings); only a thread with a candidate runs the exact tests
            const long long ago = last - (long long)base;
            const int last_rel = ago < (long long)NB_NEVER ? NB_NEVER : (int)ago;
            int lt[NB_PER], runmax = NB_NEVER;
            const bool candidate = live && (float)mmax * ratio_hi >= (float)(off + (long long)dmin);
            if (candidate) {
#pragma unroll
                for (int k = 0; k < NB_PER; k++) {
                    const int r = t * NB_PER + k;
                    const bool trig = (double)mag[k] * ratio > (double)(off + (long long)d[k]);
                    if (trig) runmax = r;
                    lt[k] = runmax;
                }
            }
            // (a wave without a candidate -- nearly every wave of nearly every tile -- has no trigger: no scan)
            const int inclm = __any(candidate) ? wave_incl_scan_max(runmax, lane) : NB_NEVER;
            if (lane == 63) wmax[w] = inclm;
            __syncthreads();
            int wm = lane < NB_T / 64 ? wmax[lane] : NB_NEVER;
            { const int o = __builtin_amdgcn_update_dpp(NB_NEVER, wm, 0x111, 0xf, 0xf, false); wm = o > wm ? o : wm; }
            { const int o = __builtin_amdgcn_update_dpp(NB_NEVER, wm, 0x112, 0xf, 0xf, false); wm = o > wm ? o : wm; }
            { const int o = __builtin_amdgcn_update_dpp(NB_NEVER, wm, 0x114, 0xf, 0xf, false); wm = o > wm ? o : wm; }
            int before = last_rel;
            if (wu > 0) { const int o = __builtin_amdgcn_readlane(wm, wu - 1); before = o > before ? o : before; }
            const int upto = wave_prev_lane(inclm);
            if (upto > before) before = upto;
            const int tile_last = __builtin_amdgcn_readlane(wm, NB_T / 64 - 1);
            {
                constexpr int LPW = 32 / NB_PER;
                unsigned nib = 0;
                if (live) {
                    if (candidate) {
#pragma unroll
                        for (int k = 0; k < NB_PER; k++) {
                            const int r = t * NB_PER + k;
                            const int l = lt[k] > before ? lt[k] : before;
                            if (r - l < W) nib |= 1u << k;
                        }
                    } else {
                        // no trigger among this thread's samples: sample r is blanked while r - before < W
                        const int nb = W + before - t * NB_PER;                 // how many of the eight, from the first
                        nib = nb <= 0 ? 0u : (nb >= NB_PER ? 0xffu : (1u << nb) - 1u);
                    }
                }
                static_assert(LPW == 4, "four lanes make a mask word: one quad");
                unsigned wv = nib << (NB_PER * (lane & (LPW - 1)));
                wv |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)wv, 0xb1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
                wv |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)wv, 0x4e, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
                if ((lane & (LPW - 1)) == 0 && nskip == 0 && live) mrow[(base + (long)t * NB_PER) >> 5] = wv;
            }
            S0 += total;
            rb += NB_TILE; rb = rb >= NB_RING ? rb - NB_RING : rb;
            if (tile_last > NB_NEVER) last = (long long)base + tile_last;
            // (no barrier here: wsum is next written behind this tile's second barrier, which every wave reaches only after
            // its reads of wsum; wmax is next written behind the NEXT tile's first barrier, reached only after the reads of wmax)
        };
        for (long base = first; base < seg_b; base += PF * NB_TILE) {
            tile(base, pwq[0]);
#pragma unroll
            for (int q = 1; q < PF; q++)
                if (base + q * NB_TILE < seg_b) tile(base + q * NB_TILE, pwq[q]);
        }
        if (t == 0 && last_seg) {
            NbChan N = C;
            N.sum = (double)S0 * (1.0 / 256.0);
            long long age = (long long)n - last;
            if (age > (1LL << 40)) age = 1LL << 40;
            N.since_trig = age;
            a.chan_next[ch] = N;
        }
    } else {
        if (t == 0 && last_seg) a.chan_next[ch] = C;
        for (long wv = (seg_a >> 5) + t; wv < ((seg_b + 31) >> 5); wv += NB_T) mrow[wv] = 0u;
    }
    if (last_seg)
        for (long j = t; j < NB_HIST; j += NB_T) hist_next[j] = X((long)n - NB_HIST + j);
}

hipError_t noiseblank_launch(const NbArgs &a, hipStream_t stream)
{
    if (a.out == nullptr && a.ring && a.int_ok && a.wire.pk) {      // datagrams in, mask out, integral state: the integer form
        if (a.wire.pkt_len == 1444) {
            hipError_t e = CSDR_MAX_LDS_ONCE((&noiseblank_mask_int_kernel<1444>), NB_RING * 4);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((noiseblank_mask_int_kernel<1444>), dim3(a.channels * a.nseg), dim3(NB_T), NB_RING * 4, stream, a);
        } else {
            hipError_t e = CSDR_MAX_LDS_ONCE((&noiseblank_mask_int_kernel<1028>), NB_RING * 4);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((noiseblank_mask_int_kernel<1028>), dim3(a.channels * a.nseg), dim3(NB_T), NB_RING * 4, stream, a);
        }
        return hipGetLastError();
    }
    if (a.out == nullptr && a.ring) {
        hipError_t e = CSDR_MAX_LDS_ONCE((&noiseblank_kernel<true, true>), NB_RING * 4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((noiseblank_kernel<true, true>), dim3(a.channels * a.nseg), dim3(NB_T), NB_RING * 4, stream, a);
    }
    else if (a.out == nullptr)
        hipLaunchKernelGGL((noiseblank_kernel<true, false>), dim3(a.channels * a.nseg), dim3(NB_T), 0, stream, a);
    else if (a.ring) {
        // once per device (static + dynamic LDS are above 64 KB)
        hipError_t e = CSDR_MAX_LDS_ONCE((&noiseblank_kernel<false, true>), NB_RING * 4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((noiseblank_kernel<false, true>), dim3(a.channels * a.nseg), dim3(NB_T), NB_RING * 4, stream, a);
    } else hipLaunchKernelGGL((noiseblank_kernel<false, false>), dim3(a.channels * a.nseg), dim3(NB_T), 0, stream, a);
    return hipGetLastError();
}

// ---------------- wire format -> complex fp32 ----------------
// A thread converts two consecutive samples of a datagram with aligned 32-bit loads (datagram
// lengths, the 4-byte header and a sample pair -- 12 or 8 bytes -- are all multiples of 4) and one
// 16-byte store.
__global__ void unpack_kernel(const unsigned char *pk, long chan_stride, int npackets, int pkt_len, int per,
                              float *out, long out_stride, const double *dc)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;        // sample pair within the channel
    const int ch = blockIdx.y;
    const int half = per / 2;
    if (p >= (long)npackets * half) return;
    const long q = p / half;
    const int j = (int)(p - q * half);
    const unsigned *w = reinterpret_cast<const unsigned *>(pk + (long)ch * chan_stride + q * pkt_len + 4);
    float v[4];
    if (pkt_len == 1444) {                                            // 24 bit: value << 8 in an int32, / 65536
        const unsigned d0 = w[3 * j], d1 = w[3 * j + 1], d2 = w[3 * j + 2];
        const int i0 = (int)(d0 << 8);
        const int q0 = (int)(((d0 >> 24) << 8) | (d1 << 16));
        const int i1 = (int)(((d1 >> 16) << 8) | (d2 << 24));
        const int q1 = (int)(d2 & 0xffffff00u);
        v[0] = (float)i0 * (1.0f / 65536.0f); v[1] = (float)q0 * (1.0f / 65536.0f);   // exact: 24 significant bits
        v[2] = (float)i1 * (1.0f / 65536.0f); v[3] = (float)q1 * (1.0f / 65536.0f);
    } else {                                                          // 16 bit
        const unsigned d0 = w[2 * j], d1 = w[2 * j + 1];
        v[0] = (float)(short)(d0 & 0xffffu); v[1] = (float)(short)(d0 >> 16);
        v[2] = (float)(short)(d1 & 0xffffu); v[3] = (float)(short)(d1 >> 16);
    }
    if (dc) {
        const double di = dc[2 * ch], dq = dc[2 * ch + 1];
        v[0] = (float)((double)v[0] - di); v[1] = (float)((double)v[1] - dq);
        v[2] = (float)((double)v[2] - di); v[3] = (float)((double)v[3] - dq);
    }
    float4 *o = reinterpret_cast<float4 *>(out + 2 * ((long)ch * out_stride + q * per + 2 * j));
    *o = make_float4(v[0], v[1], v[2], v[3]);
}

hipError_t unpack_launch(const unsigned char *pk, long chan_stride_bytes, int channels, int npackets, int pkt_len,
                         float *out, long out_stride, const double *dc, hipStream_t stream)
{
    const int per = pkt_len == 1444 ? 240 : 256;
    const long tot = (long)npackets * (per / 2);
    if (tot == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((tot + 255) / 256), channels), dim3(256), 0, stream,
                       pk, chan_stride_bytes, npackets, pkt_len, per, out, out_stride, dc);
    return hipGetLastError();
}

// ---------------- NCO spur (DC) estimate ----------------
// y_n = (1-a)^n y_0 + a * sum_k (1-a)^(n-1-k) x_k : a weighted reduction, one workgroup per channel
__global__ __launch_bounds__(256)
void spurcal_kernel(const float *iq, long in_stride, int n, double *dc)
{
    __shared__ double red[2][256];
    const int ch = blockIdx.x, t = threadIdx.x;
    const f2 *x = reinterpret_cast<const f2 *>(iq) + (long)ch * in_stride;
    const double a = 1.0 / 100000.0, l1 = log1p(-a);
    double si = 0.0, sq = 0.0;
    for (long k = t; k < n; k += 256) {
        const double wgt = exp(l1 * (double)(n - 1 - k));
        const f2 v = x[k];
        si += wgt * (double)v.x; sq += wgt * (double)v.y;
    }
    red[0][t] = si; red[1][t] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        const double decay = exp(l1 * (double)n);
        dc[2 * ch] = decay * dc[2 * ch] + a * red[0][0];
        dc[2 * ch + 1] = decay * dc[2 * ch + 1] + a * red[1][0];
    }
}

hipError_t spurcal_launch(const float *iq, long in_stride, int channels, int n, double *dc, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(spurcal_kernel, dim3(channels), dim3(256), 0, stream, iq, in_stride, n, dc);
    return hipGetLastError();
}

}  // namespace csdr
