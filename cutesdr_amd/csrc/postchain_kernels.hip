// postchain_kernels.hip -- S-meter, AGC and demodulators for gfx950 (K4 in DESIGN.md).
//
// Replaces, per channel and per burst of band-pass output, the tail of
// CDemodulator::ProcessData (reference dsp/demodulator.cpp:182-207):
//   CSMeter::ProcessData (dsp/smeter.cpp:62-93) -> CAgc::ProcessData (dsp/agc.cpp:174-296)
//   -> C{Am,Sam,Fm,Ssb}Demod::ProcessData (dsp/amdemod.cpp:66-104, samdemod.cpp:78-158,
//   fmdemod.cpp:113-236, ssbdemod.cpp:48-60) with their CFir (dsp/fir.cpp:72-127) and CIir
//   (dsp/iir.cpp:171-201) helpers.
//
// The reference runs these as one per-sample loop full of running state.  Here one wave owns one
// channel and walks it in 1024-sample tiles staged in LDS; what is not a recurrence runs on all
// 64 lanes, linear recurrences are lane scans, the two piecewise-linear ones (AGC averagers, PLLs)
// are solved under a guess that is checked afterwards and walked sample by sample when the guess
// fails (see the comment blocks below and DESIGN.md K4).  The rate here is the decimated one
// (<= 78 kS/s per channel): the kernel is measured in cycles per sample, not against a roofline.
#include <hip/hip_runtime.h>
#include "postchain.h"

namespace csdr {

constexpr double kTwoPiD = 2.0 * 3.14159265358979323846;

// ---- CFir: y[n] = sum_k h[k] x[n-k] over a ring (dsp/fir.cpp:72-127) -----------------------------
__device__ __forceinline__ float fir_real(PcFir &f, float x)
{
    int p = f.pos + 1;
    if (p >= f.ntaps) p = 0;
    f.pos = p;
    f.zreal[p] = x;
    float acc = 0.f;
    int q = p;
    for (int k = 0; k < f.ntaps; k++) {
        acc += f.coef[k] * f.zreal[q];
        if (--q < 0) q = f.ntaps - 1;
    }
    return acc;
}
__device__ __forceinline__ void fir_cpx(PcFir &f, float &re, float &im)
{
    int p = f.pos + 1;
    if (p >= f.ntaps) p = 0;
    f.pos = p;
    f.zr[p] = re; f.zi[p] = im;
    float ar = 0.f, ai = 0.f;
    int q = p;
    for (int k = 0; k < f.ntaps; k++) {
        ar += f.icoef[k] * f.zr[q];
        ai += f.qcoef[k] * f.zi[q];
        if (--q < 0) q = f.ntaps - 1;
    }
    re = ar; im = ai;
}
// ---- CIir (dsp/iir.cpp:171-201) ------------------------------------------------------------------
__device__ __forceinline__ float iir_a(PcIir &f, float x)
{
    const double w0 = (double)x - f.a1 * f.w1a - f.a2 * f.w2a;
    const double y = f.b0 * w0 + f.b1 * f.w1a + f.b2 * f.w2a;
    f.w2a = f.w1a; f.w1a = w0;
    return (float)y;
}
__device__ __forceinline__ float iir_b(PcIir &f, float x)
{
    const double w0 = (double)x - f.a1 * f.w1b - f.a2 * f.w2b;
    const double y = f.b0 * w0 + f.b1 * f.w1b + f.b2 * f.w2b;
    f.w2b = f.w1b; f.w1b = w0;
    return (float)y;
}

// =====================================================================================================
// One wave per channel.  Everything that does not depend on the previous output sample (log
// magnitudes, sliding-window peak, gain law, delay line, arg(x), envelopes, FIR dot products) is
// computed by the 64 lanes in parallel over a tile of samples staged in LDS; only the genuinely
// recurrent scalars (S-meter and AGC averagers, PLL frequency/phase, DC blockers, squelch average,
// biquad) are walked sample by sample by lane 0, with their transcendentals hoisted out:
//   * the AGC peak  m_Peak  of agc.cpp:210-231 (compare, equality test, rescan) is exactly the
//     maximum of the last WindowSamples log-magnitudes, so it is a sliding-window maximum (log-step
//     doubling over [history | tile]);
//   * the PLL error  -atan2(rot(x, phi))  of fmdemod.cpp:166-172 / samdemod.cpp:83-89 equals
//     -wrap(arg(x) + sgn*phi), so arg(x) is taken for the whole tile up front and the loop carries
//     a dozen fp64 operations per sample and no transcendental.
// =====================================================================================================
constexpr int LC = 16;                   // samples per lane in the linear-recurrence scans
constexpr int BQ_TAB = 17 * 4 + 16 * 2;  // biquad chunk tables: M^k (k <= 16), c M^k (k < 16)
constexpr int PC_NCHUNK = (PC_AGC_RING + 1024) / 16, PC_RLEVELS = 8;
constexpr int PT = 1024;                 // tile length (samples)
constexpr int PH = PC_AGC_RING;          // longest history (AGC delay / window)
constexpr double kPiD = 3.14159265358979323846;

struct PcLds {
    float2 dl[PH + PT];                  // [last dly_n inputs | tile] : delay line, then AGC output in place
    float mg[PH + PT];                   // [last win_n-1 log-magnitudes | tile], doubled in place
    float pk[PT + 16];                    // sliding peak, then gain
    float w0[PT + PC_FIR_MAX + 17];           // [FIR history | tile] work array (audio / envelope / I)
    float w1[PT + PC_FIR_MAX + 17];           // second work array (theta / Q)
    float h0[PC_FIR_MAX + 5], h1[PC_FIR_MAX + 5];   // FIR taps of the active demodulator
    float w2[PT + 16];
    float rt[PC_RLEVELS][PC_NCHUNK];
    double pw_sm[LC + 1], pw_dc[LC + 1], pw_sq[LC + 1], pw_fd[LC + 1];   // powers of the averager coefficients
    double bq[BQ_TAB];                   // biquad chunk tables
    double pm[(LC + 1) * 4];             // PLL transition-matrix powers     // log table over the chunk maxima of the sliding peak                    // third work array (S-meter dB, PLL phase)
};

// acc[j] = sum_k h[k] * x[i_j - k] for this lane's outputs i_j = lane + 64 j; x points at the tile
// (ntaps-1 history samples sit in front of it), h and x in LDS.  Each tap is fetched once per lane
// and reused for the 16 outputs.
__device__ __forceinline__ void fir16(const float *h, int ntaps, const float *x, int lane, float (&acc)[16])
{
#pragma unroll
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    const float *p = x + lane;
#pragma unroll 3
    for (int k = 0; k < ntaps; k++) {
        const float hk = h[k];
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] += hk * p[64 * j - k];
    }
}
__device__ __forceinline__ double wrap_turn(double a) { return a - rint(a); }    // [-0.5, 0.5]
constexpr double kInvTwoPiD = 1.0 / (2.0 * 3.14159265358979323846);
// keep the last `hist` entries of [hist | n] in front for the next tile
__device__ __forceinline__ void slide(float *w, int hist, int n, int lane)
{
    float keep[2] = {0.f, 0.f};
    for (int j = 0; j < 2; j++) { const int i = lane + 64 * j; if (i < hist) keep[j] = w[n + i]; }
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < 2; j++) { const int i = lane + 64 * j; if (i < hist) w[i] = keep[j]; }
}

// Lane 0 walks src[0..n) in order, f(value, index); the next eight values are fetched from LDS
// while the current eight go through the recurrence, so no LDS latency sits on the dependent chain.
// src must be readable up to n+15.
template <class F>
__device__ __forceinline__ void seq_walk(const float *src, int n, F f)
{
    float cur[8], nxt[8];
#pragma unroll
    for (int j = 0; j < 8; j++) cur[j] = src[j];
    for (int i0 = 0; i0 < n; i0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) nxt[j] = src[i0 + 8 + j];
        if (i0 + 8 <= n) {
#pragma unroll
            for (int j = 0; j < 8; j++) f(cur[j], i0 + j);
        } else {
            for (int j = 0; j < 8; j++) if (i0 + j < n) f(cur[j], i0 + j);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = nxt[j];
    }
}


// =====================================================================================================
// Constant-coefficient LINEAR recurrences (averagers, DC blockers, the biquad) are not walked sample
// by sample: lane l takes the 16 consecutive samples [16 l, 16 l + 16), runs them from a zero state,
// the chunk-start states follow from a 64-lane scan over the affine chunk maps (a^cnt, p_last), and
// the homogeneous part a^(j+1) * S_start is added back.  ~100 instructions per lane and 1024-sample
// tile instead of 3-9 per sample on one lane.  Results differ from the sequential fp64 loop by
// rounding only.
// =====================================================================================================

__device__ __forceinline__ void pow_table(double *tab, double a, int lane)    // tab[k] = a^k, k = 0..16
{
    if (lane <= LC) { double p = 1.0; for (int k = 0; k < lane; k++) p *= a; tab[lane] = p; }
}

// s_i = a s_{i-1} + g x_i  over x[0..n), n <= 1024, s_{-1} = s0.  emit(i, x_i, s_i, s_{i-1}) for every
// sample (skipped when EMIT is false).  Returns s_{n-1} on every lane.
template <bool EMIT, class F>
__device__ __forceinline__ double lin1_scan(const float *x, int n, double a, double g, double s0,
                                            const double *apw, int lane, F emit)
{
    const int base = LC * lane;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    float xv[LC];
    double loc[LC], p = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        xv[j] = j < cnt ? x[base + j] : 0.f;
        p = j < cnt ? a * p + g * (double)xv[j] : p;
        loc[j] = p;
    }
    double A = apw[cnt], B = p;          // chunk map s -> A s + B; inclusive scan over lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double A1 = __shfl_up(A, d), B1 = __shfl_up(B, d);
        if (lane >= d) { B = A * B1 + B; A = A * A1; }
    }
    const double Ae = __shfl_up(A, 1), Be = __shfl_up(B, 1);
    const double S = lane == 0 ? s0 : Ae * s0 + Be;                  // state entering this lane's chunk
    if (EMIT) {
        double prev = S;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                const double sj = loc[j] + apw[j + 1] * S;
                emit(base + j, xv[j], sj, prev);
                prev = sj;
            }
        }
    }
    return __shfl(A, 63) * s0 + __shfl(B, 63);
}

// CSMeter (smeter.cpp:62-93) over one tile, final state only.  att is a plain averager; dec obeys
// dec' = max(att', (1-da) dec + da mag), and maps x -> max(A x + B, C) are closed under composition.
__device__ __forceinline__ void smeter_tile(PcSMeter &sm, const float *db, int n, const double *apw_att, int lane)
{
    const double aa = sm.att_a, ia = 1.0 - sm.att_a, da = sm.dec_a, id = 1.0 - sm.dec_a;
    const int base = LC * lane;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    float xv[LC];
    double loc[LC], p = 0.0, pk = -1.0e300;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        xv[j] = j < cnt ? db[base + j] : 0.f;
        p = j < cnt ? ia * p + aa * (double)xv[j] : p;
        loc[j] = p;
        if (j < cnt) pk = fmax(pk, (double)xv[j]);
    }
    double A = apw_att[cnt], B = p;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double A1 = __shfl_up(A, d), B1 = __shfl_up(B, d);
        if (lane >= d) { B = A * B1 + B; A = A * A1; }
    }
    const double Ae = __shfl_up(A, 1), Be = __shfl_up(B, 1);
    const double S = lane == 0 ? sm.att_ave : Ae * sm.att_ave + Be;
    const double att_end = __shfl(A, 63) * sm.att_ave + __shfl(B, 63);
    // chunk map of the decay average: x -> max(MA x + MB, MC)
    double MA = 1.0, MB = 0.0, MC = -1.0e300;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) {
            const double att = loc[j] + apw_att[j + 1] * S;          // updated attack average at this sample
            MA = id * MA; MB = id * MB + da * (double)xv[j]; MC = fmax(id * MC + da * (double)xv[j], att);
        }
    }
    // ordered reduction (lane 0 first): compose (earlier) then (later)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double A1 = __shfl_up(MA, d), B1 = __shfl_up(MB, d), C1 = __shfl_up(MC, d);
        if (lane >= d) { MC = fmax(MA * C1 + MB, MC); MB = MA * B1 + MB; MA = MA * A1; }
    }
    const double dec_end = fmax(__shfl(MA, 63) * sm.dec_ave + __shfl(MB, 63), __shfl(MC, 63));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pk = fmax(pk, __shfl_xor(pk, d));
    sm.att_ave = att_end; sm.dec_ave = dec_end; sm.ave_mag = dec_end; sm.peak_mag = fmax(sm.peak_mag, pk);
}

// CIir direct form II (iir.cpp:171-186) over x[0..n) in place.  State s = (w1, w2):
// s' = M s + (x, 0), y = b0 x + c . s, M = [[-a1, -a2], [1, 0]], c = (b1 - b0 a1, b2 - b0 a2).
// tab: M^k (4 doubles each, k = 0..16) then r_k = c M^k (2 doubles each, k = 0..15)
__device__ __forceinline__ void biquad_table(double *tab, const PcIir &f, int lane)
{
    if (lane == 0) {
        double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
        const double c0 = f.b1 - f.b0 * f.a1, c1 = f.b2 - f.b0 * f.a2;
        for (int k = 0; k <= LC; k++) {
            tab[4 * k] = m00; tab[4 * k + 1] = m01; tab[4 * k + 2] = m10; tab[4 * k + 3] = m11;
            if (k < LC) { tab[68 + 2 * k] = c0 * m00 + c1 * m10; tab[68 + 2 * k + 1] = c0 * m01 + c1 * m11; }
            const double n00 = -f.a1 * m00 - f.a2 * m10, n01 = -f.a1 * m01 - f.a2 * m11;     // M * M^k
            m10 = m00; m11 = m01; m00 = n00; m01 = n01;
        }
    }
}
__device__ __forceinline__ void biquad_scan(float *x, int n, PcIir &f, const double *tab, int lane)
{
    const int base = LC * lane;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    double y[LC], w1 = 0.0, w2 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const double xv = j < cnt ? (double)x[base + j] : 0.0;
        const double w0 = xv - f.a1 * w1 - f.a2 * w2;
        y[j] = f.b0 * w0 + f.b1 * w1 + f.b2 * w2;
        if (j < cnt) { w2 = w1; w1 = w0; }
    }
    // chunk map s -> M^cnt s + v
    double m00 = tab[4 * cnt], m01 = tab[4 * cnt + 1], m10 = tab[4 * cnt + 2], m11 = tab[4 * cnt + 3], v0 = w1, v1 = w2;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double p00 = __shfl_up(m00, d), p01 = __shfl_up(m01, d), p10 = __shfl_up(m10, d), p11 = __shfl_up(m11, d);
        const double q0 = __shfl_up(v0, d), q1 = __shfl_up(v1, d);
        if (lane >= d) {
            const double nv0 = m00 * q0 + m01 * q1 + v0, nv1 = m10 * q0 + m11 * q1 + v1;
            const double n00 = m00 * p00 + m01 * p10, n01 = m00 * p01 + m01 * p11;
            const double n10 = m10 * p00 + m11 * p10, n11 = m10 * p01 + m11 * p11;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11; v0 = nv0; v1 = nv1;
        }
    }
    const double e00 = __shfl_up(m00, 1), e01 = __shfl_up(m01, 1), e10 = __shfl_up(m10, 1), e11 = __shfl_up(m11, 1);
    const double ev0 = __shfl_up(v0, 1), ev1 = __shfl_up(v1, 1);
    const double S1 = lane == 0 ? f.w1a : e00 * f.w1a + e01 * f.w2a + ev0;
    const double S2 = lane == 0 ? f.w2a : e10 * f.w1a + e11 * f.w2a + ev1;
#pragma unroll
    for (int j = 0; j < LC; j++)
        if (j < cnt) x[base + j] = (float)(y[j] + tab[68 + 2 * j] * S1 + tab[68 + 2 * j + 1] * S2);
    const double f00 = __shfl(m00, 63), f01 = __shfl(m01, 63), f10 = __shfl(m10, 63), f11 = __shfl(m11, 63);
    const double fv0 = __shfl(v0, 63), fv1 = __shfl(v1, 63);
    const double nw1 = f00 * f.w1a + f01 * f.w2a + fv0, nw2 = f10 * f.w1a + f11 * f.w2a + fv1;
    f.w1a = nw1; f.w2a = nw2;
}

// CAgc's attack / decay averagers (agc.cpp:233-262):  ave += alpha (pk - ave),  alpha = rise when
// pk > ave else fall.  Piecewise linear, so: guess the selector bits, solve the then-linear
// recurrence with a lane scan, recompute the selectors from the solution, repeat until they
// reproduce themselves -- at that point the sequence IS the sequential one (each selector was
// taken against the true previous average).  The correct prefix grows every round; the peak moves
// slowly, so two or three rounds are typical.  Returns false after PC_AGC_ROUNDS without a fixed
// point (the caller then walks the tile sample by sample).
constexpr int PC_AGC_ROUNDS = 8;
template <class F>
__device__ __forceinline__ bool agc_ave_scan(const float *pk, int n, double rise, double fall, double &ave,
                                             int lane, F emit)
{
    const int base = LC * lane;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    const double ave0 = ave;
    float pv[LC];
    unsigned sel = 0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        pv[j] = j < cnt ? pk[base + j] : 0.f;
        if (j < cnt && (double)pv[j] > ave0) sel |= 1u << j;
    }
    for (int round = 0; round < PC_AGC_ROUNDS; round++) {
        double A = 1.0, B = 0.0;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                const double al = (sel >> j & 1) ? rise : fall;
                A = A - al * A;
                B = B + al * ((double)pv[j] - B);
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double A1 = __shfl_up(A, d), B1 = __shfl_up(B, d);
            if (lane >= d) { B = A * B1 + B; A = A * A1; }
        }
        const double Ae = __shfl_up(A, 1), Be = __shfl_up(B, 1);
        double x = lane == 0 ? ave0 : Ae * ave0 + Be;
        double val[LC];
        unsigned nsel = 0;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                if ((double)pv[j] > x) nsel |= 1u << j;
                const double al = (sel >> j & 1) ? rise : fall;
                x = x + al * ((double)pv[j] - x);
            }
            val[j] = x;
        }
        if (!__any(nsel != sel)) {
#pragma unroll
            for (int j = 0; j < LC; j++) if (j < cnt) emit(base + j, val[j]);
            ave = __shfl(A, 63) * ave0 + __shfl(B, 63);
            return true;
        }
        sel = nsel;
    }
    return false;
}

// The second-order PLL of CFmDemod / CSamDemod (fmdemod.cpp:166-177, samdemod.cpp:83-97), in turns:
//   e = -wrap(theta_i + phi),  f += beta e (clamped to [lo, hi]),  phi += f + alpha e.
// With the input phase unwrapped (Theta_i, a prefix sum of wrapped differences) a locked loop keeps
// Theta_i + phi next to ONE integer K for a whole tile and never touches the clamp; under that
// guess e = (K - Theta_i) - phi and the loop is the constant-coefficient linear system
//   [phi; f] <- [[1-alpha-beta, 1], [-beta, 1]] [phi; f] + (alpha+beta, beta) (K - Theta_i),
// solved by a lane scan like the biquad.  The guess is then checked sample by sample (|e| < 1/2, f
// inside the clamp); if it holds everywhere the result is the sequential one, otherwise (cycle slip,
// acquisition, noise) the caller walks the tile.  emit(i, phi_before, f_after).
// tab: M^k, k = 0..16 (4 doubles each)
__device__ __forceinline__ void pll_table(double *tab, double alpha, double beta, int lane)
{
    if (lane == 0) {
        const double a00 = 1.0 - alpha - beta, a01 = 1.0, a10 = -beta, a11 = 1.0;
        double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
        for (int k = 0; k <= LC; k++) {
            tab[4 * k] = m00; tab[4 * k + 1] = m01; tab[4 * k + 2] = m10; tab[4 * k + 3] = m11;
            const double n00 = a00 * m00 + a01 * m10, n01 = a00 * m01 + a01 * m11;
            const double n10 = a10 * m00 + a11 * m10, n11 = a10 * m01 + a11 * m11;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
        }
    }
}
template <class F>
__device__ __forceinline__ bool pll_scan(const float *th, int n, double alpha, double beta, double lo, double hi,
                                         double &ph, double &fr, const double *tab, int lane, F emit)
{
    const int base = LC * lane;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    float tv[LC];
#pragma unroll
    for (int j = 0; j < LC; j++) tv[j] = j < cnt ? th[base + j] : 0.f;
    // unwrapped input phase relative to the first sample of the tile
    float last = tv[0];
#pragma unroll
    for (int j = 1; j < LC; j++) if (j < cnt) last = tv[j];
    const float before = __shfl_up(last, 1);
    double c[LC], run = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const float pv = j == 0 ? (lane == 0 ? tv[0] : before) : tv[j - 1];
        const float d = tv[j] - pv;
        if (j < cnt) run += (double)(d - rintf(d));
        c[j] = run;
    }
    double incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const double first = (double)__shfl(tv[0], 0);
    const double K = rint(first + ph);
    const double off = K - first - (incl - run);
#pragma unroll
    for (int j = 0; j < LC; j++) c[j] = off - c[j];                  // K - Theta_i
    // zero-state chunk response
    const double ab = alpha + beta;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) {
            const double e = c[j] - v0;
            v1 = v1 + beta * e;
            v0 = v0 + v1 + alpha * e;
        }
    }
    double m00 = tab[4 * cnt], m01 = tab[4 * cnt + 1], m10 = tab[4 * cnt + 2], m11 = tab[4 * cnt + 3];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double p00 = __shfl_up(m00, d), p01 = __shfl_up(m01, d), p10 = __shfl_up(m10, d), p11 = __shfl_up(m11, d);
        const double q0 = __shfl_up(v0, d), q1 = __shfl_up(v1, d);
        if (lane >= d) {
            const double nv0 = m00 * q0 + m01 * q1 + v0, nv1 = m10 * q0 + m11 * q1 + v1;
            const double n00 = m00 * p00 + m01 * p10, n01 = m00 * p01 + m01 * p11;
            const double n10 = m10 * p00 + m11 * p10, n11 = m10 * p01 + m11 * p11;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11; v0 = nv0; v1 = nv1;
        }
    }
    (void)ab;
    const double e00 = __shfl_up(m00, 1), e01 = __shfl_up(m01, 1), e10 = __shfl_up(m10, 1), e11 = __shfl_up(m11, 1);
    const double ev0 = __shfl_up(v0, 1), ev1 = __shfl_up(v1, 1);
    double x0 = lane == 0 ? ph : e00 * ph + e01 * fr + ev0;           // state entering this lane's chunk
    double x1 = lane == 0 ? fr : e10 * ph + e11 * fr + ev1;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) {
            const double e = c[j] - x0;
            const double f = x1 + beta * e;
            bad = bad || !(fabs(e) < 0.4999) || !(f >= lo && f <= hi);
            emit(base + j, x0, f);
            x1 = f;
            x0 = x0 + f + alpha * e;
        }
    }
    if (__any(bad)) return false;
    const double f00 = __shfl(m00, 63), f01 = __shfl(m01, 63), f10 = __shfl(m10, 63), f11 = __shfl(m11, 63);
    const double fv0 = __shfl(v0, 63), fv1 = __shfl(v1, 63);
    const double nph = f00 * ph + f01 * fr + fv0, nfr = f10 * ph + f11 * fr + fv1;
    ph = nph - rint(nph); fr = nfr;
    return true;
}

#define PC_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// pk[i] = max(E[i .. i+W1]), i < n, for E = S.mg[0 .. W1+n).  Chunks of 16: per-chunk prefix and
// suffix maxima (registers), a log table over the <= 192 chunk maxima for the chunks strictly
// inside a window, then window = suffix(first chunk) | inner chunks | prefix(last chunk).
// Only prefix values at tile positions (-> S.pk) and suffix values at i < n (-> S.w2) are kept.
constexpr float kNegBig = -1.0e30f;
__device__ __forceinline__ void sliding_max(PcLds &S, int W1, int n, int lane)
{
    const int len = W1 + n;
    if (W1 < 16) {                                      // short windows: direct
        for (int i = lane; i < n; i += 64) {
            float v = S.mg[i];
            for (int k = 1; k <= W1; k++) v = fmaxf(v, S.mg[i + k]);
            S.pk[i] = v;
        }
        PC_SYNC();
        return;
    }
    const int nc = (len + 15) >> 4;
    for (int c = lane; c < PC_NCHUNK; c += 64) {
        float e[16];
        const float4 *src = reinterpret_cast<const float4 *>(S.mg + 16 * c);
        if (c < nc) {
#pragma unroll
            for (int q = 0; q < 4; q++) { const float4 v = src[q]; e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
#pragma unroll
            for (int j = 0; j < 16; j++) if (16 * c + j >= len) e[j] = kNegBig;
            float pre[16], suf[16];
            pre[0] = e[0]; suf[15] = e[15];
#pragma unroll
            for (int j = 1; j < 16; j++) { pre[j] = fmaxf(pre[j - 1], e[j]); suf[15 - j] = fmaxf(suf[16 - j], e[15 - j]); }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int idx = 16 * c + j;
                const int ip = idx - W1;                                  // prefix wanted at tile positions
                S.pk[(ip >= 0 && ip < n) ? ip : PT + 15] = pre[j];
                S.w2[idx < n ? idx : PT + 15] = suf[j];
            }
            S.rt[0][c] = pre[15];
        } else {
            S.rt[0][c] = kNegBig;
        }
    }
    PC_SYNC();
    for (int l = 1; l < PC_RLEVELS; l++) {
        const int h = 1 << (l - 1);
        for (int c = lane; c < PC_NCHUNK; c += 64) {
            const float v0 = S.rt[l - 1][c];
            const float v1 = (c + h < PC_NCHUNK) ? S.rt[l - 1][c + h] : kNegBig;
            S.rt[l][c] = fmaxf(v0, v1);
        }
        PC_SYNC();
    }
    for (int i = lane; i < n; i += 64) {
        const int ci = i >> 4, ce = (i + W1) >> 4, cnt = ce - ci - 1;     // W1 >= 16: ce > ci
        float v = fmaxf(S.w2[i], S.pk[i]);
        if (cnt > 0) {
            const int l = 31 - __clz(cnt);
            v = fmaxf(v, fmaxf(S.rt[l][ci + 1], S.rt[l][ce - (1 << l)]));
        }
        S.pk[i] = v;
    }
    PC_SYNC();
}

__global__ __launch_bounds__(64)
void postchain_kernel(PcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    PcLds &S = *reinterpret_cast<PcLds *>(pc_smem);
    const int ch = blockIdx.x, lane = threadIdx.x;
    PcChannel &C = a.chan[ch];
    float *g_dly = a.agc_dly + (long)ch * PC_AGC_RING * 2;      // linear: last dly_n inputs, oldest first
    float *g_mag = a.agc_mag + (long)ch * PC_AGC_RING;          // linear: last win_n-1 magnitudes
    const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)ch * a.in_stride;
    const bool stereo = a.flags & PC_STEREO;
    const long orow = (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    float *outm = a.out ? a.out + orow : nullptr;                               // mono
    float2 *outs = a.out ? reinterpret_cast<float2 *>(a.out) + orow : nullptr;  // stereo / complex
    const int mode = (a.flags & PC_DO_DEMOD) ? C.mode : PC_MODE_NONE;
    const bool do_sm = a.flags & PC_DO_SMETER, do_agc = a.flags & PC_DO_AGC, agc_real = a.flags & PC_AGC_REAL;
    const bool cpx_out = stereo || mode == PC_MODE_NONE;

    // scalar state (only lane 0's copy is meaningful and written back)
    PcSMeter sm = C.sm;
    PcAgc agc = C.agc;
    const int D = agc.dly_n > 0 ? agc.dly_n : 1, W1 = agc.win_n > 0 ? agc.win_n - 1 : 0;
    double am_z1 = C.am.z1;
    double sam_z1 = C.sam.z1, sam_y1 = C.sam.y1, sam_ph = C.sam.phase, sam_fr = C.sam.freq;
    double fm_ph = C.fm.phase, fm_fr = C.fm.freq, fm_dc = C.fm.err_dc, fm_sq = C.fm.sq_ave;
    int fm_squelched = C.fm.squelched;
    PcIir lp = C.fm.lp;

    // histories -> LDS
    if (do_agc && agc.on) {
        for (int i = lane; i < D; i += 64) S.dl[i] = make_float2(g_dly[2 * i], g_dly[2 * i + 1]);
        for (int i = lane; i < W1; i += 64) S.mg[i] = g_mag[i];
    }
    const PcFir *fir = mode == PC_MODE_AM ? &C.am.fir : mode == PC_MODE_SAM ? &C.sam.fir : mode == PC_MODE_FM ? &C.fm.hp : nullptr;
    const int nt = fir ? fir->ntaps : 1;
    if (fir) {
        for (int i = lane; i < nt; i += 64) {
            S.h0[i] = (mode == PC_MODE_FM || !stereo) ? fir->coef[i] : fir->icoef[i];
            S.h1[i] = fir->qcoef[i];
        }
        for (int i = lane; i < nt - 1; i += 64) { S.w0[i] = (mode == PC_MODE_FM || mode == PC_MODE_AM && !stereo) ? fir->zreal[i] : fir->zr[i]; S.w1[i] = fir->zi[i]; }
    }
    pow_table(S.pw_sm, 1.0 - sm.att_a, lane);
    pow_table(S.pw_dc, 0.99, lane);
    pow_table(S.pw_sq, 1.0 - C.fm.sq_alpha, lane);
    pow_table(S.pw_fd, 1.0 - C.fm.dc_alpha, lane);
    biquad_table(S.bq, lp, lane);
    if (mode == PC_MODE_FM) pll_table(S.pm, C.fm.alpha, C.fm.beta, lane);
    if (mode == PC_MODE_SAM) pll_table(S.pm, C.sam.alpha, C.sam.beta, lane);
    PC_SYNC();

#ifdef PC_PROFILE
    unsigned long long tk[10] = {0}, tlast = __builtin_readcyclecounter();
#define PC_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define PC_TICK(k)
#endif
    for (int b = 0; b < a.nbursts; b++) {
        for (int t0 = 0; t0 < a.burst; t0 += PT) {
            PC_TICK(7);
            const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
            const long gi = (long)b * a.burst + t0;
            float2 *x = S.dl + ((do_agc && agc.on) ? D : 0);           // tile samples (AGC: behind the delay history)
            for (int i = lane; i < n; i += 64) x[i] = in[gi + i];
            PC_SYNC();
            // ---------------- S-meter (smeter.cpp:62-93) ----------------
            if (do_sm) {
                for (int i = lane; i < n; i += 64) {
                    const float pw = (x[i].x * x[i].x + x[i].y * x[i].y) * (1.0f / (32767.0f * 32767.0f));
                    S.w2[i] = pw > 0.f ? 10.0f * log10f(pw) : -500.0f;
                }
                PC_SYNC();
                smeter_tile(sm, S.w2, n, S.pw_sm, lane);
            }
            // ---------------- AGC (agc.cpp:174-296 / 301-401) ----------------
            if (do_agc) {
                if (!agc.on) {
                    const float g = (float)agc.manual_gain;
                    for (int i = lane; i < n; i += 64) { x[i].x *= g; x[i].y *= g; }
                } else {
                    float *mg = S.mg + W1;
                    for (int i = lane; i < n; i += 64) {
                        float m = fabsf(x[i].x);
                        if (!agc_real) { const float mi = fabsf(x[i].y); if (mi > m) m = mi; }
                        mg[i] = log10f(m + 3.2767e-4f) - 4.51543987f;
                    }
                    PC_SYNC();
            PC_TICK(1);
                    // sliding maximum: pk[i] = max E[i .. i+W1], E = [W1 history | tile] = S.mg
                    sliding_max(S, W1, n, lane);
                    // the last W1 magnitudes are the next tile's history (forward move, 64 at a time)
                    for (int i0 = 0; i0 < W1; i0 += 64) {
                        const int i = i0 + lane;
                        const float v = S.mg[(i < W1 ? i : 0) + n];
                        PC_SYNC();
                        if (i < W1) S.mg[i] = v;
                        PC_SYNC();
                    }
                    // attack / decay averagers -> log gain argument max(att, dec) per sample in S.pk
                    {
                        double att = agc.attack_ave, dec = agc.decay_ave;
                        bool ok = !agc.hang;                              // the hang timer is a counter: walked
                        if (ok) ok = agc_ave_scan(S.pk, n, agc.att_rise, agc.att_fall, att, lane,
                                                  [&](int i, double v) { S.w2[i] = (float)v; });
                        if (ok) ok = agc_ave_scan(S.pk, n, agc.dec_rise, agc.dec_fall, dec, lane,
                                                  [&](int i, double v) { S.pk[i] = fmaxf(S.w2[i], (float)v); });
                        if (ok) { agc.attack_ave = att; agc.decay_ave = dec; }
                        else {
                            PC_SYNC();
                            att = agc.attack_ave; dec = agc.decay_ave;
                            int timer = agc.hang_timer;
                            if (lane == 0) {
                                const double ar = agc.att_rise, af = agc.att_fall, dr = agc.dec_rise, df = agc.dec_fall;
                                const bool hang = agc.hang;
                                const int hang_time = agc.hang_time;
                                float *dst = S.pk;
                                seq_walk(S.pk, n, [&](float v, int i) {
                                    const double pk = v;
                                    const double da = pk - att, dd = pk - dec;
                                    att = att + (da > 0.0 ? ar : af) * da;
                                    const bool up = dd > 0.0, hold = hang && !up && timer < hang_time;
                                    dec = dec + (up ? dr : (hold ? 0.0 : df)) * dd;
                                    if (hang) timer = up ? 0 : (hold ? timer + 1 : timer);
                                    dst[i] = (float)fmax(att, dec);
                                });
                            }
                            agc.attack_ave = __shfl(att, 0); agc.decay_ave = __shfl(dec, 0); agc.hang_timer = __shfl(timer, 0);
                        }
                    }
                    PC_SYNC();
            PC_TICK(3);
                    // gain law + delay line: out[i] = in[i - D] * gain[i]; S.dl = [D old | n new]
                    const float knee = (float)agc.knee, slm1 = (float)(agc.gain_slope - 1.0), fixed_gain = (float)agc.fixed_gain;
                    float2 outv[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int i = lane + 64 * j;
                        if (i < n) {
                            const float m = S.pk[i];
                            const float g = (m <= knee) ? fixed_gain : 0.7f * exp10f(m * slm1);
                            const float2 d = S.dl[i];
                            outv[j] = make_float2(d.x * g, d.y * g);
                        }
                    }
                    float2 keepd[32];
#pragma unroll
                    for (int j = 0; j < 32; j++) { const int i = lane + 64 * j; if (i < D) keepd[j] = S.dl[n + i]; }
                    PC_SYNC();
#pragma unroll
                    for (int j = 0; j < 32; j++) { const int i = lane + 64 * j; if (i < D) S.dl[i] = keepd[j]; }
#pragma unroll
                    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < n) x[i] = outv[j]; }
                    PC_SYNC();
                }
            }
            // x[0..n) now holds the AGC output (or the input); x = S.dl + D
            PC_TICK(4);
            // ---------------- demodulators ----------------
            if (mode == PC_MODE_NONE || mode >= PC_MODE_USB) {
                if (a.out) {
                    for (int i = lane; i < n; i += 64) {
                        if (cpx_out) outs[gi + i] = x[i];
                        else outm[gi + i] = x[i].x;                       // ssbdemod.cpp:48-53
                    }
                }
            } else if (mode == PC_MODE_AM) {
                float *w = S.w0 + (nt - 1);
                for (int i = lane; i < n; i += 64) w[i] = sqrtf(x[i].x * x[i].x + x[i].y * x[i].y);
                PC_SYNC();
                // DC block z0 = x + 0.99 z1, out = z0 - z1 (amdemod.cpp:70-80)
                am_z1 = lin1_scan<true>(w, n, 0.99, 1.0, am_z1, S.pw_dc, lane,
                                        [&](int i, float, double z0, double z1) { w[i] = (float)(z0 - z1); });
                PC_SYNC();
                float acc[16], acq[16];
                fir16(S.h0, nt, w, lane, acc);
                if (stereo) {
                    fir16(S.h1, nt, w, lane, acq);
#pragma unroll
                    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < n) outs[gi + i] = make_float2(acc[j], acq[j]); }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) { const int i = lane + 64 * j; if (i < n) outm[gi + i] = acc[j]; }
                }
                PC_SYNC();
                slide(S.w0, nt - 1, n, lane);
                PC_SYNC();
            } else {
                // PLL modes: theta = arg(x), r = |x| for the whole tile
                float *th = S.w1 + (nt - 1), *au = S.w0 + (nt - 1);
                for (int i = lane; i < n; i += 64) th[i] = atan2f(x[i].y, x[i].x) * (float)kInvTwoPiD;   // turns
                PC_SYNC();
                if (mode == PC_MODE_FM) {
                    const PcFm &F = C.fm;
                    bool scanned;
                    {
                        double ph = fm_ph * kInvTwoPiD, fr = fm_fr * kInvTwoPiD;
                        scanned = pll_scan(th, n, F.alpha, F.beta, F.lo * kInvTwoPiD, F.hi * kInvTwoPiD, ph, fr, S.pm, lane,
                                           [&](int i, double, double f) { au[i] = (float)f; });
                        if (scanned) { fm_ph = ph * kTwoPiD; fm_fr = fr * kTwoPiD; }
                    }
                    if (!scanned) PC_SYNC();
                    if (!scanned && lane == 0) {
                        // phase, frequency and error in turns: wrapping is a - rint(a)
                        const double beta = F.beta, alpha = F.alpha, hi = F.hi * kInvTwoPiD, lo = F.lo * kInvTwoPiD;
                        double ph = fm_ph * kInvTwoPiD, fr = fm_fr * kInvTwoPiD;
                        seq_walk(th, n, [&](float v, int i) {              // fmdemod.cpp:166-177
                            const double err = -wrap_turn((double)v + ph);
                            fr = fmin(fmax(fr + beta * err, lo), hi);
                            ph = wrap_turn(ph + fr + alpha * err);
                            au[i] = (float)fr;                             // NCO frequency, turns per sample
                        });
                        fm_ph = ph * kTwoPiD; fm_fr = fr * kTwoPiD;
                    }
                    if (!scanned) { fm_ph = __shfl(fm_ph, 0); fm_fr = __shfl(fm_fr, 0); }
                    PC_SYNC();
                    {   // audio = (freq - its running mean) * gain  (fmdemod.cpp:178-186): the mean is linear
                        const double og = F.out_gain * kTwoPiD;
                        fm_dc = kTwoPiD * lin1_scan<true>(au, n, 1.0 - F.dc_alpha, F.dc_alpha, fm_dc * kInvTwoPiD, S.pw_fd, lane,
                                    [&](int i, float f, double dc, double) { au[i] = (float)(((double)f - dc) * og); });
                    }
                    PC_SYNC();
            PC_TICK(5);
                    // raw audio to the output row; squelch is decided at the end of the burst
                    for (int i = lane; i < n; i += 64) { if (stereo) outs[gi + i] = make_float2(au[i], au[i]); else outm[gi + i] = au[i]; }
                    if (a.burst <= 16384) {                               // MAX_SQBUF_SIZE
            PC_TICK(6);
                        float acc[16];
                        fir16(S.h0, nt, au, lane, acc);
#pragma unroll
                        for (int j = 0; j < 16; j++) S.w2[(lane + 64 * j) & (PT - 1)] = fabsf(acc[j]);
                        PC_SYNC();
                        fm_sq = lin1_scan<false>(S.w2, n, 1.0 - F.sq_alpha, F.sq_alpha, fm_sq, S.pw_sq, lane,
                                                 [](int, float, double, double) {});
                    }
                    PC_SYNC();
                    slide(S.w0, nt - 1, n, lane);
                    PC_SYNC();
                } else {                                                  // SAM, samdemod.cpp:78-158
                    const PcSam &M = C.sam;
                    const double sgn = stereo ? 1.0 : -1.0;
                    bool scanned;
                    {   // in psi = sgn phi, g = sgn f the loop has the FM form with theta as is
                        double ps = sgn * sam_ph * kInvTwoPiD, g = sgn * sam_fr * kInvTwoPiD;
                        const double l0 = M.lo * kInvTwoPiD, h0 = M.hi * kInvTwoPiD;
                        scanned = pll_scan(th, n, M.alpha, M.beta, sgn > 0 ? l0 : -h0, sgn > 0 ? h0 : -l0, ps, g, S.pm, lane,
                                           [&](int i, double p, double) { S.w2[i] = (float)(sgn * (p - rint(p))); });
                        if (scanned) { sam_ph = sgn * ps * kTwoPiD; sam_fr = sgn * g * kTwoPiD; }
                    }
                    if (!scanned) PC_SYNC();
                    if (!scanned && lane == 0) {
                        const double beta = M.beta, alpha = M.alpha, hi = M.hi * kInvTwoPiD, lo = M.lo * kInvTwoPiD;
                        double ph = sam_ph * kInvTwoPiD, fr = sam_fr * kInvTwoPiD;
                        seq_walk(th, n, [&](float v, int i) {
                            S.w2[i] = (float)ph;                          // phase used for this sample (turns)
                            const double err = -sgn * wrap_turn((double)v + sgn * ph);
                            fr = fmin(fmax(fr + beta * err, lo), hi);
                            ph = wrap_turn(ph + fr + alpha * err);
                        });
                        sam_ph = ph * kTwoPiD; sam_fr = fr * kTwoPiD;
                    }
                    if (!scanned) { sam_ph = __shfl(sam_ph, 0); sam_fr = __shfl(sam_fr, 0); }
                    PC_SYNC();
                    // rotated sample tr + j ti = |x| e^{j(theta + sgn phi)}
                    for (int i = lane; i < n; i += 64) {
                        const float r = sqrtf(x[i].x * x[i].x + x[i].y * x[i].y);
                        float s, c;
                        sincospif(2.0f * (th[i] + (float)sgn * S.w2[i]), &s, &c);
                        au[i] = r * c;                                    // tr
                        th[i] = r * s;                                    // ti
                    }
                    PC_SYNC();
                    // DC blocks
                    sam_z1 = lin1_scan<true>(au, n, 0.99, 1.0, sam_z1, S.pw_dc, lane,
                                             [&](int i, float, double z0, double z1) { au[i] = (float)(z0 - z1); });
                    if (stereo)
                        sam_y1 = lin1_scan<true>(th, n, 0.99, 1.0, sam_y1, S.pw_dc, lane,
                                                 [&](int i, float, double y0, double y1) { th[i] = (float)(y0 - y1); });
                    PC_SYNC();
                    if (!stereo) {
                        for (int i = lane; i < n; i += 64) outm[gi + i] = au[i];
                    } else {
                        float ar[16], ai[16];
                        fir16(S.h0, nt, au, lane, ar);
                        fir16(S.h1, nt, th, lane, ai);
#pragma unroll
                        for (int j = 0; j < 16; j++) {                    // lower sideband left, upper right
                            const int i = lane + 64 * j;
                            if (i < n) outs[gi + i] = make_float2(ar[j] + ai[j], ar[j] - ai[j]);
                        }
                        PC_SYNC();
                        slide(S.w0, nt - 1, n, lane);
                        slide(S.w1, nt - 1, n, lane);
                    }
                    PC_SYNC();
                }
            }
        }
            PC_TICK(8);
        // ---------------- end of burst: FM squelch decision (fmdemod.cpp:128-151) ----------------
        if (mode == PC_MODE_FM && a.burst <= 16384) {
            const PcFm &F = C.fm;
            if (0 == F.sq_thresh) fm_squelched = 1;
            else if (fm_squelched) { if (fm_sq < (F.sq_thresh - 100.0)) fm_squelched = 0; }
            else { if (fm_sq >= (F.sq_thresh + 100.0)) fm_squelched = 1; }
            fm_squelched = __shfl(fm_squelched, 0);
            const long g0 = (long)b * a.burst;
            PC_SYNC();
            for (int t0 = 0; t0 < a.burst; t0 += PT) {
                const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
                if (fm_squelched) {
                    for (int i = lane; i < n; i += 64) { if (stereo) outs[g0 + t0 + i] = make_float2(0.f, 0.f); else outm[g0 + t0 + i] = 0.f; }
                } else {                                                  // low-pass biquad over the burst
                    for (int i = lane; i < n; i += 64) S.w2[i] = stereo ? outs[g0 + t0 + i].x : outm[g0 + t0 + i];
                    PC_SYNC();
                    biquad_scan(S.w2, n, lp, S.bq, lane);
                    PC_SYNC();
                    for (int i = lane; i < n; i += 64) { const float y = S.w2[i]; if (stereo) outs[g0 + t0 + i] = make_float2(y, y); else outm[g0 + t0 + i] = y; }
                    PC_SYNC();
                }
            }
        }
    }

            PC_TICK(9);
#ifdef PC_PROFILE
    if (lane == 0 && ch == 0)
        printf("pcprof mode %d: load+smeter %llu mag %llu slide %llu agcseq %llu gain %llu demod-pre+pll %llu rawout %llu fir+ema+slide %llu burstend %llu (x100 cycles)\n",
               mode, tk[0] / 100, tk[1] / 100, tk[2] / 100, tk[3] / 100, tk[4] / 100, tk[5] / 100, tk[6] / 100, tk[7] / 100, tk[8] / 100);
#endif
    // ---------------- write the state back ----------------
    PC_SYNC();
    if (do_agc && agc.on) {
        for (int i = lane; i < D; i += 64) { g_dly[2 * i] = S.dl[i].x; g_dly[2 * i + 1] = S.dl[i].y; }
        for (int i = lane; i < W1; i += 64) g_mag[i] = S.mg[i];
    }
    if (fir) {
        PcFir *fw = const_cast<PcFir *>(fir);
        for (int i = lane; i < nt - 1; i += 64) {
            if (mode == PC_MODE_FM || (mode == PC_MODE_AM && !stereo)) fw->zreal[i] = S.w0[i];
            else { fw->zr[i] = S.w0[i]; fw->zi[i] = mode == PC_MODE_AM ? S.w0[i] : S.w1[i]; }
        }
    }
    // each stage writes only its own state: the stages of one channel may run as separate,
    // concurrent launches (S-meter | AGC | demodulator pipeline of the batch chain)
    if (lane == 0) {
        if (do_sm) C.sm = sm;
        if (do_agc) C.agc = agc;
        if (mode == PC_MODE_AM) C.am.z1 = am_z1;
        if (mode == PC_MODE_SAM) { C.sam.z1 = sam_z1; C.sam.y1 = sam_y1; C.sam.phase = sam_ph; C.sam.freq = sam_fr; }
        if (mode == PC_MODE_FM) {
            C.fm.phase = fm_ph; C.fm.freq = fm_fr; C.fm.err_dc = fm_dc; C.fm.sq_ave = fm_sq; C.fm.squelched = fm_squelched;
            C.fm.lp = lp;
        }
    }
}

// stand-alone CFir / CIir objects (one lane each): op 0 FIR real, 1 FIR complex, 2 IIR real, 3 IIR complex
__global__ void filter_leaf_kernel(PcFir *fir, PcIir *iir, const float *in, float *out, int n, int op)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == 0) {
        for (int i = 0; i < n; i++) out[i] = fir_real(*fir, in[i]);
    } else if (op == 1) {
        for (int i = 0; i < n; i++) {
            float re = in[2 * i], im = in[2 * i + 1];
            fir_cpx(*fir, re, im);
            out[2 * i] = re; out[2 * i + 1] = im;
        }
    } else if (op == 2) {
        for (int i = 0; i < n; i++) out[i] = iir_a(*iir, in[i]);
    } else {
        for (int i = 0; i < n; i++) { out[2 * i] = iir_a(*iir, in[2 * i]); out[2 * i + 1] = iir_b(*iir, in[2 * i + 1]); }
    }
}

hipError_t filter_leaf_launch(PcFir *fir, PcIir *iir, const float *in, float *out, int n, int op, hipStream_t stream)
{
    hipLaunchKernelGGL(filter_leaf_kernel, dim3(1), dim3(64), 0, stream, fir, iir, in, out, n, op);
    return hipGetLastError();
}

hipError_t postchain_launch(const PcArgs &a, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&postchain_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PcLds));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(postchain_kernel, dim3(a.channels), dim3(64), sizeof(PcLds), stream, a);
    return hipGetLastError();
}

}  // namespace csdr
