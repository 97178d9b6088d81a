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
#include "ref_constants.hpp"
#include <cstdlib>
#include "launch_once.hpp"

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
// One workgroup of NW waves (1 or 4) per channel, 1024-sample tiles staged in LDS.  Thread t owns the
// LC = 1024 / (64 NW) consecutive samples [LC t, LC t + LC) in every recurrence; element-wise work is
// strided over the workgroup.
// =====================================================================================================
constexpr int PT = 1024;                 // tile length (samples)
constexpr int PH = PC_AGC_RING;          // longest history (AGC delay / window)
constexpr int PC_NCHUNK = (PC_AGC_RING + PT) / 16, PC_RLEVELS = 8;
constexpr int LCMAX = 16;                // samples per thread at NW = 1
constexpr int BQ_TAB = (LCMAX + 1) * 4 + LCMAX * 2;   // biquad chunk tables: M^k (k <= 16), c M^k (k < 16)
constexpr double kInvTwoPiD = 1.0 / (2.0 * 3.14159265358979323846);
constexpr int PC_PLL_WARM_MAX = 192;     // longest warm-up of the overlapped PLL walks (pll_overlap) worth running
constexpr float kNegBig = -1.0e30f;

// what a workgroup's scans and broadcasts exchange through LDS (Wg<NW> holds a pointer to it)
struct alignas(16) PcSync {               // (16: the arrays behind it are read and written in 16-byte pieces)
    double xch[2][8][8];                 // per-wave totals of a workgroup scan; two banks used in turn, so that a scan
                                         // needs ONE workgroup barrier (the next scan's writes go to the other bank)
    double bc[4];                        // broadcast slot (thread 0 -> workgroup)
    int flag;                            // workgroup-wide "any"
};

struct PcLds {
    PcSync sy;
    alignas(16) float2 dl[PH + PT];                  // [last dly_n inputs | tile]: delay line, then the AGC output in place
    alignas(16) float mg[PH + PT];                   // [last win_n-1 log-magnitudes | tile]
    alignas(16) float pk[PT + 16];                   // sliding peak, then the log gain argument
    alignas(16) float w0[PT + PC_FIR_MAX + 17];      // [FIR history | tile] work array (audio / envelope / I)
    alignas(16) float w1[PT + PC_FIR_MAX + 17];      // second work array (theta / Q)
    alignas(16) float h0[PC_FIR_MAX + 17], h1[PC_FIR_MAX + 17]; // FIR taps of the active demodulator, reversed and zero padded:
                                                    // h[4 + r] = tap ntaps-1-r (16-byte aligned rows of four)
    alignas(16) float w2[PT + 16];       // third work array (S-meter dB, attack average, PLL phase, |hp|)
    alignas(16) float rt[PC_RLEVELS][PC_NCHUNK];   // log table over the chunk maxima of the sliding peak
    double pw_sm[LCMAX + 1], pw_dc[LCMAX + 1], pw_sq[LCMAX + 1], pw_fd[LCMAX + 1];   // powers of the averager coefficients
    double bq[BQ_TAB];                   // biquad chunk tables
    double pm[(LCMAX + 1) * 4];          // PLL transition-matrix powers
};

// the image of the AGC peaks kernel (agc_peaks_kernel)
static_assert(sizeof(float) * PC_RLEVELS * PC_NCHUNK >= 8 * 512 && sizeof(float) * (PT + 16) >= 8 * 512,
              "pll_overlap keeps one double per thread in PcLds::rt and in PcLds::w2");
struct PreLds {
    PcSync sy;
    alignas(16) float mg[PH + PT];                   // [last win_n-1 log-magnitudes | tile]
    alignas(16) float pk[PT + 16];
    alignas(16) float w2[PT + 16];
    alignas(16) float rt[PC_RLEVELS][PC_NCHUNK];
};

// the squelch kernels' image (fm_squelch_*): a quarter of PcLds, so that eight of their workgroups share a CU
struct SqLds {
    PcSync sy;
    alignas(16) float w0[PT + PC_FIR_MAX + 17];      // [FIR history | tile] audio
    alignas(16) float h0[PC_FIR_MAX + 17];           // high-pass taps, reversed and zero padded
    alignas(16) float w2[PT + 16];                   // |hp| / audio being low-passed
    double pw_sq[LCMAX + 1];
    double bq[BQ_TAB];
};

__device__ __forceinline__ double wrap_turn(double a) { return a - rint(a); }    // [-0.5, 0.5]

// An EXACTLY ZERO sample in front of a PLL (round 6).  The kernels run the loops in the phase domain -- theta = arg(x), error
// = -wrap(theta + phi) -- and a zero has no argument.  The reference (fmdemod.cpp:166-172, samdemod.cpp:83-89 / 120-126) rotates
// first, t = x (cos phi + j s sin phi), and takes atan2(t.im, t.re): for x = (+-0, +-0) the products are signed zeros, their
// sum / difference is a signed zero by IEEE 754 (a - b = -0 only for (-0) - (+0); a + b = -0 only for (-0) + (-0)), and
// atan2(+-0, +0) = +-0 but atan2(+-0, -0) = +-pi: depending on the QUADRANT of phi the loop is left alone or kicked by half a
// turn.  Zeros are not exotic: the AGC's delay line is zeros for the first 15 ms of every stream, a muted input is zeros, and
// an fp32 filter's start-up is full of them (its samples are small multiples of one quantum) -- until round 5 the kernels
// took theta = 0 there, the error -wrap(phi), and the first burst of an FM receiver fed the SAME filter output differed from
// the reference's arithmetic by up to 0.6 of full scale (tests/test_randomized_gpu.py's loop-only check).
// The tile's theta array carries a zero sample as the code 4 + signbit(re) + 2 signbit(im) (no angle is beyond 0.5 turns).
__device__ __forceinline__ float pll_theta_of(float re, float im)
{
    if (re == 0.f && im == 0.f) return 4.0f + (float)(__float_as_uint(re) >> 31) + 2.0f * (float)(__float_as_uint(im) >> 31);
    return atan2f(im, re) * 0.15915494309189535f;                  // 1 / 2 pi
}
__device__ __forceinline__ bool pll_theta_is_zero_code(float v) { return v >= 3.5f; }
// the reference's phase error (turns) on such a sample; p = phi in turns, sgn = the sign s above (FM: +1; the error is -sgn atan2)
__device__ __forceinline__ double pll_zero_err(float code, double p, double sgn)
{
    const int bits = (int)code - 4;
    const bool sxr = bits & 1, sxi = bits & 2;
    const double fr = p - floor(p);                                    // [0, 1)
    const bool sc = fr > 0.25 && fr < 0.75;                            // cos(2 pi p) < 0
    const bool ss = (fr > 0.5) != (sgn < 0.0);                         // s sin(2 pi p) negative (-0 included: -1 x (+0))
    const bool neg_re = (sc != sxr) && !(ss != sxi);                   // c xr - s xi = -0
    const bool neg_im = (sc != sxi) && (ss != sxr);                    // c xi + s xr = -0
    const double a = neg_re ? (neg_im ? -0.5 : 0.5) : 0.0;             // atan2(t.im, t.re) in turns
    return -sgn * a;
}

// A wave's scan steps on the DPP network instead of __shfl_up (two ds_bpermute per double and step: an LDS-pipe
// round trip each, and a tile runs some ninety such steps one after the other on a single wave per SIMD).
// pc_dpp<CTRL, ROW_MASK>(v, ident): v of the source lane, `ident` where the step has none.  The six steps
// row_shr 1, 2, 4, 8, row_bcast 15 (rows 1, 3), row_bcast 31 (rows 2, 3) leave in every lane the combination of
// lanes 0 .. lane, like the six __shfl_up steps (associativity is all they need; identities make the lane guards
// unnecessary); wave_shr 1 then gives the exclusive value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double pc_dpp(double v, double ident)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v), o = (unsigned long long)__double_as_longlong(ident);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define PC_SCAN_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)

// workgroup context: thread id, lane, wave; barrier that orders LDS traffic of the whole workgroup
template <int NW>
struct Wg {
    static constexpr int NT = 64 * NW, LC = PT / NT;
    int t, lane, w;
    PcSync *S;
    mutable int bank = 0;                // exchange bank of the next workgroup scan (uniform)
    __device__ __forceinline__ double (*xbank() const)[8] { double (*b)[8] = S->xch[bank]; bank ^= 1; return b; }
    __device__ __forceinline__ void sync() const
    {
        if constexpr (NW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();
        }
    }
    // true on every thread if pred holds on any thread
    __device__ __forceinline__ bool any(bool pred) const
    {
        if constexpr (NW == 1) return __any(pred);
        if (t == 0) S->flag = 0;
        __syncthreads();
        if (__any(pred) && lane == 0) S->flag = 1;
        __syncthreads();
        const bool r = S->flag != 0;
        __syncthreads();
        return r;
    }
    // value held by thread 0 -> every thread
    __device__ __forceinline__ double bcast0(double v, int slot) const
    {
        if constexpr (NW == 1) return __shfl(v, 0);
        if (t == 0) S->bc[slot] = v;
        __syncthreads();
        const double r = S->bc[slot];
        __syncthreads();
        return r;
    }
    // Affine maps s -> A s + B, thread t's map applied after those of threads < t.
    // In: this thread's chunk map.  Out: (A, B) = composition of all EARLIER threads (exclusive),
    // (At, Bt) = composition of all threads.
    __device__ __forceinline__ void scan1(double &A, double &B, double &At, double &Bt) const
    {
#define PC_STEP(C_, R_) { const double A1 = pc_dpp<C_, R_>(A, 1.0), B1 = pc_dpp<C_, R_>(B, 0.0); B = A * B1 + B; A = A * A1; }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double Ae = pc_dpp<0x138, 0xf>(A, 1.0), Be = pc_dpp<0x138, 0xf>(B, 0.0);
        if constexpr (NW == 1) {
            At = __shfl(A, 63); Bt = __shfl(B, 63);
        } else {
            double (*xc)[8] = xbank();
            if (lane == 63) { xc[w][0] = A; xc[w][1] = B; }
            __syncthreads();
            double PA = 1.0, PB = 0.0;
            At = 1.0; Bt = 0.0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const double qa = xc[q][0], qb = xc[q][1];
                if (q < w) { PB = qa * PB + qb; PA = qa * PA; }
                Bt = qa * Bt + qb; At = qa * At;
            }
            Be = Ae * PB + Be; Ae = Ae * PA;
        }
        A = Ae; B = Be;
    }
    // scan1 and, in lockstep with it, the workgroup maximum of pk (returned in pk on every thread)
    __device__ __forceinline__ void scan1_max(double &A, double &B, double &At, double &Bt, double &pk) const
    {
#define PC_STEP(C_, R_) { const double A1 = pc_dpp<C_, R_>(A, 1.0), B1 = pc_dpp<C_, R_>(B, 0.0), P1 = pc_dpp<C_, R_>(pk, -1.0e300); \
                          B = A * B1 + B; A = A * A1; pk = fmax(pk, P1); }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double Ae = pc_dpp<0x138, 0xf>(A, 1.0), Be = pc_dpp<0x138, 0xf>(B, 0.0);
        if constexpr (NW == 1) {
            At = __shfl(A, 63); Bt = __shfl(B, 63); pk = __shfl(pk, 63);
        } else {
            double (*xc)[8] = xbank();
            if (lane == 63) { xc[w][0] = A; xc[w][1] = B; xc[w][2] = pk; }
            __syncthreads();
            double PA = 1.0, PB = 0.0;
            At = 1.0; Bt = 0.0; pk = -1.0e300;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const double qa = xc[q][0], qb = xc[q][1];
                if (q < w) { PB = qa * PB + qb; PA = qa * PA; }
                Bt = qa * Bt + qb; At = qa * At; pk = fmax(pk, xc[q][2]);
            }
            Be = Ae * PB + Be; Ae = Ae * PA;
        }
        A = Ae; B = Be;
    }
    // scan1 and scan2 in lockstep (the FM squelch average and the speculative audio low-pass of a one-tile burst)
    __device__ __forceinline__ void scan1_2(double &A, double &B, double &At, double &Bt, double (&m)[4], double (&v)[2], double (&mt)[4],
                                            double (&vt)[2]) const
    {
#define PC_STEP(C_, R_) { \
            const double A1 = pc_dpp<C_, R_>(A, 1.0), B1 = pc_dpp<C_, R_>(B, 0.0); \
            const double p00 = pc_dpp<C_, R_>(m[0], 1.0), p01 = pc_dpp<C_, R_>(m[1], 0.0), p10 = pc_dpp<C_, R_>(m[2], 0.0), p11 = pc_dpp<C_, R_>(m[3], 1.0); \
            const double q0 = pc_dpp<C_, R_>(v[0], 0.0), q1 = pc_dpp<C_, R_>(v[1], 0.0); \
            B = A * B1 + B; A = A * A1; \
            const double nv0 = m[0] * q0 + m[1] * q1 + v[0], nv1 = m[2] * q0 + m[3] * q1 + v[1]; \
            const double n00 = m[0] * p00 + m[1] * p10, n01 = m[0] * p01 + m[1] * p11; \
            const double n10 = m[2] * p00 + m[3] * p10, n11 = m[2] * p01 + m[3] * p11; \
            m[0] = n00; m[1] = n01; m[2] = n10; m[3] = n11; v[0] = nv0; v[1] = nv1; }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double Ae = pc_dpp<0x138, 0xf>(A, 1.0), Be = pc_dpp<0x138, 0xf>(B, 0.0);
        double e[4] = {pc_dpp<0x138, 0xf>(m[0], 1.0), pc_dpp<0x138, 0xf>(m[1], 0.0), pc_dpp<0x138, 0xf>(m[2], 0.0), pc_dpp<0x138, 0xf>(m[3], 1.0)};
        double ev[2] = {pc_dpp<0x138, 0xf>(v[0], 0.0), pc_dpp<0x138, 0xf>(v[1], 0.0)};
        static_assert(NW > 1, "scan1_2 is used by the four-wave kernel only");
        double (*xc)[8] = xbank();
        if (lane == 63) {
            xc[w][0] = A; xc[w][1] = B;
#pragma unroll
            for (int k = 0; k < 4; k++) xc[w][2 + k] = m[k];
            xc[w][6] = v[0]; xc[w][7] = v[1];
        }
        __syncthreads();
        double PA = 1.0, PB = 0.0, P[4] = {1.0, 0.0, 0.0, 1.0}, Pv[2] = {0.0, 0.0};
        At = 1.0; Bt = 0.0; mt[0] = 1.0; mt[1] = 0.0; mt[2] = 0.0; mt[3] = 1.0; vt[0] = 0.0; vt[1] = 0.0;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const double *x = xc[q];
            auto apply = [&](double (&M)[4], double (&V)[2]) {
                const double nv0 = x[2] * V[0] + x[3] * V[1] + x[6], nv1 = x[4] * V[0] + x[5] * V[1] + x[7];
                const double n00 = x[2] * M[0] + x[3] * M[2], n01 = x[2] * M[1] + x[3] * M[3];
                const double n10 = x[4] * M[0] + x[5] * M[2], n11 = x[4] * M[1] + x[5] * M[3];
                M[0] = n00; M[1] = n01; M[2] = n10; M[3] = n11; V[0] = nv0; V[1] = nv1;
            };
            if (q < w) { PB = x[0] * PB + x[1]; PA = x[0] * PA; apply(P, Pv); }
            Bt = x[0] * Bt + x[1]; At = x[0] * At; apply(mt, vt);
        }
        Be = Ae * PB + Be; Ae = Ae * PA;
        {
            const double nv0 = e[0] * Pv[0] + e[1] * Pv[1] + ev[0], nv1 = e[2] * Pv[0] + e[3] * Pv[1] + ev[1];
            const double n00 = e[0] * P[0] + e[1] * P[2], n01 = e[0] * P[1] + e[1] * P[3];
            const double n10 = e[2] * P[0] + e[3] * P[2], n11 = e[2] * P[1] + e[3] * P[3];
            e[0] = n00; e[1] = n01; e[2] = n10; e[3] = n11; ev[0] = nv0; ev[1] = nv1;
        }
        A = Ae; B = Be;
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = e[k];
        v[0] = ev[0]; v[1] = ev[1];
    }
    // two independent scans of that kind in lockstep: one wave per SIMD pays every instruction's latency, and
    // the two dependency chains fill each other's gaps; one exchange, one barrier
    __device__ __forceinline__ void scan1x2(double &A, double &B, double &At, double &Bt, double &C, double &D, double &Ct, double &Dt) const
    {
#define PC_STEP(C_, R_) { const double A1 = pc_dpp<C_, R_>(A, 1.0), B1 = pc_dpp<C_, R_>(B, 0.0), C1 = pc_dpp<C_, R_>(C, 1.0), D1 = pc_dpp<C_, R_>(D, 0.0); \
                          B = A * B1 + B; A = A * A1; D = C * D1 + D; C = C * C1; }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double Ae = pc_dpp<0x138, 0xf>(A, 1.0), Be = pc_dpp<0x138, 0xf>(B, 0.0), Ce = pc_dpp<0x138, 0xf>(C, 1.0), De = pc_dpp<0x138, 0xf>(D, 0.0);
        if constexpr (NW == 1) {
            At = __shfl(A, 63); Bt = __shfl(B, 63); Ct = __shfl(C, 63); Dt = __shfl(D, 63);
        } else {
            double (*xc)[8] = xbank();
            if (lane == 63) { xc[w][0] = A; xc[w][1] = B; xc[w][2] = C; xc[w][3] = D; }
            __syncthreads();
            double PA = 1.0, PB = 0.0, PC = 1.0, PD = 0.0;
            At = 1.0; Bt = 0.0; Ct = 1.0; Dt = 0.0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const double qa = xc[q][0], qb = xc[q][1], qc = xc[q][2], qd = xc[q][3];
                if (q < w) { PB = qa * PB + qb; PA = qa * PA; PD = qc * PD + qd; PC = qc * PC; }
                Bt = qa * Bt + qb; At = qa * At; Dt = qc * Dt + qd; Ct = qc * Ct;
            }
            Be = Ae * PB + Be; Ae = Ae * PA; De = Ce * PD + De; Ce = Ce * PC;
        }
        A = Ae; B = Be; C = Ce; D = De;
    }
    // maps x -> max(A x + B, C): same contract
    __device__ __forceinline__ void scan_max(double &A, double &B, double &Cc, double &At, double &Bt, double &Ct) const
    {
#define PC_STEP(C_, R_) { const double A1 = pc_dpp<C_, R_>(A, 1.0), B1 = pc_dpp<C_, R_>(B, 0.0), C1 = pc_dpp<C_, R_>(Cc, -1.0e300); \
                          Cc = fmax(A * C1 + B, Cc); B = A * B1 + B; A = A * A1; }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        if constexpr (NW == 1) {
            At = __shfl(A, 63); Bt = __shfl(B, 63); Ct = __shfl(Cc, 63);
        } else {
            double (*xc)[8] = xbank();
            if (lane == 63) { xc[w][0] = A; xc[w][1] = B; xc[w][2] = Cc; }
            __syncthreads();
            At = 1.0; Bt = 0.0; Ct = -1.0e300;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const double qa = xc[q][0], qb = xc[q][1], qc = xc[q][2];
                Ct = fmax(qa * Ct + qb, qc); Bt = qa * Bt + qb; At = qa * At;
            }
        }
    }
    // 2x2 affine maps s -> M s + v: same contract as scan1 (m, v in: chunk map; out: exclusive), totals in mt, vt
    __device__ __forceinline__ void scan2(double (&m)[4], double (&v)[2], double (&mt)[4], double (&vt)[2]) const
    {
#define PC_STEP(C_, R_) { \
            const double p00 = pc_dpp<C_, R_>(m[0], 1.0), p01 = pc_dpp<C_, R_>(m[1], 0.0), p10 = pc_dpp<C_, R_>(m[2], 0.0), p11 = pc_dpp<C_, R_>(m[3], 1.0); \
            const double q0 = pc_dpp<C_, R_>(v[0], 0.0), q1 = pc_dpp<C_, R_>(v[1], 0.0); \
            const double nv0 = m[0] * q0 + m[1] * q1 + v[0], nv1 = m[2] * q0 + m[3] * q1 + v[1]; \
            const double n00 = m[0] * p00 + m[1] * p10, n01 = m[0] * p01 + m[1] * p11; \
            const double n10 = m[2] * p00 + m[3] * p10, n11 = m[2] * p01 + m[3] * p11; \
            m[0] = n00; m[1] = n01; m[2] = n10; m[3] = n11; v[0] = nv0; v[1] = nv1; }
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double e[4] = {pc_dpp<0x138, 0xf>(m[0], 1.0), pc_dpp<0x138, 0xf>(m[1], 0.0), pc_dpp<0x138, 0xf>(m[2], 0.0), pc_dpp<0x138, 0xf>(m[3], 1.0)};
        double ev[2] = {pc_dpp<0x138, 0xf>(v[0], 0.0), pc_dpp<0x138, 0xf>(v[1], 0.0)};
        if constexpr (NW == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) mt[k] = __shfl(m[k], 63);
            vt[0] = __shfl(v[0], 63); vt[1] = __shfl(v[1], 63);
        } else {
            double (*xc)[8] = xbank();
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < 4; k++) xc[w][k] = m[k];
                xc[w][4] = v[0]; xc[w][5] = v[1];
            }
            __syncthreads();
            double P[4] = {1.0, 0.0, 0.0, 1.0}, Pv[2] = {0.0, 0.0};
            mt[0] = 1.0; mt[1] = 0.0; mt[2] = 0.0; mt[3] = 1.0; vt[0] = 0.0; vt[1] = 0.0;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const double *x = xc[q];
                auto apply = [&](double (&M)[4], double (&V)[2]) {
                    const double nv0 = x[0] * V[0] + x[1] * V[1] + x[4], nv1 = x[2] * V[0] + x[3] * V[1] + x[5];
                    const double n00 = x[0] * M[0] + x[1] * M[2], n01 = x[0] * M[1] + x[1] * M[3];
                    const double n10 = x[2] * M[0] + x[3] * M[2], n11 = x[2] * M[1] + x[3] * M[3];
                    M[0] = n00; M[1] = n01; M[2] = n10; M[3] = n11; V[0] = nv0; V[1] = nv1;
                };
                if (q < w) apply(P, Pv);
                apply(mt, vt);
            }
            // exclusive of this thread = (wave-exclusive e) after (previous waves P)
            const double nv0 = e[0] * Pv[0] + e[1] * Pv[1] + ev[0], nv1 = e[2] * Pv[0] + e[3] * Pv[1] + ev[1];
            const double n00 = e[0] * P[0] + e[1] * P[2], n01 = e[0] * P[1] + e[1] * P[3];
            const double n10 = e[2] * P[0] + e[3] * P[2], n11 = e[2] * P[1] + e[3] * P[3];
            e[0] = n00; e[1] = n01; e[2] = n10; e[3] = n11; ev[0] = nv0; ev[1] = nv1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) m[k] = e[k];
        v[0] = ev[0]; v[1] = ev[1];
    }
    // exclusive prefix sum over the threads
    __device__ __forceinline__ double scan_sum_excl(double x) const
    {
        double incl = x;
#define PC_STEP(C_, R_) incl += pc_dpp<C_, R_>(incl, 0.0);
        PC_SCAN_STEPS(PC_STEP)
#undef PC_STEP
        double ex = incl - x;
        if constexpr (NW > 1) {
            double (*xc)[8] = xbank();
            if (lane == 63) xc[w][0] = incl;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NW; q++) if (q < w) ex += xc[q][0];
        }
        return ex;
    }
    // value of thread t-1 (thread 0 gets `first`)
    __device__ __forceinline__ float prev_thread(float v, float first) const
    {
        float p = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x138, 0xf, 0xf, false));   // wave_shr:1
        if constexpr (NW > 1) {
            double (*xc)[8] = xbank();
            if (lane == 63) xc[w][6] = (double)v;
            __syncthreads();
            if (lane == 0 && w > 0) p = (float)xc[w - 1][6];
        }
        return t == 0 ? first : p;
    }
};

// The outputs a thread of the tile owns: LC consecutive ones with four waves (the FIRs below then share one register
// window of the input), the strided t + NT j otherwise.
template <int NW> __device__ __forceinline__ int pc_out_index(int t, int j)
{ return NW == 4 ? Wg<NW>::LC * t + j : t + Wg<NW>::NT * j; }

// acc[j] = sum_k tap[k] * x[i_j - k], k ascending, for this thread's outputs i_j = pc_out_index(t, j).  `arr` is the
// work array [ntaps-1 history | tile | zeros], `hr` the taps as PcLds holds them (reversed, zero padded).
// Four waves: the four outputs are consecutive, so the thread walks ONE window of the array in 16-byte reads
// (ntaps/4 + 1 of them, and as many of the taps) instead of fetching every tap's sample for every output -- the
// squelch filter alone was a fifth of an FM tile.  The products are added in the same order as the plain loop
// (padded taps contribute exact zeros), so the words do not change.
template <int NW>
__device__ __forceinline__ void fir_blk(const float *hr, int ntaps, const float *arr, int t, float (&acc)[Wg<NW>::LC])
{
    constexpr int NT = Wg<NW>::NT, NO = Wg<NW>::LC;
#pragma unroll
    for (int j = 0; j < NO; j++) acc[j] = 0.f;
    if constexpr (NW == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int Q = (ntaps + 2) / 4 + 1;                      // samples 4t .. 4t + ntaps + 2 of the array
        const f4 *v = reinterpret_cast<const f4 *>(arr) + t;
        const f4 *g = reinterpret_cast<const f4 *>(hr);
        f4 B = g[Q];
        for (int q = Q - 1; q >= 0; q--) {
            const f4 A = g[q], V = v[q];
            const float cw[7] = {A.y, A.z, A.w, B.x, B.y, B.z, B.w};     // taps r = 4q-3 .. 4q+3 (+4 in hr)
            const float sv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
            for (int e = 3; e >= 0; e--)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] += cw[e - j + 3] * sv[e];
            B = A;
        }
    } else {
        const float *p = arr + (ntaps - 1) + t;
#pragma unroll 3
        for (int k = 0; k < ntaps; k++) {
            const float hk = hr[4 + ntaps - 1 - k];
#pragma unroll
            for (int j = 0; j < NO; j++) acc[j] += hk * p[NT * j - k];
        }
    }
}
// keep the last `hist` (< 128) entries of [hist | n] in front for the next tile; barrier inside
template <int NW>
__device__ __forceinline__ void slide(const Wg<NW> &g, float *w, int hist, int n)
{
    float keep[2] = {0.f, 0.f};
    if (g.t < 64) for (int j = 0; j < 2; j++) { const int i = g.t + 64 * j; if (i < hist) keep[j] = w[n + i]; }
    g.sync();
    if (g.t < 64) for (int j = 0; j < 2; j++) { const int i = g.t + 64 * j; if (i < hist) w[i] = keep[j]; }
}

// Thread 0 walks src[0..n) in order, f(value, index); the next eight values are fetched from LDS
// while the current eight go through the recurrence, so no LDS latency sits on the dependent chain.
// src must be readable up to n+15.  (The fallback of the guessed solves below.)
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
// by sample: a thread runs its LC consecutive samples from a zero state, the chunk-start states
// follow from a workgroup scan over the affine chunk maps (a^cnt, p_last), and the homogeneous part
// a^(j+1) * S_start is added back.  Results differ from the sequential fp64 loop by rounding only.
// =====================================================================================================
__device__ __forceinline__ void pow_table(double *tab, double a, int t)    // tab[k] = a^k, k = 0..16
{
    if (t <= LCMAX) { double p = 1.0; for (int k = 0; k < t; k++) p *= a; tab[t] = p; }
}

// s_i = a s_{i-1} + g x_i  over x[0..n), n <= 1024, s_{-1} = s0.  emit(i, x_i, s_i, s_{i-1}) for every
// sample (skipped when EMIT is false).  Returns s_{n-1} on every thread.
template <bool EMIT, int NW, class F>
__device__ __forceinline__ double lin1_scan(const Wg<NW> &g, const float *x, int n, double a, double gn, double s0,
                                            const double *apw, F emit)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    float xv[LC];
    double loc[LC], p = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        xv[j] = j < cnt ? x[base + j] : 0.f;
        p = j < cnt ? a * p + gn * (double)xv[j] : p;
        loc[j] = p;
    }
    double A = apw[cnt], B = p, At, Bt;
    g.scan1(A, B, At, Bt);
    const double S = A * s0 + B;                                     // state entering this thread's chunk
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
    return At * s0 + Bt;
}

// CSMeter (smeter.cpp:62-93) over one tile, final state only.  att is a plain averager; dec obeys
// dec' = max(att', (1-da) dec + da mag), and maps x -> max(A x + B, C) are closed under composition.
template <int NW>
__device__ __forceinline__ void smeter_tile(const Wg<NW> &g, PcSMeter &sm, const float *db, int n, const double *apw_att)
{
    constexpr int LC = Wg<NW>::LC;
    const double aa = sm.att_a, ia = 1.0 - sm.att_a, da = sm.dec_a, id = 1.0 - sm.dec_a;
    const int base = LC * g.t;
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
    double A = apw_att[cnt], B = p, At, Bt;
    g.scan1_max(A, B, At, Bt, pk);                                   // pk: now the maximum of the whole tile
    const double S = A * sm.att_ave + B;
    const double att_end = At * sm.att_ave + Bt;
    double MA = 1.0, MB = 0.0, MC = -1.0e300, TA, TB, TC;            // chunk map of the decay average
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) {
            const double att = loc[j] + apw_att[j + 1] * S;          // updated attack average at this sample
            MA = id * MA; MB = id * MB + da * (double)xv[j]; MC = fmax(id * MC + da * (double)xv[j], att);
        }
    }
    g.scan_max(MA, MB, MC, TA, TB, TC);
    const double dec_end = fmax(TA * sm.dec_ave + TB, TC);
    sm.att_ave = att_end; sm.dec_ave = dec_end; sm.ave_mag = dec_end; sm.peak_mag = fmax(sm.peak_mag, pk);
}

// CIir direct form II (iir.cpp:171-186) over x[0..n) in place.  State s = (w1, w2):
// s' = M s + (x, 0), y = b0 x + c . s, M = [[-a1, -a2], [1, 0]], c = (b1 - b0 a1, b2 - b0 a2).
// tab: M^k (4 doubles each, k = 0..16) then r_k = c M^k (2 doubles each, k = 0..15)
__device__ __forceinline__ void biquad_table(double *tab, const PcIir &f, int t)
{
    if (t == 0) {
        double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
        const double c0 = f.b1 - f.b0 * f.a1, c1 = f.b2 - f.b0 * f.a2;
        for (int k = 0; k <= LCMAX; k++) {
            tab[4 * k] = m00; tab[4 * k + 1] = m01; tab[4 * k + 2] = m10; tab[4 * k + 3] = m11;
            if (k < LCMAX) { tab[68 + 2 * k] = c0 * m00 + c1 * m10; tab[68 + 2 * k + 1] = c0 * m01 + c1 * m11; }
            const double n00 = -f.a1 * m00 - f.a2 * m10, n01 = -f.a1 * m01 - f.a2 * m11;     // M * M^k
            m10 = m00; m11 = m01; m00 = n00; m01 = n01;
        }
    }
}
template <int NW>
__device__ __forceinline__ void biquad_scan(const Wg<NW> &g, float *x, int n, PcIir &f, const double *tab)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    double y[LC], w1 = 0.0, w2 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const double xv = j < cnt ? (double)x[base + j] : 0.0;
        const double w0 = xv - f.a1 * w1 - f.a2 * w2;
        y[j] = f.b0 * w0 + f.b1 * w1 + f.b2 * w2;
        if (j < cnt) { w2 = w1; w1 = w0; }
    }
    double m[4] = {tab[4 * cnt], tab[4 * cnt + 1], tab[4 * cnt + 2], tab[4 * cnt + 3]}, v[2] = {w1, w2}, mt[4], vt[2];
    g.scan2(m, v, mt, vt);
    const double S1 = m[0] * f.w1a + m[1] * f.w2a + v[0], S2 = m[2] * f.w1a + m[3] * f.w2a + v[1];
#pragma unroll
    for (int j = 0; j < LC; j++)
        if (j < cnt) x[base + j] = (float)(y[j] + tab[68 + 2 * j] * S1 + tab[68 + 2 * j + 1] * S2);
    const double nw1 = mt[0] * f.w1a + mt[1] * f.w2a + vt[0], nw2 = mt[2] * f.w1a + mt[3] * f.w2a + vt[1];
    f.w1a = nw1; f.w2a = nw2;
}

// One-tile FM burst: the squelch average (lin1_scan without emit over sq[0..n)) and, in lockstep with it, the audio
// low-pass of the burst over x[0..n) (biquad_scan) -- speculatively: whether the burst is squelched is only known
// from the average, so the filtered samples stay in registers (y, this thread's LC consecutive ones) and the
// filter state after the burst is returned in (w1n, w2n) for the caller to commit or drop.
template <int NW>
__device__ __forceinline__ double sq_and_lowpass(const Wg<NW> &g, const float *sq, const float *x, int n, double a, double gn, double s0,
                                                 const double *apw, const PcIir &f, const double *tab, float (&y)[Wg<NW>::LC],
                                                 double &w1n, double &w2n)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    double p = 0.0, yy[LC], w1 = 0.0, w2 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const float sv = j < cnt ? sq[base + j] : 0.f;
        p = j < cnt ? a * p + gn * (double)sv : p;
        const double xv = j < cnt ? (double)x[base + j] : 0.0;
        const double w0 = xv - f.a1 * w1 - f.a2 * w2;
        yy[j] = f.b0 * w0 + f.b1 * w1 + f.b2 * w2;
        if (j < cnt) { w2 = w1; w1 = w0; }
    }
    double A = apw[cnt], B = p, At, Bt;
    double m[4] = {tab[4 * cnt], tab[4 * cnt + 1], tab[4 * cnt + 2], tab[4 * cnt + 3]}, v[2] = {w1, w2}, mt[4], vt[2];
    g.scan1_2(A, B, At, Bt, m, v, mt, vt);
    const double S1 = m[0] * f.w1a + m[1] * f.w2a + v[0], S2 = m[2] * f.w1a + m[3] * f.w2a + v[1];
#pragma unroll
    for (int j = 0; j < LC; j++) y[j] = (float)(yy[j] + tab[68 + 2 * j] * S1 + tab[68 + 2 * j + 1] * S2);
    w1n = mt[0] * f.w1a + mt[1] * f.w2a + vt[0];
    w2n = mt[2] * f.w1a + mt[3] * f.w2a + vt[1];
    return At * s0 + Bt;
}

// CAgc's attack / decay averagers (agc.cpp:233-262):  ave += alpha (pk - ave),  alpha = rise when
// pk > ave else fall.  Piecewise linear, so: guess the selector bits, solve the then-linear
// recurrence with a scan, recompute the selectors from the solution, repeat until they reproduce
// themselves -- at that point the sequence IS the sequential one (each selector was taken against
// the true previous average).  The correct prefix grows every round; the peak moves slowly, so two
// or three rounds are typical.  Returns false after PC_AGC_ROUNDS without a fixed point (the caller
// then walks the tile sample by sample).
constexpr int PC_AGC_ROUNDS = 8;
template <int NW, class F>
__device__ __forceinline__ bool agc_ave_scan(const Wg<NW> &g, const float *pk, int n, double rise, double fall, double &ave, F emit)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
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
        double A = 1.0, B = 0.0, At, Bt;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                const double al = (sel >> j & 1) ? rise : fall;
                A = A - al * A;
                B = B + al * ((double)pv[j] - B);
            }
        }
        g.scan1(A, B, At, Bt);
        double x = A * ave0 + B;
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
        if (!g.any(nsel != sel)) {
#pragma unroll
            for (int j = 0; j < LC; j++) if (j < cnt) emit(base + j, val[j]);
            ave = At * ave0 + Bt;
            return true;
        }
        sel = nsel;
    }
    return false;
}

// Both averagers (attack, decay) of a tile in the same rounds: their scans are independent, and run in lockstep
// (Wg::scan1x2).  emit(i, attack_i, decay_i).
template <int NW, class F>
__device__ __forceinline__ bool agc_ave_scan2(const Wg<NW> &g, const float *pk, int n, double a_rise, double a_fall, double d_rise,
                                              double d_fall, double &att, double &dec, F emit)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    const double att0 = att, dec0 = dec;
    float pv[LC];
    unsigned sa = 0, sd = 0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        pv[j] = j < cnt ? pk[base + j] : 0.f;
        if (j < cnt && (double)pv[j] > att0) sa |= 1u << j;
        if (j < cnt && (double)pv[j] > dec0) sd |= 1u << j;
    }
    for (int round = 0; round < PC_AGC_ROUNDS; round++) {
        double A = 1.0, B = 0.0, At, Bt, C = 1.0, D = 0.0, Ct, Dt;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                const double al = (sa >> j & 1) ? a_rise : a_fall, dl = (sd >> j & 1) ? d_rise : d_fall;
                A = A - al * A; B = B + al * ((double)pv[j] - B);
                C = C - dl * C; D = D + dl * ((double)pv[j] - D);
            }
        }
        g.scan1x2(A, B, At, Bt, C, D, Ct, Dt);
        double x = A * att0 + B, y = C * dec0 + D;
        double va[LC], vd[LC];
        unsigned na = 0, nd = 0;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (j < cnt) {
                if ((double)pv[j] > x) na |= 1u << j;
                if ((double)pv[j] > y) nd |= 1u << j;
                const double al = (sa >> j & 1) ? a_rise : a_fall, dl = (sd >> j & 1) ? d_rise : d_fall;
                x = x + al * ((double)pv[j] - x);
                y = y + dl * ((double)pv[j] - y);
            }
            va[j] = x; vd[j] = y;
        }
        if (!g.any(na != sa || nd != sd)) {
#pragma unroll
            for (int j = 0; j < LC; j++) if (j < cnt) emit(base + j, va[j], vd[j]);
            att = At * att0 + Bt; dec = Ct * dec0 + Dt;
            return true;
        }
        sa = na; sd = nd;
    }
    return false;
}

// The second-order PLL of CFmDemod / CSamDemod (fmdemod.cpp:166-177, samdemod.cpp:83-97), in turns:
//   e = -wrap(theta_i + phi),  f += beta e (clamped to [lo, hi]),  phi += f + alpha e.
// With the input phase unwrapped (Theta_i, a prefix sum of wrapped differences) a locked loop keeps
// Theta_i + phi next to ONE integer K for a whole tile and never touches the clamp; under that
// guess e = (K - Theta_i) - phi and the loop is the constant-coefficient linear system
//   [phi; f] <- [[1-alpha-beta, 1], [-beta, 1]] [phi; f] + (alpha+beta, beta) (K - Theta_i),
// solved by a scan like the biquad.  The guess is then checked sample by sample (|e| < 1/2, f
// inside the clamp); if it holds everywhere the result is the sequential one, otherwise (cycle slip,
// acquisition, noise) the caller walks the tile.  emit(i, phi_before, f_after).
// tab: M^k, k = 0..16 (4 doubles each)
__device__ __forceinline__ void pll_table(double *tab, double alpha, double beta, int t)
{
    if (t == 0) {
        const double a00 = 1.0 - alpha - beta, a01 = 1.0, a10 = -beta, a11 = 1.0;
        double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
        for (int k = 0; k <= LCMAX; k++) {
            tab[4 * k] = m00; tab[4 * k + 1] = m01; tab[4 * k + 2] = m10; tab[4 * k + 3] = m11;
            const double n00 = a00 * m00 + a01 * m10, n01 = a00 * m01 + a01 * m11;
            const double n10 = a10 * m00 + a11 * m10, n11 = a10 * m01 + a11 * m11;
            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
        }
    }
}
template <int NW, class F>
__device__ __forceinline__ bool pll_scan(const Wg<NW> &g, const float *th, int n, double alpha, double beta, double lo, double hi,
                                         double &ph, double &fr, const double *tab, F emit)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    float tv[LC];
    bool zero = false;
#pragma unroll
    for (int j = 0; j < LC; j++) { tv[j] = j < cnt ? th[base + j] : 0.f; zero = zero || pll_theta_is_zero_code(tv[j]); }
    // (an exactly zero sample has no phase: the exact walks take the tile -- decided together with the guess's own check at the
    // end: a vote of its own per tile cost a single receiver's walk 0.4 us per tile, 5 % of a C2 / C5 call)
    // unwrapped input phase relative to the first sample of the tile
    float last = tv[0];
#pragma unroll
    for (int j = 1; j < LC; j++) if (j < cnt) last = tv[j];
    const float first = th[0];
    const float before = g.prev_thread(last, first);
    double c[LC], run = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const float pv = j == 0 ? before : tv[j - 1];
        const float d = tv[j] - pv;
        if (j < cnt) run += (double)(d - rintf(d));
        c[j] = run;
    }
    const double ex = g.scan_sum_excl(run);
    const double K = rint((double)first + ph);
    const double off = K - (double)first - ex;
#pragma unroll
    for (int j = 0; j < LC; j++) c[j] = off - c[j];                  // K - Theta_i
    // zero-state chunk response
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) {
            const double e = c[j] - v[0];
            v[1] = v[1] + beta * e;
            v[0] = v[0] + v[1] + alpha * e;
        }
    }
    double m[4] = {tab[4 * cnt], tab[4 * cnt + 1], tab[4 * cnt + 2], tab[4 * cnt + 3]}, mt[4], vt[2];
    g.scan2(m, v, mt, vt);
    double x0 = m[0] * ph + m[1] * fr + v[0];                        // state entering this thread's chunk
    double x1 = m[2] * ph + m[3] * fr + v[1];
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
    if (g.any(bad || zero)) return false;
    const double nph = mt[0] * ph + mt[1] * fr + vt[0], nfr = mt[2] * ph + mt[3] * fr + vt[1];
    ph = nph - rint(nph); fr = nfr;
    return true;
}

// The same loop where that guess does not hold for a whole tile -- no carrier (an idle channel: the phase is noise),
// cycle slips, the frequency clamp at work.  Every thread walks the EXACT recurrence over its LC samples, started
// `warm` samples early from a zero state.  The loop is a contraction: a state error shrinks by its spectral radius
// rho every sample (CFmDemod at 62.5 kHz: 0.47), a wrap moves the phase by whole turns only and a clamped frequency
// is exact at once -- so by its first own sample a thread's state has met the sequential one.  That is checked, not
// assumed: a thread's start state must equal its predecessor's end state to 1e-9 turns (the first threads start
// from the true state at sample 0), and then by induction every state is the sequential one TO THAT BOUND -- not bit
// for bit: the audio of an accepted tile may differ from the one-thread walk's by rounding (tests: 1e-6 of full scale;
// include/cutesdr_mi.h states the contract).  Any mismatch -> false, and the caller walks the tile.  warm = pll_warm_len() samples: rho^(warm-8) < 1e-13; the eight
// on top are for a wrap decision that differs late in the warm-up (measured on noise: 32 samples at rho = 0.47 sent
// a quarter of the tiles to the one-thread walk, 48 none: 1.92 -> 0.45 ms for 85 idle receivers x 2^20 samples, against
// 0.34 ms with carriers).
// xp, xf: NT doubles of LDS each.  emit(i, phase_before, freq_after).
__device__ __forceinline__ int pll_warm_len(double alpha, double beta)
{
    const double tr = 2.0 - alpha - beta, det = 1.0 - alpha, disc = tr * tr - 4.0 * det;
    const double rho = disc < 0.0 ? sqrt(fabs(det)) : 0.5 * (fabs(tr) + sqrt(disc));
    if (!(rho > 0.0)) return 8;
    if (!(rho < 0.98)) return 1 << 20;
    return (int)(-30.0 / log(rho)) + 10;
}
template <int NW, class F>
__device__ __forceinline__ bool pll_overlap(const Wg<NW> &g, const float *th, int n, int warm, double alpha, double beta, double lo,
                                            double hi, double &ph, double &fr, double *xp, double *xf, F emit)
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    int i = base - warm;
    double p = 0.0, f = 0.0;
    const bool exact = i <= 0;
    if (exact) { i = 0; p = ph; f = fr; }
    auto step = [&](float v) {                                  // fmdemod.cpp:166-177, in turns
        const double err = pll_theta_is_zero_code(v) ? pll_zero_err(v, p, 1.0) : -wrap_turn((double)v + p);
        f = fmin(fmax(f + beta * err, lo), hi);
        p = wrap_turn(p + f + alpha * err);
    };
    if (cnt > 0) {
        float v = th[i];
        for (; i < base; i++) { const float vn = th[i + 1]; step(v); v = vn; }
    }
    const double ps = p, fs = f;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        if (j < cnt) { const double pb = p; step(th[base + j]); emit(base + j, pb, f); }
    }
    xp[g.t] = p; xf[g.t] = f;
    g.sync();
    bool bad = false;
    if (cnt > 0 && !exact) {
        const double dp = ps - xp[g.t - 1], df = fs - xf[g.t - 1];
        bad = !(fabs(dp - rint(dp)) < 1e-9) || !(fabs(df) < 1e-9);
    }
    const int tl = n > 0 ? (n - 1) / LC : 0;
    const double pe = xp[tl], fe = xf[tl];
    if (g.any(bad)) return false;
    if (n > 0) { ph = pe; fr = fe; }
    return true;
}

// pk[i] = max(E[i .. i+W1]), i < n, for E = S.mg[0 .. W1+n).  Chunks of 16: per-chunk prefix and
// suffix maxima (registers), a log table over the <= 192 chunk maxima for the chunks strictly
// inside a window, then window = suffix(first chunk) | inner chunks | prefix(last chunk).
// Only prefix values at tile positions (-> S.pk) and suffix values at i < n (-> S.w2) are kept.
template <int NW, class L>
__device__ __forceinline__ void sliding_max(const Wg<NW> &g, L &S, int W1, int n)
{
    constexpr int NT = Wg<NW>::NT;
    const int len = W1 + n, t = g.t;
    if (W1 < 16) {                                      // short windows: direct
        for (int i = t; i < n; i += NT) {
            float v = S.mg[i];
            for (int k = 1; k <= W1; k++) v = fmaxf(v, S.mg[i + k]);
            S.pk[i] = v;
        }
        g.sync();
        return;
    }
    const int nc = (len + 15) >> 4;
    for (int c = t; c < PC_NCHUNK; c += NT) {
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
    g.sync();
    for (int l = 1; l < PC_RLEVELS; l++) {
        const int h = 1 << (l - 1);
        for (int c = t; c < PC_NCHUNK; c += NT) {
            const float v0 = S.rt[l - 1][c];
            const float v1 = (c + h < PC_NCHUNK) ? S.rt[l - 1][c + h] : kNegBig;
            S.rt[l][c] = fmaxf(v0, v1);
        }
        g.sync();
    }
    for (int i = t; i < n; i += NT) {
        const int ci = i >> 4, ce = (i + W1) >> 4, cnt = ce - ci - 1;     // W1 >= 16: ce > ci
        float v = fmaxf(S.w2[i], S.pk[i]);
        if (cnt > 0) {
            const int l = 31 - __clz(cnt);
            v = fmaxf(v, fmaxf(S.rt[l][ci + 1], S.rt[l][ce - (1 << l)]));
        }
        S.pk[i] = v;
    }
    g.sync();
}

#ifndef CSDR_PC_WAVES_PER_EU
#define CSDR_PC_WAVES_PER_EU 1
#endif
// LEAN: the walk of a chain call of several bursts, where everything without feedback has left it -- no S-meter
// (smeter_call_kernel), the AGC's peaks from agc_peaks_kernel (PC_AGC_PRE), FM's squelch half deferred (PC_FM_DEFER).
// The code those stages needed is compiled out: it never ran in such a launch, but its live ranges set the kernel's
// register allocation (256 + 118 for the full kernel: one workgroup per CU and nothing but one down-converter wave
// beside it on a SIMD).
#ifndef CSDR_PC_LEAN_WAVES_PER_EU
#define CSDR_PC_LEAN_WAVES_PER_EU 2
#endif
template <int NW, bool LEAN>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 4 ? (LEAN ? CSDR_PC_LEAN_WAVES_PER_EU : CSDR_PC_WAVES_PER_EU) : 1)))
void postchain_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_WALK);
    using G = Wg<NW>;
    constexpr int NT = G::NT, LC = G::LC;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    PcLds &S = *reinterpret_cast<PcLds *>(pc_smem);
    const int ch = blockIdx.x, t = threadIdx.x;
    // A channel's bursts are a sequential walk: this kernel is bound by its own latency, not by throughput, and
    // in the batch chain it shares CUs with the down-converter of other groups / the next call, whose 12-16 waves
    // per CU would otherwise take most issue slots.  Highest issue priority for these few waves costs the
    // streaming kernel next to nothing and keeps the walk at the speed it has alone.
    if (a.out_rows && a.out_rows[ch] < 0) return;      // muted row (its receiver has moved on): uniform per workgroup
    __builtin_amdgcn_s_setprio(3);
    const G g{t, t & 63, t >> 6, &S.sy};
    PcChannel &C = a.chan[ch];
    float *g_dly = a.agc_dly + (long)ch * PC_AGC_RING * 2;      // linear: last dly_n inputs, oldest first
    float *g_mag = a.agc_mag + (long)ch * PC_AGC_RING;          // linear: last win_n-1 magnitudes
    const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)ch * a.in_stride;
    const bool stereo = a.flags & PC_STEREO;
    const long orow = (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    float *outm = a.out ? a.out + orow : nullptr;                               // mono
    float2 *outs = a.out ? reinterpret_cast<float2 *>(a.out) + orow : nullptr;  // stereo / complex
    const int mode = (a.flags & PC_DO_DEMOD) ? C.mode : PC_MODE_NONE;
    const bool do_sm = !LEAN && (a.flags & PC_DO_SMETER), do_agc = a.flags & PC_DO_AGC, agc_real = !LEAN && (a.flags & PC_AGC_REAL);
    const bool cpx_out = stereo || mode == PC_MODE_NONE;
    // FM with the squelch deferred: this walk ends at the raw audio (fm_squelch_launch does the rest, burst-parallel)
    const bool defer = LEAN ? mode == PC_MODE_FM : ((a.flags & PC_FM_DEFER) && mode == PC_MODE_FM && a.burst <= 16384);
    // AGC peaks from agc_peaks_kernel: the walk neither computes nor keeps the window's log magnitudes
    const bool pre = LEAN ? (do_agc && C.agc.on) : ((a.flags & PC_AGC_PRE) && do_agc && C.agc.on && !agc_real);
    const float *pkrow = pre ? a.pkbuf + (long)ch * a.nbursts * a.burst : nullptr;

    // scalar state, identical on every thread
    PcSMeter sm = C.sm;
    PcAgc agc = C.agc;
    const int D = agc.dly_n > 0 ? agc.dly_n : 1, W1 = agc.win_n > 0 ? agc.win_n - 1 : 0;
    double am_z1 = C.am.z1;
    double sam_z1 = C.sam.z1, sam_y1 = C.sam.y1, sam_ph = C.sam.phase, sam_fr = C.sam.freq;
    double fm_ph = C.fm.phase, fm_fr = C.fm.freq, fm_dc = C.fm.err_dc, fm_sq = C.fm.sq_ave;
    int fm_squelched = C.fm.squelched;
    PcIir lp = C.fm.lp;
    const int fm_warm = mode == PC_MODE_FM ? pll_warm_len(C.fm.alpha, C.fm.beta) : 0;

    // histories -> LDS
    if (do_agc && agc.on) {
        for (int i = t; i < D; i += NT) S.dl[i] = make_float2(g_dly[2 * i], g_dly[2 * i + 1]);
        if constexpr (!LEAN) if (!pre) for (int i = t; i < W1; i += NT) S.mg[i] = g_mag[i];
    }
    const PcFir *fir = mode == PC_MODE_AM ? &C.am.fir : mode == PC_MODE_SAM ? &C.sam.fir : mode == PC_MODE_FM ? &C.fm.hp : nullptr;
    const int nt = fir ? fir->ntaps : 1;
    if (fir) {
        for (int i = t; i < PC_FIR_MAX + 17; i += NT) {
            const int k = nt - 1 - (i - 4);                      // reversed, four zeros in front, zeros behind
            const bool in = k >= 0 && k < nt;
            S.h0[i] = in ? ((mode == PC_MODE_FM || !stereo) ? fir->coef[k] : fir->icoef[k]) : 0.f;
            S.h1[i] = in ? fir->qcoef[k] : 0.f;
        }
        // behind the tile the FIR window reads up to three more samples: zeros, never written again
        for (int i = t; i < PT + PC_FIR_MAX + 17; i += NT) { S.w0[i] = 0.f; S.w1[i] = 0.f; }
        g.sync();
        for (int i = t; i < nt - 1; i += NT) { S.w0[i] = (mode == PC_MODE_FM || (mode == PC_MODE_AM && !stereo)) ? fir->zreal[i] : fir->zr[i]; S.w1[i] = fir->zi[i]; }
    }
    pow_table(S.pw_sm, 1.0 - sm.att_a, t);
    static_assert(refc::AM_DC_ALPHA == refc::SAM_DC_ALPHA, "one table of DC-blocker powers serves AM and SAM");
    pow_table(S.pw_dc, refc::AM_DC_ALPHA, t);
    pow_table(S.pw_sq, 1.0 - C.fm.sq_alpha, t);
    pow_table(S.pw_fd, 1.0 - C.fm.dc_alpha, t);
    biquad_table(S.bq, lp, t);
    if (mode == PC_MODE_FM) pll_table(S.pm, C.fm.alpha, C.fm.beta, t);
    if (mode == PC_MODE_SAM) pll_table(S.pm, C.sam.alpha, C.sam.beta, t);
    g.sync();

    // with several waves per channel a tile's samples are fetched into registers one tile ahead (a
    // channel's workgroup usually has its CU to itself: nothing else would hide the HBM latency)
    constexpr bool kPrefetch = NW > 1;
    float2 nxt[LC];
    float nxp[LC];                                        // PC_AGC_PRE: the tile's peaks travel with its samples
    const long total = (long)a.nbursts * a.burst;
    auto fetch = [&](long g0, int cnt) {
#pragma unroll
        for (int j = 0; j < LC; j++) { const int i = t + NT * j; if (i < cnt) { nxt[j] = in[g0 + i]; if (pre) nxp[j] = pkrow[g0 + i]; } }
    };
    if (kPrefetch && total > 0) fetch(0, a.burst < PT ? a.burst : PT);
#ifdef PC_PROFILE
    unsigned long long tk[16] = {0}, tlast = __builtin_readcyclecounter();
#define PC_TICK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tk[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define PC_TICK(k)
#endif
    float lp_y[LC];                                       // one-tile FM burst: its low-passed audio, kept until the squelch decision
    double lp_w1 = 0.0, lp_w2 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) lp_y[j] = 0.f;
    for (int b = 0; b < a.nbursts; b++) {
        for (int t0 = 0; t0 < a.burst; t0 += PT) {
            const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
            const long gi = (long)b * a.burst + t0;
            float2 *x = S.dl + ((do_agc && agc.on) ? D : 0);           // tile samples (AGC: behind the delay history)
            if (kPrefetch) {
#pragma unroll
                for (int j = 0; j < LC; j++) { const int i = t + NT * j; if (i < n) { x[i] = nxt[j]; if (pre) S.pk[i] = nxp[j]; } }
                const long gn = gi + n;                                // bursts are contiguous: the next tile follows
                if (gn < total) {
                    const int left = a.burst - ((t0 + n) % a.burst);
                    fetch(gn, left < PT ? left : PT);
                }
            } else {
                for (int i = t; i < n; i += NT) x[i] = in[gi + i];
            }
            g.sync();
            PC_TICK(0);
            // ---------------- S-meter (smeter.cpp:62-93) ----------------
            if constexpr (!LEAN) if (do_sm) {
                for (int i = t; i < n; i += NT) {
                    const float pw = (x[i].x * x[i].x + x[i].y * x[i].y) * refc::SM_INV_MAX_PWR_F;
                    S.w2[i] = pw > 0.f ? 10.0f * log10f(pw) : -500.0f;
                }
                g.sync();
                smeter_tile(g, sm, S.w2, n, S.pw_sm);
                g.sync();
            }
            PC_TICK(1);
            // ---------------- AGC (agc.cpp:174-296 / 301-401) ----------------
            if (do_agc) {
                if (!agc.on) {
                    const float gm = (float)agc.manual_gain;
                    for (int i = t; i < n; i += NT) { x[i].x *= gm; x[i].y *= gm; }
                    g.sync();
                } else {
                    if (pre) {
                        if (!kPrefetch) { for (int i = t; i < n; i += NT) S.pk[i] = pkrow[gi + i]; g.sync(); }
                    } else if constexpr (!LEAN) {
                        float *mg = S.mg + W1;
                        for (int i = t; i < n; i += NT) {
                            float m = fabsf(x[i].x);
                            if (!agc_real) { const float mi = fabsf(x[i].y); if (mi > m) m = mi; }
                            mg[i] = log10f(m + refc::AGC_MIN_CONSTANT_F) - refc::AGC_LOG10_MAX_AMPLITUDE_F;
                        }
                        g.sync();
                        // sliding maximum: pk[i] = max E[i .. i+W1], E = [W1 history | tile] = S.mg
                        PC_TICK(2);
                        sliding_max(g, S, W1, n);
                        PC_TICK(3);
                        // the last W1 magnitudes are the next tile's history: read all, one barrier, write all
                        {
                            float keepm[PH / NT];
#pragma unroll
                            for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < W1) keepm[j] = S.mg[n + i]; }
                            g.sync();
#pragma unroll
                            for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < W1) S.mg[i] = keepm[j]; }
                            g.sync();
                        }
                    }
                    PC_TICK(4);
                    // attack / decay averagers -> log gain argument max(att, dec) per sample in S.pk
                    {
                        double att = agc.attack_ave, dec = agc.decay_ave;
                        bool ok = !agc.hang;                              // the hang timer is a counter: walked
                        if constexpr (NW == 4) {           // (16 samples per thread would not fit the registers twice)
                            if (ok) ok = agc_ave_scan2(g, S.pk, n, agc.att_rise, agc.att_fall, agc.dec_rise, agc.dec_fall, att, dec,
                                                       [&](int i, double va, double vd) { S.pk[i] = fmaxf((float)va, (float)vd); });
                        } else {
                            if (ok) ok = agc_ave_scan(g, S.pk, n, agc.att_rise, agc.att_fall, att,
                                                      [&](int i, double v) { S.w2[i] = (float)v; });
                            if (ok) ok = agc_ave_scan(g, S.pk, n, agc.dec_rise, agc.dec_fall, dec,
                                                      [&](int i, double v) { S.pk[i] = fmaxf(S.w2[i], (float)v); });
                        }
                        if (ok) { agc.attack_ave = att; agc.decay_ave = dec; }
                        else {
                            g.sync();
                            att = agc.attack_ave; dec = agc.decay_ave;
                            int timer = agc.hang_timer;
                            if (t == 0) {
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
                            agc.attack_ave = g.bcast0(att, 0); agc.decay_ave = g.bcast0(dec, 1);
                            agc.hang_timer = (int)g.bcast0((double)timer, 2);
                        }
                    }
                    g.sync();
                    PC_TICK(5);
                    // gain law + delay line: out[i] = in[i - D] * gain[i]; S.dl = [D old | n new]
                    const float knee = (float)agc.knee, slm1 = (float)(agc.gain_slope - 1.0), fixed_gain = (float)agc.fixed_gain;
                    float2 outv[LC];
#pragma unroll
                    for (int j = 0; j < LC; j++) {
                        const int i = t + NT * j;
                        if (i < n) {
                            const float m = S.pk[i];
                            const float gv = (m <= knee) ? fixed_gain : 0.7f * exp10f(m * slm1);
                            const float2 d = S.dl[i];
                            outv[j] = make_float2(d.x * gv, d.y * gv);
                        }
                    }
                    float2 keepd[PH / NT];
#pragma unroll
                    for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < D) keepd[j] = S.dl[n + i]; }
                    g.sync();
#pragma unroll
                    for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < D) S.dl[i] = keepd[j]; }
#pragma unroll
                    for (int j = 0; j < LC; j++) { const int i = t + NT * j; if (i < n) x[i] = outv[j]; }
                    g.sync();
                }
            }
            PC_TICK(6);
            // x[0..n) now holds the AGC output (or the input); x = S.dl + D
            // ---------------- demodulators ----------------
            if (mode == PC_MODE_NONE || mode >= PC_MODE_USB) {
                if (a.out) {
                    for (int i = t; i < n; i += NT) {
                        if (cpx_out) outs[gi + i] = x[i];
                        else outm[gi + i] = x[i].x;                       // ssbdemod.cpp:48-53
                    }
                }
                g.sync();
            } else if (mode == PC_MODE_AM) {
                float *w = S.w0 + (nt - 1);
                for (int i = t; i < n; i += NT) w[i] = sqrtf(x[i].x * x[i].x + x[i].y * x[i].y);
                g.sync();
                // DC block z0 = x + 0.99 z1, out = z0 - z1 (amdemod.cpp:70-80)
                am_z1 = lin1_scan<true>(g, w, n, refc::AM_DC_ALPHA, 1.0, am_z1, S.pw_dc,
                                        [&](int i, float, double z0, double z1) { w[i] = (float)(z0 - z1); });
                g.sync();
                float acc[LC], acq[LC];
                fir_blk<NW>(S.h0, nt, S.w0, t, acc);
                if (stereo) {
                    fir_blk<NW>(S.h1, nt, S.w0, t, acq);
#pragma unroll
                    for (int j = 0; j < LC; j++) { const int i = pc_out_index<NW>(t, j); if (i < n) outs[gi + i] = make_float2(acc[j], acq[j]); }
                } else {
#pragma unroll
                    for (int j = 0; j < LC; j++) { const int i = pc_out_index<NW>(t, j); if (i < n) outm[gi + i] = acc[j]; }
                }
                g.sync();
                slide(g, S.w0, nt - 1, n);
                g.sync();
            } else {
                // PLL modes: theta = arg(x) for the whole tile, in turns
                float *th = S.w1 + (nt - 1), *au = S.w0 + (nt - 1);
                for (int i = t; i < n; i += NT) th[i] = pll_theta_of(x[i].x, x[i].y);
                g.sync();
                PC_TICK(7);
                if (mode == PC_MODE_FM) {
                    const PcFm &F = C.fm;
                    bool scanned;
                    {
                        double ph = fm_ph * kInvTwoPiD, fr = fm_fr * kInvTwoPiD;
                        scanned = pll_scan(g, th, n, F.alpha, F.beta, F.lo * kInvTwoPiD, F.hi * kInvTwoPiD, ph, fr, S.pm,
                                           [&](int i, double, double f) { au[i] = (float)f; });
                        if (scanned) { fm_ph = ph * kTwoPiD; fm_fr = fr * kTwoPiD; }
                    }
                    if (!scanned && fm_warm <= PC_PLL_WARM_MAX && !(a.flags & PC_PLL_SEQ)) {       // unlocked: overlapped exact walks, verified
                        g.sync();
                        double ph = fm_ph * kInvTwoPiD, fr = fm_fr * kInvTwoPiD;
                        ph = wrap_turn(ph);
                        scanned = pll_overlap(g, th, n, fm_warm, F.alpha, F.beta, F.lo * kInvTwoPiD, F.hi * kInvTwoPiD, ph, fr,
                                              reinterpret_cast<double *>(&S.rt[0][0]), reinterpret_cast<double *>(S.w2),
                                              [&](int i, double, double f) { au[i] = (float)f; });
                        if (scanned) { fm_ph = ph * kTwoPiD; fm_fr = fr * kTwoPiD; }
                    }
                    if (!scanned) {
                        g.sync();
                        if (t == 0) {
                            // phase, frequency and error in turns: wrapping is a - rint(a)
                            const double beta = F.beta, alpha = F.alpha, hi = F.hi * kInvTwoPiD, lo = F.lo * kInvTwoPiD;
                            double ph = fm_ph * kInvTwoPiD, fr = fm_fr * kInvTwoPiD;
                            seq_walk(th, n, [&](float v, int i) {              // fmdemod.cpp:166-177
                                const double err = pll_theta_is_zero_code(v) ? pll_zero_err(v, ph, 1.0) : -wrap_turn((double)v + ph);
                                fr = fmin(fmax(fr + beta * err, lo), hi);
                                ph = wrap_turn(ph + fr + alpha * err);
                                au[i] = (float)fr;                             // NCO frequency, turns per sample
                            });
                            fm_ph = ph * kTwoPiD; fm_fr = fr * kTwoPiD;
                        }
                        fm_ph = g.bcast0(fm_ph, 0); fm_fr = g.bcast0(fm_fr, 1);
                    }
                    g.sync();
                    PC_TICK(8);
                    {   // audio = (freq - its running mean) * gain  (fmdemod.cpp:178-186): the mean is linear
                        const double og = F.out_gain * kTwoPiD;
                        fm_dc = kTwoPiD * lin1_scan<true>(g, au, n, 1.0 - F.dc_alpha, F.dc_alpha, fm_dc * kInvTwoPiD, S.pw_fd,
                                    [&](int i, float f, double dc, double) { au[i] = (float)(((double)f - dc) * og); });
                    }
                    g.sync();
                    PC_TICK(9);
                    // raw audio to the output row; squelch is decided at the end of the burst (a burst of one tile
                    // keeps it in LDS instead: the low-pass at the end of the burst reads it from there)
                    if (a.burst > PT || defer)
                        for (int i = t; i < n; i += NT) { if (stereo) outs[gi + i] = make_float2(au[i], au[i]); else outm[gi + i] = au[i]; }
                    if constexpr (!LEAN) if (a.burst <= 16384 && !defer) {                     // MAX_SQBUF_SIZE
                        float acc[LC];
                        fir_blk<NW>(S.h0, nt, S.w0, t, acc);
#pragma unroll
                        for (int j = 0; j < LC; j++) S.w2[pc_out_index<NW>(t, j) & (PT - 1)] = fabsf(acc[j]);
                        g.sync();
                        if constexpr (NW == 4) {
                            if (a.burst <= PT) fm_sq = sq_and_lowpass(g, S.w2, au, n, 1.0 - F.sq_alpha, F.sq_alpha, fm_sq, S.pw_sq, lp, S.bq,
                                                                      lp_y, lp_w1, lp_w2);
                            else fm_sq = lin1_scan<false>(g, S.w2, n, 1.0 - F.sq_alpha, F.sq_alpha, fm_sq, S.pw_sq,
                                                          [](int, float, double, double) {});
                        } else
                        fm_sq = lin1_scan<false>(g, S.w2, n, 1.0 - F.sq_alpha, F.sq_alpha, fm_sq, S.pw_sq,
                                                 [](int, float, double, double) {});
                    }
                    g.sync();
                    PC_TICK(10);
                    if (!defer) {
                        slide(g, S.w0, nt - 1, n);
                        g.sync();
                    }
                    PC_TICK(11);
                } else {                                                  // SAM, samdemod.cpp:78-158
                    const PcSam &M = C.sam;
                    const double sgn = stereo ? 1.0 : -1.0;
                    bool scanned;
                    {   // in psi = sgn phi, g = sgn f the loop has the FM form with theta as is
                        double ps = sgn * sam_ph * kInvTwoPiD, gf = sgn * sam_fr * kInvTwoPiD;
                        const double l0 = M.lo * kInvTwoPiD, h0 = M.hi * kInvTwoPiD;
                        scanned = pll_scan(g, th, n, M.alpha, M.beta, sgn > 0 ? l0 : -h0, sgn > 0 ? h0 : -l0, ps, gf, S.pm,
                                           [&](int i, double p, double) { S.w2[i] = (float)(sgn * (p - rint(p))); });
                        if (scanned) { sam_ph = sgn * ps * kTwoPiD; sam_fr = sgn * gf * kTwoPiD; }
                    }
                    if (!scanned) {
                        g.sync();
                        if (t == 0) {
                            const double beta = M.beta, alpha = M.alpha, hi = M.hi * kInvTwoPiD, lo = M.lo * kInvTwoPiD;
                            double ph = sam_ph * kInvTwoPiD, fr = sam_fr * kInvTwoPiD;
                            seq_walk(th, n, [&](float v, int i) {
                                S.w2[i] = (float)ph;                          // phase used for this sample (turns)
                                const double err = pll_theta_is_zero_code(v) ? pll_zero_err(v, ph, sgn) : -sgn * wrap_turn((double)v + sgn * ph);
                                fr = fmin(fmax(fr + beta * err, lo), hi);
                                ph = wrap_turn(ph + fr + alpha * err);
                            });
                            sam_ph = ph * kTwoPiD; sam_fr = fr * kTwoPiD;
                        }
                        sam_ph = g.bcast0(sam_ph, 0); sam_fr = g.bcast0(sam_fr, 1);
                    }
                    g.sync();
                    // rotated sample tr + j ti = |x| e^{j(theta + sgn phi)}
                    for (int i = t; i < n; i += NT) {
                        const float r = sqrtf(x[i].x * x[i].x + x[i].y * x[i].y);
                        float sn, cs;
                        sincospif(2.0f * (th[i] + (float)sgn * S.w2[i]), &sn, &cs);
                        au[i] = r * cs;                                   // tr
                        th[i] = r * sn;                                   // ti
                    }
                    g.sync();
                    // DC blocks
                    sam_z1 = lin1_scan<true>(g, au, n, refc::SAM_DC_ALPHA, 1.0, sam_z1, S.pw_dc,
                                             [&](int i, float, double z0, double z1) { au[i] = (float)(z0 - z1); });
                    if (stereo)
                        sam_y1 = lin1_scan<true>(g, th, n, refc::SAM_DC_ALPHA, 1.0, sam_y1, S.pw_dc,
                                                 [&](int i, float, double y0, double y1) { th[i] = (float)(y0 - y1); });
                    g.sync();
                    if (!stereo) {
                        for (int i = t; i < n; i += NT) outm[gi + i] = au[i];
                    } else {
                        float ar[LC], ai[LC];
                        fir_blk<NW>(S.h0, nt, S.w0, t, ar);
                        fir_blk<NW>(S.h1, nt, S.w1, t, ai);
#pragma unroll
                        for (int j = 0; j < LC; j++) {                    // lower sideband left, upper right
                            const int i = pc_out_index<NW>(t, j);
                            if (i < n) outs[gi + i] = make_float2(ar[j] + ai[j], ar[j] - ai[j]);
                        }
                        g.sync();
                        slide(g, S.w0, nt - 1, n);
                        slide(g, S.w1, nt - 1, n);
                    }
                    g.sync();
                }
            }
        }
        // ---------------- end of burst: FM squelch decision (fmdemod.cpp:128-151) ----------------
        if constexpr (!LEAN) if (mode == PC_MODE_FM && a.burst <= 16384 && !defer) {
            const PcFm &F = C.fm;
            if (0 == F.sq_thresh) fm_squelched = 1;
            else if (fm_squelched) { if (fm_sq < (F.sq_thresh - refc::FM_SQUELCH_HYSTERESIS)) fm_squelched = 0; }
            else { if (fm_sq >= (F.sq_thresh + refc::FM_SQUELCH_HYSTERESIS)) fm_squelched = 1; }
            const long g0 = (long)b * a.burst;
            g.sync();
            for (int t0 = 0; t0 < a.burst; t0 += PT) {
                const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
                if (fm_squelched) {
                    for (int i = t; i < n; i += NT) { if (stereo) outs[g0 + t0 + i] = make_float2(0.f, 0.f); else outm[g0 + t0 + i] = 0.f; }
                } else if (NW == 4 && a.burst <= PT) {                    // the low-pass ran with the squelch average: commit it
#pragma unroll
                    for (int j = 0; j < LC; j++) {
                        const int i = LC * t + j;
                        if (i < n) { if (stereo) outs[g0 + i] = make_float2(lp_y[j], lp_y[j]); else outm[g0 + i] = lp_y[j]; }
                    }
                    lp.w1a = lp_w1; lp.w2a = lp_w2;
                } else {                                                  // low-pass biquad over the burst
                    if (a.burst > PT) { for (int i = t; i < n; i += NT) S.w2[i] = stereo ? outs[g0 + t0 + i].x : outm[g0 + t0 + i]; }
                    else { const float *au = S.w0 + (nt - 1); for (int i = t; i < n; i += NT) S.w2[i] = au[i]; }
                    g.sync();
                    biquad_scan(g, S.w2, n, lp, S.bq);
                    g.sync();
                    for (int i = t; i < n; i += NT) { const float y = S.w2[i]; if (stereo) outs[g0 + t0 + i] = make_float2(y, y); else outm[g0 + t0 + i] = y; }
                    g.sync();
                }
            }
        }
    }

    PC_TICK(12);
#ifdef PC_PROFILE
    if (blockIdx.x == 0 && t == 0)
        printf("pcprof mode %d: load %llu smeter %llu agcmag %llu slmax %llu histmove %llu aver %llu gain %llu atan %llu pll %llu dc %llu sqfir %llu slide %llu burstend %llu\n",
               mode, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6], tk[7], tk[8], tk[9], tk[10], tk[11], tk[12]);
#endif
    // ---------------- write the state back ----------------
    g.sync();
    if (do_agc && agc.on) {
        for (int i = t; i < D; i += NT) { g_dly[2 * i] = S.dl[i].x; g_dly[2 * i + 1] = S.dl[i].y; }
        if (pre) { const float *tail = a.magtail + (long)ch * PC_AGC_RING; for (int i = t; i < W1; i += NT) g_mag[i] = tail[i]; }
        else if constexpr (!LEAN) for (int i = t; i < W1; i += NT) g_mag[i] = S.mg[i];
    }
    if (fir && !defer) {
        PcFir *fw = const_cast<PcFir *>(fir);
        for (int i = t; i < nt - 1; i += NT) {
            if (mode == PC_MODE_FM || (mode == PC_MODE_AM && !stereo)) fw->zreal[i] = S.w0[i];
            else { fw->zr[i] = S.w0[i]; fw->zi[i] = mode == PC_MODE_AM ? S.w0[i] : S.w1[i]; }
        }
    }
    // each stage writes only its own state: the stages of one channel may run as separate,
    // concurrent launches (S-meter | AGC | demodulator pipeline of the batch chain)
    if (t == 0) {
        if (do_sm) C.sm = sm;
        if (do_agc) C.agc = agc;
        if (mode == PC_MODE_AM) C.am.z1 = am_z1;
        if (mode == PC_MODE_SAM) { C.sam.z1 = sam_z1; C.sam.y1 = sam_y1; C.sam.phase = sam_ph; C.sam.freq = sam_fr; }
        if (mode == PC_MODE_FM) {
            C.fm.phase = fm_ph; C.fm.freq = fm_fr; C.fm.err_dc = fm_dc;
            if (!defer) { C.fm.sq_ave = fm_sq; C.fm.squelched = fm_squelched; C.fm.lp = lp; }   // else: fm_squelch_decide_kernel's
        }
    }
}

// =====================================================================================================
// CAgc's log magnitudes and their sliding maximum (agc.cpp:196-231) for every sample of a call, ahead of the walk
// (PC_AGC_PRE).  Neither needs state from the loop -- only the win_n-1 magnitudes in front of a burst, which are
// the input's own -- so they leave the sequential walk (a fifth of an FM tile, a third of an SSB one) for one
// workgroup per (channel, group of bursts).  The same fp32 operations in the same order: the peaks, and with them
// every audio word, are what the walk computes itself.
// =====================================================================================================
__global__ __launch_bounds__(256)
void agc_peaks_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_PEAKS);
    using G = Wg<4>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    PreLds &S = *reinterpret_cast<PreLds *>(pc_smem);
    const int ngrp = (a.nbursts + a.pre_bpw - 1) / a.pre_bpw;
    const int t = threadIdx.x, ch = blockIdx.x / ngrp, b0 = (blockIdx.x % ngrp) * a.pre_bpw;
    const int b1 = b0 + a.pre_bpw < a.nbursts ? b0 + a.pre_bpw : a.nbursts;
    if (a.out_rows && a.out_rows[ch] < 0) return;
    const PcChannel &C = a.chan[ch];
    if (!C.agc.on) return;                               // uniform per workgroup
    const G g{t, t & 63, t >> 6, &S.sy};
    const int W1 = C.agc.win_n > 0 ? C.agc.win_n - 1 : 0;
    const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)ch * a.in_stride;
    const float *g_mag = a.agc_mag + (long)ch * PC_AGC_RING;   // the last W1 magnitudes of the previous call
    float *pkrow = a.pkbuf + (long)ch * a.nbursts * a.burst;
    auto logmag = [](float2 v) {
        float m = fabsf(v.x);
        const float mi = fabsf(v.y);
        if (mi > m) m = mi;
        return log10f(m + refc::AGC_MIN_CONSTANT_F) - refc::AGC_LOG10_MAX_AMPLITUDE_F;     // agc.cpp:196-201
    };
    const long p0 = (long)b0 * a.burst, p1 = (long)b1 * a.burst, total = (long)a.nbursts * a.burst;
    // the window in front of this group's first sample
    for (int i = t; i < W1; i += NT) {
        const long k = p0 - W1 + i;
        S.mg[i] = k >= 0 ? logmag(in[k]) : g_mag[W1 + (int)k];
    }
    for (long pos = p0; pos < p1; ) {
        const long bend = (pos / a.burst + 1) * a.burst;  // tiles do not straddle bursts (as in the walk)
        const int n = (int)((bend - pos) < PT ? (bend - pos) : PT);
        float *mg = S.mg + W1;
        for (int i = t; i < n; i += NT) mg[i] = logmag(in[pos + i]);
        g.sync();
        sliding_max(g, S, W1, n);
        for (int i = t; i < n; i += NT) pkrow[pos + i] = S.pk[i];
        {   // the last W1 magnitudes are the next tile's window: read all, one barrier, write all
            float keepm[PH / NT];
#pragma unroll
            for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < W1) keepm[j] = S.mg[n + i]; }
            g.sync();
#pragma unroll
            for (int j = 0; j < PH / NT; j++) { const int i = t + NT * j; if (i < W1) S.mg[i] = keepm[j]; }
            g.sync();
        }
        pos += n;
    }
    if (p1 == total) {                                   // the window the next call starts with
        float *tail = a.magtail + (long)ch * PC_AGC_RING;
        for (int i = t; i < W1; i += NT) tail[i] = S.mg[i];
    }
}

hipError_t agc_peaks_launch(const PcArgs &a, hipStream_t stream)
{
    static_assert(sizeof(PreLds) <= 40 * 1024, "four peaks workgroups per CU");
    PcArgs b = a;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long slots = 4L * (cus > 0 ? cus : 256);
    b.pre_bpw = (int)(((long)a.channels * a.nbursts + slots - 1) / slots);
    if (b.pre_bpw < 1) b.pre_bpw = 1;
    const int ngrp = (a.nbursts + b.pre_bpw - 1) / b.pre_bpw;
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL(agc_peaks_kernel, dim3(a.channels * ngrp), dim3(256), sizeof(PreLds), stream, b);
    return hipGetLastError();
}

// =====================================================================================================
// CSMeter (smeter.cpp:62-93) over a WHOLE call, out of the walk (round 4).  The meter feeds nothing -- its averages
// go to the GUI -- and both of them are compositions of associative maps (the attack average an affine one, the decay
// average x -> max(A x + B, C)), so nothing about them is sequential: one workgroup per receiver takes the call in
// super-tiles of 8192 samples, a thread owning 32 CONSECUTIVE samples of a super-tile (power -> dB once, staged in
// LDS with a pad word every 32 so that both the coalesced fill and the per-thread runs are conflict-free), and pays
// two workgroup scans per 8192 samples where the walk paid two per 1024 -- a quarter to a half of a walked tile's
// time (S-meter alone as a walk: 4.5 us per 1024 samples, 290 us per call; here ~20 us).  The same fp64 maps as
// smeter_tile, chunked differently: states agree to rounding (1e-15 relative).
// =====================================================================================================
constexpr int SM_ST = 8192, SM_LC = SM_ST / 256;
struct SmLds {
    PcSync sy;
    alignas(16) float db[SM_ST + SM_ST / 32];
    double pw[SM_LC + 1];                                // (1 - att_a)^k
};
__global__ __launch_bounds__(256)
void smeter_call_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_SMETER);
    using G = Wg<4>;
    __shared__ SmLds S;
    const int t = threadIdx.x, ch = blockIdx.x;
    if (a.out_rows && a.out_rows[ch] < 0) return;
    const G g{t, t & 63, t >> 6, &S.sy};
    PcChannel &C = a.chan[ch];
    PcSMeter sm = C.sm;
    const double aa = sm.att_a, ia = 1.0 - sm.att_a, da = sm.dec_a, id = 1.0 - sm.dec_a;
    if (t <= SM_LC) { double p = 1.0; for (int k = 0; k < t; k++) p *= ia; S.pw[t] = p; }
    const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)ch * a.in_stride;
    const long total = (long)a.nbursts * a.burst;
    for (long p0 = 0; p0 < total; p0 += SM_ST) {
        const int n = (int)((total - p0) < SM_ST ? (total - p0) : SM_ST);
        g.sync();                                        // the previous super-tile's runs have been read (and S.pw written)
        for (int i = t; i < n; i += 256) {
            const float2 v = in[p0 + i];
            const float pw = (v.x * v.x + v.y * v.y) * refc::SM_INV_MAX_PWR_F;
            S.db[i + (i >> 5)] = pw > 0.f ? 10.0f * log10f(pw) : -500.0f;
        }
        g.sync();
        const int base = SM_LC * t;
        int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > SM_LC ? SM_LC : cnt);
        const float *run = S.db + base + t;              // = index(base + j) for j < 32
        double p = 0.0, pk = -1.0e300;
        for (int j = 0; j < cnt; j++) { const double x = (double)run[j]; p = ia * p + aa * x; pk = fmax(pk, x); }
        double A = S.pw[cnt], B = p, At, Bt;
        g.scan1_max(A, B, At, Bt, pk);
        double att = A * sm.att_ave + B;                 // attack average entering this thread's run
        const double att_end = At * sm.att_ave + Bt;
        double MA = 1.0, MB = 0.0, MC = -1.0e300, TA, TB, TC;
        for (int j = 0; j < cnt; j++) {
            const double x = (double)run[j];
            att = ia * att + aa * x;
            MA = id * MA; MB = id * MB + da * x; MC = fmax(id * MC + da * x, att);
        }
        g.scan_max(MA, MB, MC, TA, TB, TC);
        const double dec_end = fmax(TA * sm.dec_ave + TB, TC);
        sm.att_ave = att_end; sm.dec_ave = dec_end; sm.ave_mag = dec_end; sm.peak_mag = fmax(sm.peak_mag, pk);
    }
    if (t == 0) C.sm = sm;
}
hipError_t smeter_call_launch(const PcArgs &a, hipStream_t stream)
{
    PcArgs b = a;
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL(smeter_call_kernel, dim3(a.channels), dim3(256), 0, stream, b);
    return hipGetLastError();
}

// =====================================================================================================
// The squelch half of CFmDemod (fmdemod.cpp:113-152), deferred out of the walk (PC_FM_DEFER).  None of it feeds
// back into the loop: the high-pass CFir over the audio, the average of its magnitude, the once-per-burst hysteresis
// decision and the CIir low-pass of the bursts that stay open are post-processing of the raw audio the walk leaves in
// the output rows -- a fifth of an FM tile's time in the sequential walk, a few microseconds as burst-parallel work:
//   1. fm_squelch_maps_kernel, one workgroup per (channel, burst): the burst's high-pass + |.| + average and its
//      low-pass, both from a ZERO start state, reduced to their affine maps (average: s -> A s + B; filter state:
//      s -> M s + v) -- the same scans the walk used, so the same words come out;
//   2. fm_squelch_decide_kernel, one workgroup per channel: the maps chained through the bursts in order (64 small
//      steps), the decision of every burst, the filter state each open burst starts from; the channel's state;
//   3. fm_squelch_apply_kernel, one workgroup per (channel, burst): zeros, or the low-pass from that start state.
// =====================================================================================================
template <int NW>
__device__ __forceinline__ void sq_lp_maps(const Wg<NW> &g, const float *sq, const float *x, int n, double a, double gn,
                                           const double *apw, const PcIir &f, const double *tab,
                                           double &At, double &Bt, double (&mt)[4], double (&vt)[2])
{
    constexpr int LC = Wg<NW>::LC;
    const int base = LC * g.t;
    int cnt = n - base; cnt = cnt < 0 ? 0 : (cnt > LC ? LC : cnt);
    double p = 0.0, w1 = 0.0, w2 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; j++) {
        const float sv = j < cnt ? sq[base + j] : 0.f;
        p = j < cnt ? a * p + gn * (double)sv : p;
        const double xv = j < cnt ? (double)x[base + j] : 0.0;
        const double w0 = xv - f.a1 * w1 - f.a2 * w2;
        if (j < cnt) { w2 = w1; w1 = w0; }
    }
    double A = apw[cnt], B = p;
    double m[4] = {tab[4 * cnt], tab[4 * cnt + 1], tab[4 * cnt + 2], tab[4 * cnt + 3]}, v[2] = {w1, w2};
    g.scan1_2(A, B, At, Bt, m, v, mt, vt);
}


__global__ __launch_bounds__(256)
void fm_squelch_maps_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_SQ_MAPS);
    using G = Wg<4>;
    constexpr int NT = G::NT, LC = G::LC;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    SqLds &S = *reinterpret_cast<SqLds *>(pc_smem);
    const int ngrp = (a.nbursts + a.sq_bpw - 1) / a.sq_bpw;
    const int t = threadIdx.x, ch = blockIdx.x / ngrp, b0 = (blockIdx.x % ngrp) * a.sq_bpw;
    const int b1 = b0 + a.sq_bpw < a.nbursts ? b0 + a.sq_bpw : a.nbursts;
    if (a.out_rows && a.out_rows[ch] < 0) return;
    const PcChannel &C = a.chan[ch];
    if (C.mode != PC_MODE_FM) return;                    // uniform per workgroup
    const G g{t, t & 63, t >> 6, &S.sy};
    const bool stereo = a.flags & PC_STEREO;
    const long orow = (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    const float *outm = a.out + orow;
    const float2 *outs = reinterpret_cast<const float2 *>(a.out) + orow;
    auto raw = [&](long i) -> float { return stereo ? outs[i].x : outm[i]; };
    const PcFir &fir = C.fm.hp;
    const int nt = fir.ntaps;
    for (int i = t; i < PC_FIR_MAX + 17; i += NT) {
        const int k = nt - 1 - (i - 4);                  // reversed, four zeros in front, zeros behind (as in the walk)
        S.h0[i] = (k >= 0 && k < nt) ? fir.coef[k] : 0.f;
    }
    for (int i = t; i < PT + PC_FIR_MAX + 17; i += NT) S.w0[i] = 0.f;
    pow_table(S.pw_sq, 1.0 - C.fm.sq_alpha, t);
    PcIir lp = C.fm.lp;
    biquad_table(S.bq, lp, t);
    g.sync();
    // the high-pass delay line at the start of this group's first burst: the audio in front of it (still raw: the apply
    // kernel runs after every maps workgroup has finished), the saved delay line in front of the call
    {
        const long g0 = (long)b0 * a.burst;
        for (int i = t; i < nt - 1; i += NT) {
            const long k = g0 - (nt - 1) + i;
            S.w0[i] = k >= 0 ? raw(k) : fir.zreal[i];    // (k < 0 only for burst 0: burst >= PC_FIR_MAX > ntaps - 1)
        }
    }
    float *au = S.w0 + (nt - 1);
    // the audio of the next tile is fetched while the current one goes through its filter and scan
    float nx[LC];
    const long first = (long)b0 * a.burst, end = (long)b1 * a.burst;
    auto fetch = [&](long p) {
#pragma unroll
        for (int j = 0; j < LC; j++) { const long i = p + t + NT * j; nx[j] = i < end ? raw(i) : 0.f; }
    };
    fetch(first);
    for (int b = b0; b < b1; b++) {
        const long g0 = (long)b * a.burst;
        double A_tot = 1.0, B_tot = 0.0, M_tot[4] = {1.0, 0.0, 0.0, 1.0}, v_tot[2] = {0.0, 0.0};
        for (int t0 = 0; t0 < a.burst; t0 += PT) {
            const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
#pragma unroll
            for (int j = 0; j < LC; j++) { const int i = t + NT * j; if (i < n) au[i] = nx[j]; }
            fetch(g0 + t0 + n);                          // bursts are contiguous: the next tile follows
            g.sync();
            float acc[LC];
            fir_blk<4>(S.h0, nt, S.w0, t, acc);
#pragma unroll
            for (int j = 0; j < LC; j++) S.w2[pc_out_index<4>(t, j) & (PT - 1)] = fabsf(acc[j]);
            g.sync();
            double At, Bt, mt[4], vt[2];
            sq_lp_maps(g, S.w2, au, n, 1.0 - C.fm.sq_alpha, C.fm.sq_alpha, S.pw_sq, lp, S.bq, At, Bt, mt, vt);
            // this tile's maps behind what the burst has so far
            B_tot = At * B_tot + Bt; A_tot = At * A_tot;
            const double n0 = mt[0] * M_tot[0] + mt[1] * M_tot[2], n1 = mt[0] * M_tot[1] + mt[1] * M_tot[3];
            const double n2 = mt[2] * M_tot[0] + mt[3] * M_tot[2], n3 = mt[2] * M_tot[1] + mt[3] * M_tot[3];
            const double u0 = mt[0] * v_tot[0] + mt[1] * v_tot[1] + vt[0], u1 = mt[2] * v_tot[0] + mt[3] * v_tot[1] + vt[1];
            M_tot[0] = n0; M_tot[1] = n1; M_tot[2] = n2; M_tot[3] = n3; v_tot[0] = u0; v_tot[1] = u1;
            g.sync();
            slide(g, S.w0, nt - 1, n);                   // the next tile's / burst's delay line
            g.sync();
        }
        if (t == 0) {
            double *r = a.sqbuf + ((long)ch * a.nbursts + b) * PC_SQ_REC;
            r[0] = A_tot; r[1] = B_tot; r[2] = M_tot[0]; r[3] = M_tot[1]; r[4] = M_tot[2]; r[5] = M_tot[3]; r[6] = v_tot[0]; r[7] = v_tot[1];
        }
    }
}

__global__ __launch_bounds__(64)
void fm_squelch_decide_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_SQ_DECIDE);
    const int t = threadIdx.x, ch = blockIdx.x;
    if (a.out_rows && a.out_rows[ch] < 0) return;
    PcChannel &C = a.chan[ch];
    if (C.mode != PC_MODE_FM) return;
    const bool stereo = a.flags & PC_STEREO;
    const long orow = (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    const float *outm = a.out + orow;
    const float2 *outs = reinterpret_cast<const float2 *>(a.out) + orow;
    PcFm &F = C.fm;
    // the high-pass delay line after the call: the last ntaps-1 raw audio samples (the apply kernel has not run yet)
    const int nt = F.hp.ntaps;
    const long total = (long)a.nbursts * a.burst;
    float keep = 0.f;
    if (t < nt - 1) {
        const long k = total - (nt - 1) + t;
        keep = k >= 0 ? (stereo ? outs[k].x : outm[k]) : F.hp.zreal[t + (int)total];
    }
    __syncthreads();
    if (t < nt - 1) F.hp.zreal[t] = keep;
    // the bursts' maps come in 64 at a time, one record per thread (a walk over global memory would pay a dependent
    // miss per burst); thread 0 chains them
    __shared__ double rec[64][8];
    __shared__ double res[64][3];
    double sq = F.sq_ave, w1 = F.lp.w1a, w2 = F.lp.w2a;
    int sqd = F.squelched;
    for (int c0 = 0; c0 < a.nbursts; c0 += 64) {
        const int cn = a.nbursts - c0 < 64 ? a.nbursts - c0 : 64;
        __syncthreads();
        if (t < cn) {
            const double *r = a.sqbuf + ((long)ch * a.nbursts + c0 + t) * PC_SQ_REC;
#pragma unroll
            for (int k = 0; k < 8; k++) rec[t][k] = r[k];
        }
        __syncthreads();
        if (t == 0) {
            for (int b = 0; b < cn; b++) {
                const double *r = rec[b];
                sq = r[0] * sq + r[1];
                if (0 == F.sq_thresh) sqd = 1;                                  // fmdemod.cpp:128-151
                else if (sqd) { if (sq < (F.sq_thresh - refc::FM_SQUELCH_HYSTERESIS)) sqd = 0; }
                else { if (sq >= (F.sq_thresh + refc::FM_SQUELCH_HYSTERESIS)) sqd = 1; }
                res[b][0] = (double)sqd; res[b][1] = w1; res[b][2] = w2;
                if (!sqd) {                                                     // the low-pass runs on open bursts only
                    const double nw1 = r[2] * w1 + r[3] * w2 + r[6], nw2 = r[4] * w1 + r[5] * w2 + r[7];
                    w1 = nw1; w2 = nw2;
                }
            }
        }
        __syncthreads();
        if (t < cn) {
            double *r = a.sqbuf + ((long)ch * a.nbursts + c0 + t) * PC_SQ_REC;
            r[8] = res[t][0]; r[9] = res[t][1]; r[10] = res[t][2];
        }
    }
    if (t == 0) { F.sq_ave = sq; F.squelched = sqd; F.lp.w1a = w1; F.lp.w2a = w2; }
}

__global__ __launch_bounds__(256)
void fm_squelch_apply_kernel(PcArgs a)
{
    CSDR_WG_TRACE_SCOPE(a.trace, WGT_SQ_APPLY);
    using G = Wg<4>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    SqLds &S = *reinterpret_cast<SqLds *>(pc_smem);
    const int ngrp = (a.nbursts + a.sq_bpw - 1) / a.sq_bpw;
    const int t = threadIdx.x, ch = blockIdx.x / ngrp, b0 = (blockIdx.x % ngrp) * a.sq_bpw;
    const int b1 = b0 + a.sq_bpw < a.nbursts ? b0 + a.sq_bpw : a.nbursts;
    if (a.out_rows && a.out_rows[ch] < 0) return;
    const PcChannel &C = a.chan[ch];
    if (C.mode != PC_MODE_FM) return;
    const G g{t, t & 63, t >> 6, &S.sy};
    const bool stereo = a.flags & PC_STEREO;
    const long orow = (long)(a.out_rows ? a.out_rows[ch] : ch) * a.out_stride;
    float *outm = a.out + orow;
    float2 *outs = reinterpret_cast<float2 *>(a.out) + orow;
    PcIir lp = C.fm.lp;
    biquad_table(S.bq, lp, t);
    g.sync();
    for (int b = b0; b < b1; b++) {
        const double *r = a.sqbuf + ((long)ch * a.nbursts + b) * PC_SQ_REC;
        const long g0 = (long)b * a.burst;
        if (r[8] != 0.0) {                               // squelched: zeros (fmdemod.cpp:139-143); uniform per workgroup
            for (int i = t; i < a.burst; i += NT) { if (stereo) outs[g0 + i] = make_float2(0.f, 0.f); else outm[g0 + i] = 0.f; }
            continue;
        }
        lp.w1a = r[9]; lp.w2a = r[10];
        for (int t0 = 0; t0 < a.burst; t0 += PT) {
            const int n = (a.burst - t0) < PT ? (a.burst - t0) : PT;
            for (int i = t; i < n; i += NT) S.w2[i] = stereo ? outs[g0 + t0 + i].x : outm[g0 + t0 + i];
            g.sync();
            biquad_scan(g, S.w2, n, lp, S.bq);
            g.sync();
            for (int i = t; i < n; i += NT) { const float y = S.w2[i]; if (stereo) outs[g0 + t0 + i] = make_float2(y, y); else outm[g0 + t0 + i] = y; }
            g.sync();
        }
    }
}

hipError_t fm_squelch_launch(const PcArgs &a, hipStream_t stream)
{
    static_assert(sizeof(SqLds) <= 20 * 1024, "eight squelch workgroups per CU");
    // bursts per workgroup: as few as still give every workgroup a slot in ONE round (four 256-thread workgroups per
    // CU at these kernels' 105-122 registers), so that taps, tables and zeroing are paid once per slot
    PcArgs b = a;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long slots = 4L * (cus > 0 ? cus : 256);
    b.sq_bpw = (int)(((long)a.channels * a.nbursts + slots - 1) / slots);
    if (b.sq_bpw < 1) b.sq_bpw = 1;
    const int ngrp = (a.nbursts + b.sq_bpw - 1) / b.sq_bpw;
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL(fm_squelch_maps_kernel, dim3(a.channels * ngrp), dim3(256), sizeof(SqLds), stream, b);
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL(fm_squelch_decide_kernel, dim3(a.channels), dim3(64), 0, stream, b);
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL(fm_squelch_apply_kernel, dim3(a.channels * ngrp), dim3(256), sizeof(SqLds), stream, b);
    return hipGetLastError();
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

// waves per channel: four when the channels alone cannot fill the chip, else one.
// CSDR_POSTCHAIN_WAVES=1|4|8 overrides (measurements: 4 beats 8 on 256 channels, 3.8 vs 4.1 ms).
// CSMeter::GetAve / GetPeak (dsp/smeter.cpp:98-112) of every channel of a unit at once, without moving the
// channel state: ave[row] = average + 5 dB, peak[row] = peak + 5 dB and the peak is reset to 0 as GetPeak
// does.  rows: optional output index of each channel.
template <class T>
__global__ void smeter_collect_kernel(PcChannel *chan, int channels, const int *rows, T *ave, T *peak)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= channels) return;
    const int o = rows ? rows[c] : c;
    if (o < 0) return;                  // a row whose receiver has moved to another plan group (csdr_demod_batch_set_demod)
    if (ave) ave[o] = (T)(chan[c].sm.ave_mag + refc::SM_CALIBRATION);
    if (peak) {
        peak[o] = (T)(chan[c].sm.peak_mag + refc::SM_CALIBRATION);
        chan[c].sm.peak_mag = 0.0;
    }
}
hipError_t smeter_collect_launch(PcChannel *chan, int channels, const int *rows, float *ave, float *peak, hipStream_t stream)
{
    hipLaunchKernelGGL(smeter_collect_kernel<float>, dim3((channels + 255) / 256), dim3(256), 0, stream, chan, channels, rows, ave, peak);
    return hipGetLastError();
}
hipError_t smeter_collect_launch(PcChannel *chan, int channels, const int *rows, double *ave, double *peak, hipStream_t stream)
{
    hipLaunchKernelGGL(smeter_collect_kernel<double>, dim3((channels + 255) / 256), dim3(256), 0, stream, chan, channels, rows, ave, peak);
    return hipGetLastError();
}

template <int NW, bool LEAN>
static hipError_t pc_launch_nw(const PcArgs &a, hipStream_t stream)
{
    // once per device and instantiation (the attribute belongs to the device, and a process may drive several)
    if (sizeof(PcLds) > 64 * 1024) {
        hipError_t e = CSDR_MAX_LDS_ONCE((&postchain_kernel<NW, LEAN>), sizeof(PcLds));
        if (e != hipSuccess) return e;
    }
    PcArgs b = a;
#ifdef CSDR_WG_TRACE
    b.trace = wgtrace_next();
#endif
    hipLaunchKernelGGL((postchain_kernel<NW, LEAN>), dim3(a.channels), dim3(64 * NW), sizeof(PcLds), stream, b);
    return hipGetLastError();
}
hipError_t postchain_launch(const PcArgs &a, hipStream_t stream)
{
    static int forced = -1;
    if (forced < 0) { const char *env = getenv("CSDR_POSTCHAIN_WAVES"); forced = env ? atoi(env) : 0; }
    int nw = a.channels <= 1024 ? 4 : 1;
    if (forced == 1 || forced == 4 || forced == 8) nw = forced;
    // the lean walk (four waves): no S-meter in the launch, AGC peaks precomputed for every receiver that has the AGC
    // on, the squelch of every FM receiver deferred -- the caller (PcUnit::run) says so with PC_LEAN
    static const bool lean_ok = !(getenv("CSDR_PC_LEAN") && atoi(getenv("CSDR_PC_LEAN")) == 0);
    if (nw == 4 && (a.flags & PC_LEAN) && lean_ok) return pc_launch_nw<4, true>(a, stream);
    switch (nw) {
    case 8:  return pc_launch_nw<8, false>(a, stream);
    case 4:  return pc_launch_nw<4, false>(a, stream);
    default: return pc_launch_nw<1, false>(a, stream);
    }
}

}  // namespace csdr
