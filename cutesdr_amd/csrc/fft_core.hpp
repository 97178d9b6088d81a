// fft_core.hpp -- register-resident radix-2^k butterflies for gfx950 (wave64, packed fp32).
//
// Building blocks of the LDS-staged complex FFT used by the overlap-save FIR (CFastFIR,
// reference dsp/fastfir.cpp:268-306 + dsp/fft.cpp:416-426) and the display spectrum (CFft,
// dsp/fft.cpp:267-288).  All loops are fully unrolled over compile-time register indices,
// twiddles W_32^k are immediates; complex values are float2 ext-vectors so that hipcc can
// emit v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32.
//
// Sign convention follows the reference: SIGN=+1 is CFft::FwdFFT (X[k] = sum x[n] e^{+j2pi nk/N}),
// SIGN=-1 is CFft::RevFFT; neither normalises.
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// cos(2*pi*k/32), k = 0..31
static constexpr float kCos32[32] = {
    1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
    0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
    0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
    -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f,
    -1.0f, -0.98078528040323044913f, -0.92387953251128675613f, -0.83146961230254523708f,
    -0.70710678118654752440f, -0.55557023301960222474f, -0.38268343236508977173f, -0.19509032201612826785f,
    0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
    0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f};

// ---- packed complex arithmetic ---------------------------------------------------------------
// On the device the swizzles and sign flips of complex products ride on the VOP3P op_sel / neg
// modifiers (one instruction each) instead of v_xor/v_mov pairs.  Every asm statement holds ONE
// instruction, so hipcc's hazard recogniser still pads the 1-wait-state "packed fp32 result ->
// next VALU" hazard around it (checked in the ISA).  The host build (unit tests of the butterfly
// algebra) uses the plain C++ forms.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CSDR_NO_PK_ASM)
#define CSDR_PK_ASM 1
#else
#define CSDR_PK_ASM 0
#endif

// product with a compile-time constant twiddle: hipcc keeps w and its rotation in SGPR pairs
__host__ __device__ __forceinline__ v2f cmul_c(v2f a, v2f w)
{
    v2f wr = {-w.y, w.x};
    return a.xx * w + a.yy * wr;
}

// a * w, w a run-time value
__host__ __device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
#if CSDR_PK_ASM
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#else
    return cmul_c(a, w);
#endif
}
// a * conj(w)
__host__ __device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)
{
#if CSDR_PK_ASM
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#else
    v2f wc = {w.x, -w.y}, wr = {w.y, w.x};
    return a.xx * wc + a.yy * wr;
#endif
}
// u + S*j*v  (S = +1 or -1)
template <int S>
__host__ __device__ __forceinline__ v2f add_jv(v2f u, v2f v)
{
#if CSDR_PK_ASM
    v2f r;
    if constexpr (S > 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(u), "v"(v));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(v));
    return r;
#else
    return S > 0 ? v2f{u.x - v.y, u.y + v.x} : v2f{u.x + v.y, u.y - v.x};
#endif
}
// S*j*(u - v)
template <int S>
__host__ __device__ __forceinline__ v2f sub_j(v2f u, v2f v)
{
#if CSDR_PK_ASM
    v2f r;
    if constexpr (S > 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(v));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(u), "v"(v));
    return r;
#else
    return S > 0 ? v2f{-(u.y - v.y), u.x - v.x} : v2f{u.y - v.y, -(u.x - v.x)};
#endif
}

constexpr float kSqrtHalf = 0.70710678118654752440f;

// DIF butterfly: a' = u + v, b' = (u - v) * e^{SIGN j 2 pi K/32}, K in [0,16)
template <int K, int SIGN>
__host__ __device__ __forceinline__ void bfly_dif(v2f &xa, v2f &xb)
{
    const v2f u = xa, v = xb;
    xa = u + v;
    if constexpr (K == 0) {
        xb = u - v;
    } else if constexpr (K == 8) {
        xb = sub_j<SIGN>(u, v);
    } else if constexpr (K == 4) {
        const v2f d = u - v;
        xb = add_jv<SIGN>(d, d) * kSqrtHalf;            // d (1 + S j)/sqrt2
    } else if constexpr (K == 12) {
        const v2f d = u - v;
        xb = add_jv<-SIGN>(d, d) * (-kSqrtHalf);        // d (-1 + S j)/sqrt2
    } else {
        constexpr float c = kCos32[K];
        constexpr float s = (SIGN > 0 ? 1.0f : -1.0f) * kCos32[(K + 24) & 31];
        xb = cmul_c(u - v, v2f{c, s});
    }
}
// FMA forms of the decimation-in-time butterfly (translation units that define CSDR_FMA_BFLY):
//   a' = a + b w  as two packed FMAs (b.x (c, s) + a, then b.y (-s, c) + that),  b' = 2 a - a'
// three packed instructions instead of four (complex product, sum, difference); w in an SGPR pair.
#if CSDR_PK_ASM && defined(CSDR_FMA_BFLY)
#define CSDR_FMA_DIT 1
#else
#define CSDR_FMA_DIT 0
#endif
// a' = a + b*w, b' = a - b*w, w a run-time (VGPR) or compile-time value
__host__ __device__ __forceinline__ void bfly_fma(v2f &xa, v2f &xb, v2f w)
{
#if CSDR_FMA_DIT
    v2f t, a;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(xb), "v"(w), "v"(xa));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "v"(xb), "v"(w), "v"(t));
    xb = xa * 2.0f - a;
    xa = a;
#else
    const v2f v = cmul(xb, w);
    const v2f u = xa;
    xa = u + v; xb = u - v;
#endif
}
// a' = a + b*conj(w), b' = a - b*conj(w)
__host__ __device__ __forceinline__ void bfly_fma_conj(v2f &xa, v2f &xb, v2f w)
{
#if CSDR_FMA_DIT
    v2f t, a;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(t) : "v"(xb), "v"(w), "v"(xa));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(a) : "v"(xb), "v"(w), "v"(t));
    xb = xa * 2.0f - a;
    xa = a;
#else
    const v2f v = cmul_conj(xb, w);
    const v2f u = xa;
    xa = u + v; xb = u - v;
#endif
}

// DIT butterfly: v = b * e^{SIGN j 2 pi K/32}; a' = a + v, b' = a - v
template <int K, int SIGN>
__host__ __device__ __forceinline__ void bfly_dit(v2f &xa, v2f &xb)
{
    const v2f u = xa;
#if CSDR_FMA_DIT
    if constexpr (K != 0 && K != 8) {
        // (K = 4, 12 included: c = +-s = sqrt(1/2) needs no special case in this form)
        constexpr float c = kCos32[K];
        constexpr float s = (SIGN > 0 ? 1.0f : -1.0f) * kCos32[(K + 24) & 31];
#ifdef CSDR_PLAIN_CONST_FMA
        // experiment: no asm, both constant vectors as literals
        const v2f w1 = {c, s}, w2 = {-s, c};
        const v2f t = __builtin_elementwise_fma(xb.xx, w1, u);
        const v2f a = __builtin_elementwise_fma(xb.yy, w2, t);
        xb = u * 2.0f - a;
        xa = a;
        return;
#else
        v2f t, a;
        const v2f w = {c, s};
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(xb), "s"(w), "v"(u));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(a) : "v"(xb), "s"(w), "v"(t));
        xb = u * 2.0f - a;
        xa = a;
        return;
#endif
    }
#endif
    if constexpr (K == 0) {
        const v2f v = xb;
        xa = u + v; xb = u - v;
    } else if constexpr (K == 8) {
        const v2f v = xb;
        xa = add_jv<SIGN>(u, v); xb = add_jv<-SIGN>(u, v);
    } else if constexpr (K == 4) {
        const v2f v = add_jv<SIGN>(xb, xb) * kSqrtHalf;
        xa = u + v; xb = u - v;
    } else if constexpr (K == 12) {
        const v2f v = add_jv<-SIGN>(xb, xb) * (-kSqrtHalf);
        xa = u + v; xb = u - v;
    } else {
        constexpr float c = kCos32[K];
        constexpr float s = (SIGN > 0 ? 1.0f : -1.0f) * kCos32[(K + 24) & 31];
        const v2f v = cmul_c(xb, v2f{c, s});
        xa = u + v; xb = u - v;
    }
}

template <int R> __host__ __device__ constexpr int bitrev(int v)
{
    int r = 0;
    for (int m = 1; m < R; m <<= 1) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// compile-time loop
template <int I, int N, class F>
__host__ __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Decimation-in-frequency radix-R DFT over registers: natural order in, X[k] ends in x[bitrev(k)].
template <int LEN, int R, int SIGN>
__host__ __device__ __forceinline__ void dif_stage(v2f (&x)[R])
{
    constexpr int H = LEN / 2;
    static_for<0, R / LEN>([&](auto B) {
        static_for<0, H>([&](auto I) {
            constexpr int a = B.value * LEN + I.value, b = a + H;
            bfly_dif<I.value *(32 / LEN), SIGN>(x[a], x[b]);
        });
    });
    if constexpr (LEN > 2) dif_stage<LEN / 2, R, SIGN>(x);
}
template <int R, int SIGN>
__host__ __device__ __forceinline__ void dft_dif(v2f (&x)[R])
{
    if constexpr (R > 1) dif_stage<R, R, SIGN>(x);
}

// Decimation-in-time radix-R DFT: input y[k] in x[bitrev(k)], natural order out.
template <int LEN, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_stage(v2f (&x)[R])
{
    constexpr int H = LEN / 2;
    static_for<0, R / LEN>([&](auto B) {
        static_for<0, H>([&](auto I) {
            constexpr int a = B.value * LEN + I.value, b = a + H;
            bfly_dit<I.value *(32 / LEN), SIGN>(x[a], x[b]);
        });
    });
    if constexpr (LEN < R) dit_stage<LEN * 2, R, SIGN>(x);
}
template <int R, int SIGN>
__host__ __device__ __forceinline__ void dft_dit(v2f (&x)[R])
{
    if constexpr (R > 1) dit_stage<2, R, SIGN>(x);
}

// ---- the same transforms cut into groups, for passes that overlap their LDS traffic with the
//      butterflies (fastfir_kernels.hip): a radix-R DIF is  head<I> (I = 0..R/4-1: its first two
//      stages on the elements {I, I+R/4, I+R/2, I+3R/4}, needs only those four loaded), the
//      middle stages 8 (R = 32 only), then tail4<G> (G = 0..R/4-1: stages 4 and 2 on elements
//      4G..4G+3, which are final afterwards).  The DIT mirrors it: head4<G>, middle, tail<I>
//      (last two stages on {I, I+R/4, I+R/2, I+3R/4}, final afterwards).
template <int LEN, int R, int SIGN>
__host__ __device__ __forceinline__ void dif_single(v2f (&x)[R])
{
    constexpr int H = LEN / 2;
    static_for<0, R / LEN>([&](auto B) {
        static_for<0, H>([&](auto I) {
            constexpr int a = B.value * LEN + I.value, b = a + H;
            bfly_dif<I.value *(32 / LEN), SIGN>(x[a], x[b]);
        });
    });
}
template <int LEN, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_single(v2f (&x)[R])
{
    constexpr int H = LEN / 2;
    static_for<0, R / LEN>([&](auto B) {
        static_for<0, H>([&](auto I) {
            constexpr int a = B.value * LEN + I.value, b = a + H;
            bfly_dit<I.value *(32 / LEN), SIGN>(x[a], x[b]);
        });
    });
}
template <int I, int R, int SIGN>
__host__ __device__ __forceinline__ void dif_head(v2f (&x)[R])
{
    static_assert(R >= 8 && I < R / 4, "dif_head");
    constexpr int Q = R / 4;
    bfly_dif<I *(32 / R), SIGN>(x[I], x[I + 2 * Q]);                 // stage R
    bfly_dif<(I + Q) * (32 / R), SIGN>(x[I + Q], x[I + 3 * Q]);
    bfly_dif<I *(64 / R), SIGN>(x[I], x[I + Q]);                     // stage R/2
    bfly_dif<I *(64 / R), SIGN>(x[I + 2 * Q], x[I + 3 * Q]);
}
template <int G, int R, int SIGN>
__host__ __device__ __forceinline__ void dif_tail4(v2f (&x)[R])
{
    constexpr int a = 4 * G;
    bfly_dif<0, SIGN>(x[a], x[a + 2]);                               // stage 4
    bfly_dif<8, SIGN>(x[a + 1], x[a + 3]);
    bfly_dif<0, SIGN>(x[a], x[a + 1]);                               // stage 2
    bfly_dif<0, SIGN>(x[a + 2], x[a + 3]);
}
template <int G, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_head4(v2f (&x)[R])
{
    constexpr int a = 4 * G;
    bfly_dit<0, SIGN>(x[a], x[a + 1]);                               // stage 2
    bfly_dit<0, SIGN>(x[a + 2], x[a + 3]);
    bfly_dit<0, SIGN>(x[a], x[a + 2]);                               // stage 4
    bfly_dit<8, SIGN>(x[a + 1], x[a + 3]);
}
// dit_head4 with the conjugate pass twiddles of its four inputs folded in: x[4G+q] *= conj(tq) first
// (FIRST_PLAIN: x[4G] carries no twiddle).  The odd inputs' products ride in the FMA butterflies.
template <int G, int R, int SIGN, bool FIRST_PLAIN>
__host__ __device__ __forceinline__ void dit_head4_conjtw(v2f (&x)[R], v2f t0, v2f t1, v2f t2, v2f t3)
{
    constexpr int a = 4 * G;
    if constexpr (!FIRST_PLAIN) x[a] = cmul_conj(x[a], t0);
    x[a + 2] = cmul_conj(x[a + 2], t2);
    bfly_fma_conj(x[a], x[a + 1], t1);                               // stage 2
    bfly_fma_conj(x[a + 2], x[a + 3], t3);
    bfly_dit<0, SIGN>(x[a], x[a + 2]);                               // stage 4
    bfly_dit<8, SIGN>(x[a + 1], x[a + 3]);
}
// dit_head4 with run-time multipliers of its four inputs folded in: x[4G+q] *= tq first (the frequency response
// between the forward and the inverse transform of the overlap-save filter).  The odd inputs' products ride in
// the FMA butterflies: 10 packed instructions + 4 for the stage-4 pair instead of 8 + 8.
template <int G, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_head4_tw(v2f (&x)[R], v2f t0, v2f t1, v2f t2, v2f t3)
{
    constexpr int a = 4 * G;
    x[a] = cmul(x[a], t0);
    x[a + 2] = cmul(x[a + 2], t2);
    bfly_fma(x[a], x[a + 1], t1);                                    // stage 2
    bfly_fma(x[a + 2], x[a + 3], t3);
    bfly_dit<0, SIGN>(x[a], x[a + 2]);                               // stage 4
    bfly_dit<8, SIGN>(x[a + 1], x[a + 3]);
}
// DIT butterfly of which only the difference output is wanted: b' = a - b * e^{SIGN j 2 pi K/32} (a is left alone)
template <int K, int SIGN>
__host__ __device__ __forceinline__ void bfly_dit_lower(const v2f xa, v2f &xb)
{
    if constexpr (K == 0) {
        xb = xa - xb;
    } else if constexpr (K == 8) {
        xb = add_jv<-SIGN>(xa, xb);
    } else {
        constexpr float c = kCos32[K];
        constexpr float s = (SIGN > 0 ? 1.0f : -1.0f) * kCos32[(K + 24) & 31];
        const v2f w1 = {-c, -s}, w2 = {s, -c};
        const v2f t = __builtin_elementwise_fma(xb.xx, w1, xa);
        xb = __builtin_elementwise_fma(xb.yy, w2, t);
    }
}
// dit_tail whose caller keeps only the upper half of the transform's outputs (x[I + R/2], x[I + 3R/4]): the
// overlap-save filter discards the first half of every inverse transform (fastfir.cpp:291-300)
template <int I, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_tail_upper(v2f (&x)[R])
{
    static_assert(R >= 8 && I < R / 4, "dit_tail_upper");
    constexpr int Q = R / 4;
    bfly_dit<I *(64 / R), SIGN>(x[I], x[I + Q]);                     // stage R/2
    bfly_dit<I *(64 / R), SIGN>(x[I + 2 * Q], x[I + 3 * Q]);
    bfly_dit_lower<I *(32 / R), SIGN>(x[I], x[I + 2 * Q]);           // stage R, difference outputs only
    bfly_dit_lower<(I + Q) * (32 / R), SIGN>(x[I + Q], x[I + 3 * Q]);
}
template <int I, int R, int SIGN>
__host__ __device__ __forceinline__ void dit_tail(v2f (&x)[R])
{
    static_assert(R >= 8 && I < R / 4, "dit_tail");
    constexpr int Q = R / 4;
    bfly_dit<I *(64 / R), SIGN>(x[I], x[I + Q]);                     // stage R/2
    bfly_dit<I *(64 / R), SIGN>(x[I + 2 * Q], x[I + 3 * Q]);
    bfly_dit<I *(32 / R), SIGN>(x[I], x[I + 2 * Q]);                 // stage R
    bfly_dit<(I + Q) * (32 / R), SIGN>(x[I + Q], x[I + 3 * Q]);
}

// w^k for k = 0..R-1 by a log-depth product tree (pw[0] unused = 1)
template <int R>
__host__ __device__ __forceinline__ void twiddle_powers(v2f w, v2f (&pw)[R])
{
    pw[0] = v2f{1.0f, 0.0f};
    if constexpr (R > 1) pw[1] = w;
    static_for<2, R>([&](auto K) {
        constexpr int k = K.value;
        constexpr int hb = (k & (k - 1)) == 0 ? k / 2 : (1 << (31 - __builtin_clz(k)));
        pw[k] = cmul(pw[hb], pw[k - hb]);
    });
}

// Hide a loop-invariant value from LICM (used where recomputing is cheaper than the registers
// a hoisted copy would pin for the whole block loop).
__host__ __device__ __forceinline__ v2f opaque(v2f v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// 8-byte LDS access that stays a single ds_read_b64 / ds_write_b64 in program order
__device__ __forceinline__ v2f lds_ld8(const v2f *p)
{
    return __builtin_bit_cast(v2f, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WAVEFRONT));
}
__device__ __forceinline__ void lds_st8(v2f *p, v2f v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}


// LDS index padding: 2 elements (16 B) every 32 elements keeps float4 alignment and makes the
// stride-32 / stride-1024 access patterns of the three passes bank-conflict free.
__host__ __device__ __forceinline__ int lds_pad(int pos) { return pos + ((pos >> 5) << 1); }

}  // namespace csdr
