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

__host__ __device__ __forceinline__ v2f cmul(v2f a, v2f w)
{
    v2f wr = {-w.y, w.x};
    return a.xx * w + a.yy * wr;
}
__host__ __device__ __forceinline__ v2f cmul_conj(v2f a, v2f w)      // a * conj(w)
{
    v2f wc = {w.x, -w.y}, wr = {w.y, w.x};
    return a.xx * wc + a.yy * wr;
}
// a * (+j) and a * (-j)
__host__ __device__ __forceinline__ v2f mul_pj(v2f a) { return v2f{-a.y, a.x}; }
__host__ __device__ __forceinline__ v2f mul_mj(v2f a) { return v2f{a.y, -a.x}; }

// a * e^{SIGN * j * 2*pi*K/32}, K in [0,16): trivial rotations cost no multiplies
template <int K, int SIGN>
__host__ __device__ __forceinline__ v2f mul_w32(v2f a)
{
    if constexpr (K == 0) {
        return a;
    } else if constexpr (K == 8) {
        return SIGN > 0 ? mul_pj(a) : mul_mj(a);
    } else if constexpr (K == 4) {
        constexpr float h = 0.70710678118654752440f;
        v2f r = SIGN > 0 ? mul_pj(a) : mul_mj(a);
        return (a + r) * h;
    } else if constexpr (K == 12) {
        constexpr float h = 0.70710678118654752440f;
        v2f r = SIGN > 0 ? mul_pj(a) : mul_mj(a);
        return (r - a) * h;
    } else {
        constexpr float c = kCos32[K];
        constexpr float s = (SIGN > 0 ? 1.0f : -1.0f) * kCos32[(K + 24) & 31];
        v2f w = {c, s};
        return cmul(a, w);
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
            v2f u = x[a], v = x[b];
            x[a] = u + v;
            x[b] = mul_w32<I.value *(32 / LEN), SIGN>(u - v);
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
            v2f u = x[a], v = mul_w32<I.value *(32 / LEN), SIGN>(x[b]);
            x[a] = u + v;
            x[b] = u - v;
        });
    });
    if constexpr (LEN < R) dit_stage<LEN * 2, R, SIGN>(x);
}
template <int R, int SIGN>
__host__ __device__ __forceinline__ void dft_dit(v2f (&x)[R])
{
    if constexpr (R > 1) dit_stage<2, R, SIGN>(x);
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

// LDS index padding: 2 elements (16 B) every 32 elements keeps float4 alignment and makes the
// stride-32 / stride-1024 access patterns of the three passes bank-conflict free.
__host__ __device__ __forceinline__ int lds_pad(int pos) { return pos + ((pos >> 5) << 1); }

}  // namespace csdr
