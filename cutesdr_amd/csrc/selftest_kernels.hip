// selftest_kernels.hip -- device-side unit checks of the packed-complex primitives of
// fft_core.hpp (the VOP3P-modifier asm forms only exist on the device).  Test-only entry
// point csdr__selftest_fft(), not part of the public ABI.
#include "capi_common.hpp"
#include "fft_core.hpp"
#include "stream_pool.hpp"

namespace csdr {

// out layout (v2f): [0] cmul, [1] cmul_conj, [2] add_jv<+1>, [3] add_jv<-1>, [4] sub_j<+1>,
// [5] sub_j<-1>, [8..39] dif32(+1), [40..71] dit32(-1) of the dif result, [72..87] dif16(+1),
// [88..103] pw[16] of in[1]
__global__ void selftest_kernel(const v2f *in, v2f *out)
{
    const int lane = threadIdx.x;
    v2f a = in[0], w = in[1], u = in[2], v = in[3];
    // perturb per lane so that nothing is constant-folded; lane 0 is compared on the host
    a += (float)lane; w += (float)lane; u += (float)lane; v += (float)lane;
    v2f r[6] = {cmul(a, w), cmul_conj(a, w), add_jv<+1>(u, v), add_jv<-1>(u, v), sub_j<+1>(u, v), sub_j<-1>(u, v)};
    v2f x[32];
#pragma unroll
    for (int i = 0; i < 32; i++) x[i] = in[4 + i] + (float)lane;
    v2f y16[16];
#pragma unroll
    for (int i = 0; i < 16; i++) y16[i] = x[i];
    dft_dif<32, +1>(x);
    v2f f[32];
#pragma unroll
    for (int i = 0; i < 32; i++) f[i] = x[i];
    dft_dit<32, -1>(x);
    dft_dif<16, +1>(y16);
    v2f pw[16];
    twiddle_powers<16>(w, pw);
    // small radices, both directions: [104..] dif8(+1) | dit8(-1) of it | dif4 | dit4 | dif2 | dit2 | dit16(-1) of dif16
    v2f s8[8], s4[4], s2[2];
#pragma unroll
    for (int i = 0; i < 8; i++) s8[i] = in[4 + i] + (float)lane;
#pragma unroll
    for (int i = 0; i < 4; i++) s4[i] = in[4 + i] + (float)lane;
#pragma unroll
    for (int i = 0; i < 2; i++) s2[i] = in[4 + i] + (float)lane;
    dft_dif<8, +1>(s8); dft_dif<4, +1>(s4); dft_dif<2, +1>(s2);
    v2f f8[8], f4[4], f2[2], t16[16];
#pragma unroll
    for (int i = 0; i < 8; i++) f8[i] = s8[i];
#pragma unroll
    for (int i = 0; i < 4; i++) f4[i] = s4[i];
#pragma unroll
    for (int i = 0; i < 2; i++) f2[i] = s2[i];
#pragma unroll
    for (int i = 0; i < 16; i++) t16[i] = y16[i];
    dft_dit<8, -1>(s8); dft_dit<4, -1>(s4); dft_dit<2, -1>(s2); dft_dit<16, -1>(t16);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) { out[104 + i] = f8[i]; out[112 + i] = s8[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) { out[120 + i] = f4[i]; out[124 + i] = s4[i]; }
#pragma unroll
        for (int i = 0; i < 2; i++) { out[128 + i] = f2[i]; out[130 + i] = s2[i]; }
#pragma unroll
        for (int i = 0; i < 16; i++) out[132 + i] = t16[i];
        for (int i = 0; i < 6; i++) out[i] = r[i];
#pragma unroll
        for (int i = 0; i < 32; i++) { out[8 + i] = f[i]; out[40 + i] = x[i]; }
#pragma unroll
        for (int i = 0; i < 16; i++) { out[72 + i] = y16[i]; out[88 + i] = pw[i]; }
    }
}

}  // namespace csdr

extern "C" int csdr__selftest_fft(int device, const float *h_in /*36 cpx*/, float *h_out /*148 cpx*/)
{
    using namespace csdr;
    if (!device_ok(device)) return CSDR_EHIP;
    float *d_in = nullptr, *d_out = nullptr;
    CSDR_HIP(hipMalloc((void **)&d_in, 36 * 8));
    CSDR_HIP(hipMalloc((void **)&d_out, 148 * 8));
    CSDR_HIP(hipMemcpy(d_in, h_in, 36 * 8, hipMemcpyHostToDevice));
    CSDR_HIP(hipMemset(d_out, 0, 148 * 8));
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, (const v2f *)d_in, (v2f *)d_out);
    CSDR_HIP(hipGetLastError());
    CSDR_HIP(hipMemcpy(h_out, d_out, 148 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_out);
    return CSDR_OK;
}

// ---- the clock the chip holds (bench.py: roofline.sclk_ghz) ------------------------------------------------------------
// One wave per workgroup runs a fixed chain of dependent multiply-adds between two readings of the shader-clock
// counter (s_memtime) and of the constant 100 MHz counter (s_memrealtime): cycles / time = the clock of that XCD while
// whatever else is running runs.  Launched on a side stream beside the measured kernel (MI355X_MICROARCH.md, DVFS: the
// chip lowers its clock under load, and the headline kernel sits at the socket's power cap).
namespace csdr {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, int spins)
{
    float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999999f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < spins; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = __builtin_fmaf(a, b, 1e-9f);
    }
    asm volatile("" : "+v"(a));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0 + (a == 12345.0f ? 1 : 0); }
}
}  // namespace csdr

/* internal (bench.py): nwg one-wave probes; d_out[2 i] = shader cycles, d_out[2 i + 1] = 100 MHz ticks.  stream == NULL:
 * on a stream BORROWED from the library's pool (the highest-priority one, the one the first plan group of a batch chain
 * takes) and handed straight back, so that the probes run BESIDE whatever the caller keeps launching on its own stream
 * without the process owning one stream more afterwards (a stream more costs the chain a hardware queue: stream_pool.hpp) */
extern "C" int csdr__clock_probe(int device, void *stream, unsigned long long *d_out, int nwg, int spins)
{
    using namespace csdr;
    if (!d_out || nwg < 1 || spins < 1) return fail(CSDR_EINVAL, "bad argument");
    if (!device_ok(device)) return CSDR_EHIP;
    hipStream_t s = (hipStream_t)stream;
    bool borrowed = false;
    if (!s) {
        int pr_lo = 0, pr_hi = 0;
        CSDR_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        CSDR_HIP(stream_pool().get(device, pr_hi, &s));
        borrowed = true;
    }
    hipLaunchKernelGGL(clock_probe_kernel, dim3(nwg), dim3(64), 0, s, d_out, spins);
    const hipError_t e = hipGetLastError();
    if (borrowed) stream_pool().put(device, s);
    CSDR_HIP(e);
    return CSDR_OK;
}
