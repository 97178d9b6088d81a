// valu_rate.hip -- issue rate of packed fp32 VALU on gfx950 by waves per SIMD (what bounds K1's butterflies).
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float *out, int iters, float seed)
{
    v2f a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = v2f{seed + i + threadIdx.x, seed - i};
    v2f w = {0.999f, 0.001f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(w));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
                if (OP == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(w.x), "v"(w.y));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(w), "v"(w));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char *name, float *d)
{
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(256), block(64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps);
        int blocks_per_cu = (64 * 4 * wps + 1023) / 1024;
        grid.x = 256 * blocks_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<grid, block>>>(d, iters, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) k<OP><<<grid, block>>>(d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double instr_per_wave = (double)iters * 32;
        // cycles per wave-instruction per SIMD at 2.4 GHz nominal
        double ns_per_instr_simd = ms * 1e6 / (instr_per_wave * wps);
        printf("%-12s waves/SIMD %d: %.3f ms, %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, wps, ms,
               ns_per_instr_simd, ns_per_instr_simd * 2.4);
    }
}

int main()
{
    float *d; hipMalloc(&d, 1024);
    run<0>("pk_fma", d); run<4>("pk_fma_opsel", d); run<1>("pk_add", d); run<2>("pk_mul", d); run<3>("fma", d);
    return 0;
}
