// stream_read.hip -- what a streaming READ with K2's access shape reaches on MI355X: one wave per workgroup,
// a segment of a channel row walked in tiles of TILE samples (16 B per lane and row of 128 samples), DEPTH tiles
// prefetched in registers, WAVES waves per SIMD; 256 rows x 2^21 complex fp32 = 4.3 GB read, almost nothing written.
// build: hipcc -O3 --offload-arch=gfx950 -o stream_read stream_read.hip ; run: ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ROWS, int DEPTH, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k(const v4f *in, long stride_pairs, int nseg, long seg_pairs, float *out, int work)
{
    const int ch = blockIdx.x / nseg, seg = blockIdx.x % nseg, t = threadIdx.x;
    const v4f *p = in + ch * stride_pairs + seg * seg_pairs + t;
    const long ntiles = seg_pairs / (64 * ROWS);
    v4f buf[DEPTH][ROWS];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int r = 0; r < ROWS; r++) buf[d][r] = p[(d * ROWS + r) * 64];
    v4f acc = {0, 0, 0, 0};
    for (long k = 0; k < ntiles; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            v4f cur[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) cur[r] = buf[d][r];
            const long nx = k + d + DEPTH;
            if (nx < ntiles) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) buf[d][r] = p[(nx * ROWS + r) * 64];
            }
#pragma unroll
            for (int r = 0; r < ROWS; r++) acc += cur[r];
            // stand-in for the cascade: `work` dependent VALU steps per tile
            for (int w = 0; w < work; w++) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(reinterpret_cast<double *>(&acc))));
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int ROWS, int DEPTH, int WAVES>
void run(const v4f *d_in, float *d_out, int work)
{
    const int C = 256; const long T = 1L << 21;          // samples per row
    const long pairs = T / 2;
    const int nseg = 32;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k<ROWS, DEPTH, WAVES>), dim3(C * nseg), dim3(64), 0, 0, d_in, pairs, nseg, pairs / nseg, d_out, work);
    hipEventRecord(e0);
    const int reps = 10;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k<ROWS, DEPTH, WAVES>), dim3(C * nseg), dim3(64), 0, 0, d_in, pairs, nseg, pairs / nseg, d_out, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("rows/tile %d (tile %d samples) depth %d waves/SIMD %d work %3d: %.3f ms  %.2f TB/s\n", ROWS, ROWS * 128, DEPTH, WAVES, work, ms, C * T * 8.0 / ms / 1e9);
}

int main()
{
    const size_t bytes = 256ul * (1ul << 21) * 8;
    v4f *d_in; float *d_out;
    hipMalloc(&d_in, bytes); hipMalloc(&d_out, 64);
    hipMemset(d_in, 0, bytes);
    for (int work : {0, 100, 400}) {
        run<4, 1, 4>(d_in, d_out, work);
        run<4, 2, 4>(d_in, d_out, work);
        run<4, 4, 4>(d_in, d_out, work);
        run<4, 1, 8>(d_in, d_out, work);
        run<4, 2, 8>(d_in, d_out, work);
        run<8, 1, 4>(d_in, d_out, work);
        run<8, 2, 2>(d_in, d_out, work);
        run<2, 4, 8>(d_in, d_out, work);
    }
    return 0;
}
