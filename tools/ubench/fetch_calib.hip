// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for a KNOWN number of bytes read, by load width: the
// guide (MI355X_MICROARCH.md, HBM) calibrates it only for 16-byte-per-lane streaming reads (it counts half of
// those bytes on gfx950) and asks for a calibration in one's own access pattern before trusting an absolute.  The
// blanker and the post-chain read 8 bytes per lane, the S-meter / audio paths 4: each kernel below streams 1 GiB
// once, coalesced, with one load width.
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib   (tools/fetch_calib.sh condenses it)
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ void rd(const T *in, long n, float *out)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const T v = in[i];
        const float *f = reinterpret_cast<const float *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += f[k];
    }
    if (acc == 12345.678f) out[0] = acc;
}
// the blanker's shape: three 8-byte streams per sample, two of them offsets behind the first
__global__ void rd3(const float2 *in, long n, long back_far, long back_near, float *out)
{
    float acc = 0.f;
    for (long i = back_far + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float2 a = in[i], b = in[i - back_far], c = in[i - back_near];
        acc += a.x + b.y + c.x;
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main()
{
    const long bytes = 1l << 30;
    void *d; float *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 64);
    hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 3; rep++) {
        rd<float><<<4096, 256>>>((const float *)d, bytes / 4, o);
        rd<float2><<<4096, 256>>>((const float2 *)d, bytes / 8, o);
        rd<float4><<<4096, 256>>>((const float4 *)d, bytes / 16, o);
        rd3<<<4096, 256>>>((const float2 *)d, bytes / 8, 10001, 20, o);
    }
    hipDeviceSynchronize();
    printf("read %ld bytes per launch (rd3: the same bytes through three 8-byte streams)\n", bytes);
    return 0;
}
