// patch_kernels.hip -- the kernel behind patch_queue.hpp: one workgroup per patch copies its words from the pinned arena
// (read over PCIe in the kernel's own loads) to their place in device memory, or fills a range with one word.
#include "patch_queue.hpp"

namespace csdr {

__global__ __launch_bounds__(256) void patch_apply_kernel(const PatchDesc *list)
{
    const PatchDesc d = list[blockIdx.x];
    unsigned *dst = reinterpret_cast<unsigned *>(d.dst);
    const unsigned n = d.bytes >> 2;
    if (d.fill) {
        const unsigned w = (unsigned)d.src;
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = w;
    } else {
        const unsigned *src = reinterpret_cast<const unsigned *>(d.src);
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

hipError_t patch_apply_launch(const PatchDesc *d_list, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(patch_apply_kernel, dim3(n), dim3(256), 0, s, d_list);
    return hipGetLastError();
}

}  // namespace csdr
