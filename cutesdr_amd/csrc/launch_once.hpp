// launch_once.hpp -- hipFuncAttributeMaxDynamicSharedMemorySize ONCE per device and kernel.  The attribute belongs to the
// device (a process may drive several), so the launch functions used to set it before every launch: host-side
// microseconds on the per-datagram host form.  `done` is a static of the launch site, one bit per
// device ordinal.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace csdr {

inline hipError_t max_dynamic_lds_once(std::atomic<unsigned long long> &done, const void *kernel, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && (done.load(std::memory_order_acquire) & (1ull << dev))) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}
#define CSDR_MAX_LDS_ONCE(kernel, bytes)                                                                  \
    [&]() -> hipError_t {                                                                                 \
        static std::atomic<unsigned long long> done_{0};                                                  \
        return ::csdr::max_dynamic_lds_once(done_, reinterpret_cast<const void *>(kernel), (int)(bytes)); \
    }()

}  // namespace csdr
