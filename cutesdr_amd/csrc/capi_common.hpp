// capi_common.hpp -- shared plumbing of the C ABI (error text, HIP checks, device scope).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/cutesdr_mi.h"

namespace csdr {

inline std::string &last_error_ref()
{
    static thread_local std::string s = "";
    return s;
}
inline int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define CSDR_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return ::csdr::fail(CSDR_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                                __FILE__, __LINE__);                                        \
    } while (0)

// true when a GPU is present and `device` is a valid ordinal; sets the error text otherwise
inline bool device_ok(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(CSDR_EHIP, "no HIP device available (%s); libcutesdr_mi has no CPU fallback",
             e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return false;
    }
    if (device < 0 || device >= n) {
        fail(CSDR_EINVAL, "device %d out of range [0,%d)", device, n);
        return false;
    }
    return hipSetDevice(device) == hipSuccess;
}

// Host staging of the drop-in classes' double* buffers ("host double* -> pinned staging -> HBM fp32"): a
// page-locked fp32 buffer the DMA engines read and write directly (a pageable source goes through the runtime's own
// bounce buffer first), grown on demand with its contents kept, and the two conversions at the boundary as plain
// loops the host compiler vectorises (cvtpd2ps / cvtps2pd).
struct PinnedBuf {
    float *p = nullptr;
    size_t cap = 0;                                      // floats
    int reserve(size_t n)
    {
        if (n <= cap) return CSDR_OK;
        float *q = nullptr;
        n = (n + 4095) / 4096 * 4096;
        if (hipHostMalloc((void **)&q, n * sizeof(float), hipHostMallocDefault) != hipSuccess)
            return fail(CSDR_ENOMEM, "hipHostMalloc(%zu) failed", n * sizeof(float));
        if (p) { memcpy(q, p, cap * sizeof(float)); (void)hipHostFree(p); }
        p = q; cap = n;
        return CSDR_OK;
    }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
};
inline void cvt_to_f32(float *__restrict dst, const double *__restrict src, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = (float)src[i]; }
inline void cvt_to_f64(double *__restrict dst, const float *__restrict src, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = (double)src[i]; }

}  // namespace csdr
