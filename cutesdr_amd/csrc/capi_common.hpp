// capi_common.hpp -- shared plumbing of the C ABI (error text, HIP checks, device scope).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
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

}  // namespace csdr
