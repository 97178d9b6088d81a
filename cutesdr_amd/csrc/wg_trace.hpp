// wg_trace.hpp -- DIAGNOSTIC BUILDS ONLY (-DCSDR_WG_TRACE, tools/wg_trace.py): every workgroup of the chain's kernels
// leaves one record {kernel tag | launch number | workgroup, start, end, where} in a device buffer, so that a step of the
// batch chain can be drawn workgroup by workgroup -- when a workgroup of one kernel really became resident beside the
// others, on which XCD / CU, and for how long.  (rocprofv3's kernel trace gives one begin / end per LAUNCH: a
// workgroup that waits for LDS or registers on a busy CU is invisible there.)  The production library is compiled
// without the macro: no field, no instruction of this is in it.
#pragma once
#ifdef CSDR_WG_TRACE
#include <hip/hip_runtime.h>

namespace csdr {

// buffer layout (unsigned long long words): [0] next record (atomic), [1] capacity in records, [2..3] unused,
// then records of 4 words: tag, start, end (100 MHz s_memrealtime ticks), where (HW_ID | XCC_ID << 32)
enum { WGT_DC = 1, WGT_FF = 2, WGT_SMETER = 3, WGT_PEAKS = 4, WGT_WALK = 5, WGT_SQ_MAPS = 6, WGT_SQ_DECIDE = 7, WGT_SQ_APPLY = 8 };

inline unsigned long long *&wgtrace_host_buf() { static unsigned long long *p = nullptr; return p; }
inline unsigned &wgtrace_host_launch() { static unsigned n = 0; return n; }
// what a launch function puts into its kernel's argument block: the buffer and this launch's number
struct WgTraceArg { unsigned long long *buf; unsigned launch; unsigned pad; };
inline WgTraceArg wgtrace_next() { return WgTraceArg{wgtrace_host_buf(), wgtrace_host_buf() ? ++wgtrace_host_launch() : 0u, 0u}; }

struct WgTraceScope {
    unsigned long long *buf, t0;
    unsigned kind, launch;
    __device__ __forceinline__ WgTraceScope(WgTraceArg a, unsigned k) : buf(a.buf), t0(0), kind(k), launch(a.launch)
    {
        if (buf) t0 = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ ~WgTraceScope()
    {
        if (!buf || threadIdx.x != 0) return;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned long long slot = atomicAdd(buf, 1ull);
        if (slot < buf[1]) {
            unsigned long long *r = buf + 4 + 4 * slot;
            r[0] = ((unsigned long long)kind << 56) | ((unsigned long long)launch << 32) | blockIdx.x;
            r[1] = t0; r[2] = t1; r[3] = hw | ((unsigned long long)xcc << 32);
        }
    }
};
#define CSDR_WG_TRACE_SCOPE(arg, kind) ::csdr::WgTraceScope wg_trace_scope_((arg), (kind))

}  // namespace csdr
#else
#define CSDR_WG_TRACE_SCOPE(arg, kind) do { } while (0)
#endif
