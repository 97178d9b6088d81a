// fastfir_dev.hpp -- device helpers shared by the overlap-save kernels (fastfir_kernels.hip,
// fastfir2_kernels.hip): buffer (SRSRC) addressing, wide-store groups, the per-size constants.
#pragma once
#include "fft_core.hpp"

namespace csdr {

// ---- buffer (SRSRC) addressing: one wave-uniform descriptor per stream, a 32-bit per-lane byte
//      offset and a scalar offset per access -- keeps 64-bit address pairs out of the VGPR file
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ v4f buf_load16(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store16(rsrc_t r, int voff, int soff, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, 0);
}
// streaming forms: the samples are read once and written once (cache-policy bits of the buffer instruction:
// 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__device__ __forceinline__ v4f buf_load16_aux(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_store16_aux(rsrc_t r, int voff, int soff, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, AUX);
}

// Wide stores (buffer_store_dwordx4 with an SGPR soffset, ds_write_b128) read their data VGPRs
// over several cycles after issue.  hipcc (ROCm 7.2) pads the ">64-bit store data overwritten by
// the next VALU" hazard only for the soffset-less form, and reuses one 128-bit tuple for
// consecutive stores (v_mov into it right behind the previous store): on gfx950 that corrupted
// the second dword in lanes 12-15 of every 16, intermittently (tools/debug_fastfir5.py).  So
// every group of wide stores first materialises all its 128-bit operands in distinct registers
// (store_operand), is fenced from the scheduler, and ends with two wait states.
#define CSDR_STORE_GROUP_BEGIN() __builtin_amdgcn_sched_barrier(0)
#define CSDR_STORE_GROUP_END()            \
    do {                                  \
        asm volatile("s_nop 1");          \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

__device__ __forceinline__ v4f store_operand(v2f lo, v2f hi)
{
    v4f v = {lo.x, lo.y, hi.x, hi.y};
    asm volatile("" : "+v"(v));         // force the tuple to exist before the fenced store group
    return v;
}

template <int LOG2N>
struct FastFirCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 32;          // threads per workgroup
    static constexpr int R0 = N / 1024;       // radix of the outer pass
    static constexpr int G = 32 / R0;         // adjacent columns handled per thread in F1/I3
    static constexpr int LDS_DATA = N + 2 * (N / 32);               // padded v2f elements
    static constexpr int LDS_BYTES = (LDS_DATA + 1024) * 8;         // + 32x32 twiddle table
};

}  // namespace csdr
