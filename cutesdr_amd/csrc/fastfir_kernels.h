// fastfir_kernels.h -- launch interface of the batched overlap-save kernel (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "wg_trace.hpp"

namespace csdr {

typedef float v2f_h __attribute__((ext_vector_type(2)));
typedef float v4f_h __attribute__((ext_vector_type(4)));

struct FastFirArgs {
    const v2f_h *in;      // [channels][in_stride] complex fp32, this call's new samples
    const v2f_h *hist;    // [channels][N/2]: last N/2 samples of the previous call (zeros at start)
    v2f_h *hist_next;     // [channels][N/2]: receives this call's tail (other ping-pong half)
    v2f_h *out;           // [channels][out_stride]
    const v4f_h *h;       // frequency response in pass-F3 register order, [16][N/32] float4 per filter
    const v2f_h *tw1;     // W_N^{n}, n = 0..1023
    const v2f_h *tw2;     // W_1024^{n*k}, [k][n], 32x32
    long in_stride;       // complex samples between channels
    long out_stride;
    long h_stride;        // float4 between channel filters (0 = one shared filter)
    int channels;
    int nblocks;          // hops of N/2 samples per channel in this call
    int blocks_per_run;   // consecutive blocks walked by one workgroup
    int runs;             // ceil(nblocks / blocks_per_run)
    int dbg_stage;        // 0 in production; >0 selects the diagnostic twin kernel
    v2f_h *dbg;           // [N] LDS image dump of the diagnostic twin
#ifdef CSDR_WG_TRACE
    WgTraceArg trace;
#endif
};

hipError_t fastfir_launch(int log2n, const FastFirArgs &a, hipStream_t stream);
int fastfir_bin_of(int log2n, int t, int r);

// software-pipelined build, N = 2048 ... 16384 (fastfir2_kernels.hip): same LDS image as fastfir_launch, its
// own H order
hipError_t fastfir2_launch(int log2n, const FastFirArgs &a, hipStream_t stream);
// natural-order spectrum bin of H slot (float4 index j*(N/32) + t, half e) of that kernel
int fastfir2_bin_of(int log2n, int t, int j, int e);

}  // namespace csdr
