// downconv_plan.hip -- the down-converter kernel (downconv_kernel.hpp) instantiated for ONE compile-time plan.
// cutesdr_amd/_build.py compiles this file once per entry of its DC_PLANS list, with
//   -DDC_PLAN_ID=<n> -DDC_PLAN_KINDS=<stage kinds, comma separated>
// (tools/list_dc_plans.py derives the list from the reference's radio rates x demodulator bandwidths), and writes
// the matching table downconv_plans.inc for downconv_kernels.hip.  Compiled without those macros (the build's glob
// over *.hip does that once) it is empty.
#ifdef DC_PLAN_ID
#include "launch_once.hpp"
#include "downconv_kernel.hpp"

namespace csdr {

#define DC_CAT2(a, b) a##b
#define DC_CAT(a, b) DC_CAT2(a, b)

hipError_t DC_CAT(downconv_launch_plan_, DC_PLAN_ID)(DcArgs &a, hipStream_t stream)
{
    using P = DcPlanT<DC_PLAN_KINDS>;
    const int lds = dc_layout_of(P::KIND, P::NS).slots * 8 + 64;
    // (once per device: launch_once.hpp; never needed in practice -- a cascade is ~10 KB)
    if (a.nb_mask) {                                    // the blanker's mask applied in the kernel's own loads
        if (lds > 64 * 1024) {
            hipError_t e = CSDR_MAX_LDS_ONCE((&downconv_kernel<P, true>), lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((downconv_kernel<P, true>), dim3(a.nchan * a.nseg), dim3(DC_T), lds, stream, a);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) {
        hipError_t e = CSDR_MAX_LDS_ONCE((&downconv_kernel<P>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(downconv_kernel<P>, dim3(a.nchan * a.nseg), dim3(DC_T), lds, stream, a);
    return hipGetLastError();
}

}  // namespace csdr
#endif
