// downconv_kernels.hip -- launch side of the NCO + decimator cascade (K2): the LDS layout, the run-time-plan
// instantiation of the kernel and the table of the precompiled plans (downconv_kernel.hpp holds the device code).
#include "launch_once.hpp"
#include "downconv_kernel.hpp"

namespace csdr {

int downconv_layout(DcArgs &a)
{
    const DcLayout l = dc_layout_of(a.kind, a.nstages);
    for (int s = 0; s <= a.nstages; s++) { a.roff[s] = l.roff[s]; a.ooff[s] = l.ooff[s]; }
    return l.slots * 8 + 64;
}

// the precompiled plans: one launch function per DC_PLAN(id, kinds...) line of the generated table
struct DcCompiledPlan {
    int ns;
    int kind[DC_MAX_STAGES];
    hipError_t (*launch)(DcArgs &, hipStream_t);
};
#if __has_include("downconv_plans.inc")
#define DC_PLAN(id, ...) hipError_t downconv_launch_plan_##id(DcArgs &, hipStream_t);
#include "downconv_plans.inc"
#undef DC_PLAN
template <int... K> static constexpr int dc_count_kinds() { return sizeof...(K); }
static const DcCompiledPlan dc_compiled[] = {
#define DC_PLAN(id, ...) {dc_count_kinds<__VA_ARGS__>(), {__VA_ARGS__}, &downconv_launch_plan_##id},
#include "downconv_plans.inc"
#undef DC_PLAN
    {-1, {0}, nullptr}};
#else
static const DcCompiledPlan dc_compiled[] = {{-1, {0}, nullptr}};
#endif

static int dc_force_dynamic = 0;
/* tests: 1 = every launch takes the run-time-plan kernel; returns the number of precompiled plans */
int downconv_force_dynamic(int on)
{
    if (on >= 0) dc_force_dynamic = on;
    int n = 0;
    while (dc_compiled[n].launch) n++;
    return n;
}

hipError_t downconv_launch(DcArgs &a, hipStream_t stream)
{
    const int lds = downconv_layout(a);
#ifdef CSDR_WG_TRACE
    a.trace = wgtrace_next();
#endif
    if (!dc_force_dynamic)
        for (const DcCompiledPlan *p = dc_compiled; p->launch; p++) {
            if (p->ns != a.nstages) continue;
            bool same = true;
            for (int s = 0; s < p->ns; s++) same = same && p->kind[s] == a.kind[s];
            if (same) return p->launch(a, stream);
        }
    // (once per device: launch_once.hpp; never needed in practice -- a cascade is ~10 KB)
    if (a.nb_mask) {
        if (lds > 64 * 1024) {
            hipError_t e = CSDR_MAX_LDS_ONCE((&downconv_kernel<DcPlanDyn, true>), lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((downconv_kernel<DcPlanDyn, true>), dim3(a.nchan * a.nseg), dim3(DC_T), lds, stream, a);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) {
        hipError_t e = CSDR_MAX_LDS_ONCE((&downconv_kernel<DcPlanDyn>), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(downconv_kernel<DcPlanDyn>, dim3(a.nchan * a.nseg), dim3(DC_T), lds, stream, a);
    return hipGetLastError();
}

}  // namespace csdr
