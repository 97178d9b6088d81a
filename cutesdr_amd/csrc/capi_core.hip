// capi_core.hip -- version, error text and device-memory helpers of the C ABI.
#include "capi_common.hpp"
#include <cstdlib>
#include "wg_trace.hpp"
#include "ref_constants.hpp"

using namespace csdr;

extern "C" {

int csdr_version(void) { return 100; }
const char *csdr_last_error(void) { return last_error_ref().c_str(); }

int csdr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void *csdr_dev_alloc(int device, unsigned long long bytes)
{
    if (!device_ok(device)) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        fail(CSDR_ENOMEM, "hipMalloc(%llu) failed", bytes);
        return nullptr;
    }
    return p;
}
int csdr_dev_free(int device, void *p)
{
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(hipFree(p));
    return CSDR_OK;
}
int csdr_dev_upload(int device, void *dst, const void *src, unsigned long long bytes)
{
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return CSDR_OK;
}
int csdr_dev_download(int device, void *dst, const void *src, unsigned long long bytes)
{
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return CSDR_OK;
}
#ifdef CSDR_WG_TRACE
/* diagnostic builds only (tools/wg_trace.py): the device buffer the chain's kernels leave their per-workgroup records
 * in (wg_trace.hpp; words [0] = 0 and [1] = capacity set by the caller), or NULL to stop tracing */
int csdr__wgtrace_set(void *buf)
{
    wgtrace_host_buf() = (unsigned long long *)buf;
    wgtrace_host_launch() = 0;
    return CSDR_OK;
}
#endif
/* internal (tests/test_reference_constants.py): the reference's named constants as this library's arithmetic uses them
 * (ref_constants.hpp) -- fills up to cap entries, returns how many there are */
int csdr__constants(const char **names, double *values, int cap)
{
    for (int i = 0; i < refc::TABLE_N && i < cap; i++) { names[i] = refc::TABLE[i].name; values[i] = refc::TABLE[i].value; }
    return refc::TABLE_N;
}
int csdr_dev_sync(int device)
{
    if (!device_ok(device)) return CSDR_EHIP;
    CSDR_HIP(hipDeviceSynchronize());
    return CSDR_OK;
}

}  // extern "C"
