// capi.hip -- library-level entry points of librqhip.so (version, error string, device info).
#include <stdarg.h>

#include "rqhip_common.h"

namespace rqhip {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int cu_count() {
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cached = n;
    return n;
}

}  // namespace rqhip

extern "C" int rqhip_version(void) { return RQHIP_VERSION; }

extern "C" const char *rqhip_last_error(void) { return rqhip::g_err; }

extern "C" int rqhip_device_cu_count(int *out) {
    if (!out) {
        rqhip::set_error("rqhip_device_cu_count: null output pointer");
        return RQHIP_EARG;
    }
    int dev = 0, n = 0;
    RQ_RETURN_IF_HIP(hipGetDevice(&dev));
    RQ_RETURN_IF_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    *out = n;
    return RQHIP_OK;
}
