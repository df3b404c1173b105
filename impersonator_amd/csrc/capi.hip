// capi.hip -- version / error / device queries of liblwg.
#include <cstring>

#include "common.h"

namespace lwg {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lwg

extern "C" {

int lwg_version(void) { return 100; /* 0.1.0 */ }

const char *lwg_last_error(void) { return lwg::g_err; }

int lwg_device_info(int *cu_count, size_t *hbm_bytes, char *name_host, size_t name_len)
{
    int dev = 0;
    LWG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    LWG_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (name_host && name_len) {
        snprintf(name_host, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return LWG_OK;
}

}  // extern "C"
