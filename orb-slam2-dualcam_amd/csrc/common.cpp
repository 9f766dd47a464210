// common.cpp -- error text, device probing, version.
#include "common.h"

#include <cstring>

namespace dcs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s); libdcs_hip has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return DCS_ERR_NO_DEVICE;
    }
    return DCS_OK;
}

}  // namespace dcs

extern "C" {

const char* dcs_last_error(void) { return dcs::g_err; }
const char* dcs_version(void) { return "dcs-hip 0.1 (gfx950)"; }
int dcs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // extern "C"
