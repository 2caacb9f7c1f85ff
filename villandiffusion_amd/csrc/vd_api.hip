// Library-level entry points of the C ABI (include/villan_hip.h): version, error string, device probe.
#include <stdarg.h>
#include <string.h>
#include "vd_common.h"

static thread_local char g_err[512] = "";

void vd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vd_abi_version(void) { return VD_ABI_VERSION; }

extern "C" const char* vd_last_error(void) { return g_err; }

extern "C" int vd_device_ok(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        vd_set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return e == hipSuccess ? VD_EINVAL : (int)e;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        vd_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        vd_set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", dev, p.gcnArchName);
        return VD_EINVAL;
    }
    return 0;
}
