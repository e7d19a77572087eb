// error reporting + version for libxggm_hip.so
#include "common.h"
#include "xggm.h"

static thread_local char g_err[512] = "";

void xggm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int xggm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        xggm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return XGGM_ERR_LAUNCH;
    }
    return XGGM_OK;
}

extern "C" int xggm_version(void) { return XGGM_VERSION; }
extern "C" const char* xggm_last_error(void) { return g_err; }
