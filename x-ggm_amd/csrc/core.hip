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

// ---- prefetch queue (common.h: PrefetchArgs)
static PrefetchArgs g_prefetch = {{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}, 0, 0, nullptr};
static int g_prefetch_on = -1;

extern "C" int xggm_prefetch_next(const void* ptr, size_t bytes) {
    if (g_prefetch_on < 0) {
        const char* e = getenv("XGGM_PREFETCH");
        g_prefetch_on = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!g_prefetch_on || !ptr || bytes < 16) return XGGM_OK;
    XGGM_REQUIRE(reinterpret_cast<uintptr_t>(ptr) % 16 == 0, "xggm_prefetch_next: the range must start on a 16-byte border");
    if (g_prefetch.k < 4) {
        g_prefetch.p[g_prefetch.k] = ptr;
        g_prefetch.n[g_prefetch.k] = bytes;
        ++g_prefetch.k;
    }  // a fifth range is dropped: the queue belongs to the next launch, which has room for four
    return XGGM_OK;
}

PrefetchArgs xggm_take_prefetch() {
    PrefetchArgs a = g_prefetch;
    unsigned long long total = 0;
    for (int i = 0; i < a.k; ++i) total += a.n[i];
    // ~48 KB per workgroup, at most one workgroup per CU
    a.blocks = a.k ? (int)std::min<unsigned long long>(256, (total + 49151) / 49152) : 0;
    g_prefetch.k = 0;
    return a;
}
