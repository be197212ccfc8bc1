// error state, version, device probe
#include "common.h"

namespace mmdgan {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static void *g_ws = nullptr;
static size_t g_ws_bytes = 0;
static bool g_prezeroed = false;
bool outputs_prezeroed() { return g_prezeroed; }
void *workspace(size_t need) { return (g_ws && need <= g_ws_bytes) ? g_ws : nullptr; }
}  // namespace mmdgan

extern "C" const char *mmdgan_last_error(void) { return mmdgan::g_err; }
extern "C" int mmdgan_version(void) { return 100; }
extern "C" int mmdgan_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
extern "C" int mmdgan_set_workspace(void *ptr, size_t bytes) {
    mmdgan::g_ws = ptr;
    mmdgan::g_ws_bytes = ptr ? bytes : 0;
    return MMDGAN_OK;
}
extern "C" int mmdgan_set_outputs_prezeroed(int on) {
    mmdgan::g_prezeroed = on != 0;
    return MMDGAN_OK;
}
