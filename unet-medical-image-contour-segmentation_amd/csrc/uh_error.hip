// Error text plumbing for the C ABI (thread-local; see include/unet_hip.h).
#include "uh_common.h"

static thread_local char g_err[512] = "";

void uh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uh_last_error(void) { return g_err; }
extern "C" int uh_version(void) { return 100; }
