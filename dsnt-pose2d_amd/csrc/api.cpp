// Error reporting and version for libdsnt_hip.so.  No other global state lives in the library.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int dsnt_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int dsnt_version(void) { return 100; }
extern "C" const char* dsnt_last_error(void) { return g_err; }
