// core.hip -- ABI version and error plumbing of libclift.so.
#include "clift_dev.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void clift_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int clift_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        clift_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

extern "C" int clift_version(void) { return 6; }
extern "C" const char* clift_last_error(void) { return g_err; }
