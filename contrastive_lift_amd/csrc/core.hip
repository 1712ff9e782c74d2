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

extern "C" int clift_version(void) { return 11; }

// Data-parallel runs: while an asynchronous RCCL all-reduce is in flight the persistent launches (one block per CU, held for the whole
// launch) leave `k` CUs to the collective's kernels.  Host state of the calling process; takes effect at the next launch.
static int g_cu_reserve = 0;
int clift_persistent_cus() { return 256 - g_cu_reserve; }
extern "C" int clift_set_cu_reserve(int k) {
    if (k < 0 || k > 128) { clift_set_error("clift_set_cu_reserve: k must be in [0,128] (got %d)", k); return 1; }
    g_cu_reserve = k;
    return 0;
}

// per-translation-unit binders (CLIFT_ROWS_LIMIT_BINDER in each kernel file)
void clift_bind_rows_limit_march(const int* p);
void clift_bind_rows_limit_heads_io(const int* p);
void clift_bind_rows_limit_gemm(const int* p);
void clift_bind_rows_limit_layer_f32(const int* p);
void clift_bind_rows_limit_layer_n128(const int* p);
void clift_bind_rows_limit_narrow_stream(const int* p);
void clift_bind_rows_limit_layer_x6(const int* p);

extern "C" int clift_bind_rows_limit(const int* dev_limit) {
    clift_bind_rows_limit_march(dev_limit);
    clift_bind_rows_limit_heads_io(dev_limit);
    clift_bind_rows_limit_gemm(dev_limit);
    clift_bind_rows_limit_layer_f32(dev_limit);
    clift_bind_rows_limit_layer_n128(dev_limit);
    clift_bind_rows_limit_narrow_stream(dev_limit);
    clift_bind_rows_limit_layer_x6(dev_limit);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { clift_set_error("clift_bind_rows_limit: %s", hipGetErrorString(e)); return 2; }
    return 0;
}
extern "C" const char* clift_last_error(void) { return g_err; }
