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

extern "C" int clift_version(void) { return 18; }

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
void clift_bind_rows_limit_layer_bf16(const int* p);
void clift_bind_rows_limit_layer_x6w(const int* p);
void clift_bind_rows_limit_layer_nb16(const int* p);
void clift_bind_rows_limit_layer_n6(const int* p);

extern "C" int clift_bind_rows_limit(const int* dev_limit) {
    clift_bind_rows_limit_march(dev_limit);
    clift_bind_rows_limit_heads_io(dev_limit);
    clift_bind_rows_limit_gemm(dev_limit);
    clift_bind_rows_limit_layer_f32(dev_limit);
    clift_bind_rows_limit_layer_n128(dev_limit);
    clift_bind_rows_limit_narrow_stream(dev_limit);
    clift_bind_rows_limit_layer_x6(dev_limit);
    clift_bind_rows_limit_layer_bf16(dev_limit);
    clift_bind_rows_limit_layer_x6w(dev_limit);
    clift_bind_rows_limit_layer_nb16(dev_limit);
    clift_bind_rows_limit_layer_n6(dev_limit);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { clift_set_error("clift_bind_rows_limit: %s", hipGetErrorString(e)); return 2; }
    return 0;
}
// ---- XCD-private gradient shards (clift_dev.h): binder, pass bracket
void clift_bind_grad_shards_march(const void* p);
void clift_bind_grad_shards_heads_io(const void* p);
void clift_bind_grad_shards_gemm(const void* p);
void clift_bind_grad_shards_layer_f32(const void* p);
void clift_bind_grad_shards_layer_n128(const void* p);
void clift_bind_grad_shards_narrow_stream(const void* p);
void clift_bind_grad_shards_layer_x6(const void* p);
void clift_bind_grad_shards_layer_bf16(const void* p);
void clift_bind_grad_shards_layer_x6w(const void* p);
void clift_bind_grad_shards_layer_nb16(const void* p);
void clift_bind_grad_shards_layer_n6(const void* p);

extern "C" int clift_bind_grad_shards(const void* dev_desc) {
    clift_bind_grad_shards_march(dev_desc);
    clift_bind_grad_shards_heads_io(dev_desc);
    clift_bind_grad_shards_gemm(dev_desc);
    clift_bind_grad_shards_layer_f32(dev_desc);
    clift_bind_grad_shards_layer_n128(dev_desc);
    clift_bind_grad_shards_narrow_stream(dev_desc);
    clift_bind_grad_shards_layer_x6(dev_desc);
    clift_bind_grad_shards_layer_bf16(dev_desc);
    clift_bind_grad_shards_layer_x6w(dev_desc);
    clift_bind_grad_shards_layer_nb16(dev_desc);
    clift_bind_grad_shards_layer_n6(dev_desc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { clift_set_error("clift_bind_grad_shards: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

// begin: the bound record := *src (enabled = 1), and `zero_n` floats at `zero` are cleared (the pass's gradient range: this launch takes the
// place of the fill the caller would issue anyway)
__global__ __launch_bounds__(256) void k_grad_shards_begin(GradShardDesc* desc, const GradShardDesc* src, float4* zero4, long n4, float* zero_tail, int ntail) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0) { GradShardDesc d = *src; d.enabled = 1; *desc = d; }
    for (long i = gid; i < n4; i += (long)gridDim.x * blockDim.x) zero4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gid < ntail) zero_tail[gid] = 0.f;
}
// fold: gradients[i] += sum over the eight shards, shards cleared for the next pass, record disabled.  n floats starting `first` floats into
// the range (a pass folds the part of the range its kernels wrote).
__global__ __launch_bounds__(256) void k_grad_shards_fold(GradShardDesc* desc, long first, long n, int disable) {
    const GradShardDesc d = *desc;        // (addresses only: `enabled` may be changing under us)
    float* dst = reinterpret_cast<float*>((uintptr_t)d.lo) + first;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = gid; i < n; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float* sp = reinterpret_cast<float*>((uintptr_t)(d.shard0 + x * d.stride)) + first + i;
            acc += *sp;
            *sp = 0.f;
        }
        dst[i] += acc;
    }
    if (disable && gid == 0) desc->enabled = 0;
}
__global__ void k_grad_shards_disable(GradShardDesc* desc) { desc->enabled = 0; }

extern "C" int clift_grad_shards_begin(void* dev_desc, const void* dev_src, float* zero, long zero_n, clift_stream_t s) {
    CLIFT_REQUIRE(dev_desc != nullptr && dev_src != nullptr, "clift_grad_shards_begin: descriptor pointers are required");
    CLIFT_REQUIRE(zero_n == 0 || (zero != nullptr && (((uintptr_t)zero) & 15) == 0), "clift_grad_shards_begin: the range to clear must be 16-byte aligned");
    const long n4 = zero_n / 4;
    const int ntail = (int)(zero_n - 4 * n4);
    long blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    k_grad_shards_begin<<<(int)blocks, 256, 0, as_stream(s)>>>(static_cast<GradShardDesc*>(dev_desc), static_cast<const GradShardDesc*>(dev_src),
                                                               reinterpret_cast<float4*>(zero), n4, zero + 4 * n4, ntail);
    return clift_check_launch("clift_grad_shards_begin");
}
extern "C" int clift_grad_shards_fold(void* dev_desc, long first, long n, int disable, clift_stream_t s) {
    CLIFT_REQUIRE(dev_desc != nullptr && first >= 0 && n >= 0, "clift_grad_shards_fold: bad arguments");
    if (n > 0) {
        long blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        k_grad_shards_fold<<<(int)blocks, 256, 0, as_stream(s)>>>(static_cast<GradShardDesc*>(dev_desc), first, n, disable);
    } else if (disable) {
        k_grad_shards_disable<<<1, 1, 0, as_stream(s)>>>(static_cast<GradShardDesc*>(dev_desc));
    }
    return clift_check_launch("clift_grad_shards_fold");
}

extern "C" const char* clift_last_error(void) { return g_err; }
