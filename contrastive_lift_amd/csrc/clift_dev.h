// clift_dev.h -- shared device helpers for libclift.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/clift.h"

#define CLIFT_WAVE 64

// ----------------------------------------------------------------------------- host-side error plumbing
void clift_set_error(const char* fmt, ...);
int clift_check_launch(const char* what);
int clift_persistent_cus();      // blocks of a one-block-per-CU persistent launch: 256 - clift_set_cu_reserve()

#define CLIFT_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            clift_set_error(__VA_ARGS__);     \
            return 1;                         \
        }                                     \
    } while (0)

static inline hipStream_t as_stream(clift_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ----------------------------------------------------------------------------- kernel-argument mirrors
struct MarchP {
    float lo[3], hi[3], inv2[3];
    float step;
    int S;
    float dist_scale, shift, thres;
};
struct VmP {
    const float* plane[3];
    const float* line[3];
    int res[3];
    int comps;
};
struct VmG {
    float* plane[3];
    float* line[3];
    long xcd_stride;  // floats between the 8 per-XCD accumulation copies; 0 = single copy, device-scope atomics
};

static inline MarchP to_dev(const clift_march_t* m) {
    MarchP p;
    for (int i = 0; i < 3; ++i) { p.lo[i] = m->lo[i]; p.hi[i] = m->hi[i]; p.inv2[i] = m->inv_ext2[i]; }
    p.step = m->step_size; p.S = m->n_samples; p.dist_scale = m->distance_scale;
    p.shift = m->density_shift; p.thres = m->weight_thres;
    return p;
}
static inline VmP to_dev(const clift_vm_t* v) {
    VmP p;
    for (int i = 0; i < 3; ++i) { p.plane[i] = v->plane[i]; p.line[i] = v->line[i]; p.res[i] = v->res[i]; }
    p.comps = v->comps;
    return p;
}
static inline VmG to_dev(const clift_vm_grad_t* g) {
    VmG p;
    for (int i = 0; i < 3; ++i) { p.plane[i] = g->plane[i]; p.line[i] = g->line[i]; }
    p.xcd_stride = g->xcd_stride;
    return p;
}

// ----------------------------------------------------------------------------- dynamic row limit
// A training step that never reads the active-sample count back to the host (no synchronisation, capturable in a hipGraph) sizes
// its buffers and its grids by a CAPACITY and lets every per-sample kernel clamp its row count to the true count, which lives in
// device memory: clift_bind_rows_limit(ptr) publishes the address once (one copy of the pointer per translation unit -- device
// globals are per object file), the caller keeps INT_MAX there except between clift_scan_counts(..., limit = ptr) and the end of the pass.
static __device__ const int* g_rows_limit = nullptr;
__device__ __forceinline__ int limit_rows(int M) {
    const int* p = g_rows_limit;
    return p ? min(M, *p) : M;
}
__device__ __forceinline__ bool rows_limited() { return g_rows_limit != nullptr; }
__device__ __forceinline__ bool rows_cut(long row) {          // is this row past the true count?
    const int* p = g_rows_limit;
    return p && row >= (long)*p;
}
// ----------------------------------------------------------------------------- XCD-private gradient shards
// The kernels that END by adding a per-block partial of a small gradient tensor to memory (weight gradients, the K = 3 / output layers'
// gradients) pay for contention, not for bandwidth: 256 blocks of all eight XCDs add to the same few cache lines at the same moment, and a
// line that is read-modify-written from several XCDs' L2s bounces between them -- 8 - 20 us per launch at every size (tools/fixed_cost_probe.py,
// DESIGN.md 5c).  With shards, a block adds into the copy of the gradient range that belongs to ITS XCD (block id & 7: block ids 8 apart share
// an XCD), so a line is only ever touched from one L2; clift_grad_shards_fold adds the eight copies into the real gradients once per pass.
// The range and the shards are described by a 40-byte record in device memory (clift_bind_grad_shards publishes its address once, like the
// row limit); `enabled` is set by clift_grad_shards_begin and cleared by the fold, both stream-ordered, so a backward pass outside such a
// bracket (the autograd path, tests) adds straight into the gradients as before.
struct GradShardDesc {
    long lo;          // address of the first gradient float of the sharded range
    long bytes;       // length of the range
    long shard0;      // address of shard 0; shard x at shard0 + x * stride
    long stride;      // bytes between shards
    int enabled;
    int pad;
};
static __device__ const GradShardDesc* g_grad_shard_desc = nullptr;
// block-uniform: where this block adds its partial of the gradient tensor at `p`
// (returned as `p + shift`, not as a pointer rebuilt from an integer: the compiler then still knows it is GLOBAL memory and emits
// global_atomic_add_f32 -- a pointer made from an integer is generic, and the flush became flat_atomic_add_f32)
__device__ __forceinline__ float* grad_target(float* p) {
    const GradShardDesc* d = g_grad_shard_desc;
    if (d == nullptr || p == nullptr) return p;
    const long off = (long)(uintptr_t)p - d->lo;
    if (!d->enabled || off < 0 || off >= d->bytes) return p;
    const long shift = d->shard0 + (long)(blockIdx.x & 7) * d->stride - d->lo;          // bytes from the gradient range to this XCD's shard
    return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + shift);
}
#define CLIFT_ROWS_LIMIT_BINDER(tu) \
    void clift_bind_rows_limit_##tu(const int* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_limit), &p, sizeof(p)); } \
    void clift_bind_grad_shards_##tu(const void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_grad_shard_desc), &p, sizeof(p)); }

// ----------------------------------------------------------------------------- wave primitives (64 lanes)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// inclusive scans over the wave, lane 0 first
__device__ __forceinline__ float wave_incl_prod(float v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d);
        if (l >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d);
        if (l >= d) v += o;
    }
    return v;
}
// inclusive suffix sum: lane l gets sum over lanes >= l
__device__ __forceinline__ float wave_incl_suffix_sum(float v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_down(v, d);
        if (l + d < 64) v += o;
    }
    return v;
}

// ----------------------------------------------------------------------------- sampling geometry
// Bit-faithful to the reference's un-fused fp32 elementwise ops (renderer.py:800-817, 633-634): every
// multiply/add is rounded separately (no FMA contraction), so z, the points and the in-box mask are exact.
struct RayG {
    float o[3], d[3];
    float tmin;
};

__device__ __forceinline__ RayG load_ray(const float* __restrict__ rays, int r, const MarchP& m) {
    RayG g;
    const float4 a = *reinterpret_cast<const float4*>(rays + (size_t)r * 8);
    const float4 b = *reinterpret_cast<const float4*>(rays + (size_t)r * 8 + 4);
    g.o[0] = a.x; g.o[1] = a.y; g.o[2] = a.z; g.d[0] = a.w; g.d[1] = b.x; g.d[2] = b.y;
    const float nearv = b.z, farv = b.w;
    float t = -INFINITY;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float vec = (g.d[i] == 0.f) ? 1e-6f : g.d[i];
        const float ra = __fdiv_rn(__fsub_rn(m.hi[i], g.o[i]), vec);
        const float rb = __fdiv_rn(__fsub_rn(m.lo[i], g.o[i]), vec);
        t = fmaxf(t, fminf(ra, rb));
    }
    g.tmin = fminf(fmaxf(t, nearv), farv);
    return g;
}

__device__ __forceinline__ float sample_z(const RayG& g, const MarchP& m, int k, float jit) {
    const float kf = __fadd_rn((float)k, jit);
    return __fadd_rn(g.tmin, __fmul_rn(m.step, kf));
}

// returns in-box flag; xn = normalised coordinates in [-1,1]
__device__ __forceinline__ bool sample_xn(const RayG& g, const MarchP& m, float z, float xn[3]) {
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float p = __fadd_rn(g.o[i], __fmul_rn(g.d[i], z));
        in = in && !(m.lo[i] > p) && !(p > m.hi[i]);
        xn[i] = __fsub_rn(__fmul_rn(__fsub_rn(p, m.lo[i]), m.inv2[i]), 1.0f);
    }
    return in;
}

// ----------------------------------------------------------------------------- VM tap geometry
// grid_sample(bilinear, align_corners=True, padding zeros): index = (x+1)/2*(size-1); weights from the
// far corner, taps outside [0,size) contribute zero (weight forced to 0, index clamped).
struct Tap2 {
    int i0, i1;    // clamped indices
    float w0, w1;  // weights (0 where the tap is out of range)
};
__device__ __forceinline__ Tap2 make_tap(float x, int size) {
    Tap2 t;
    const float f = ((x + 1.0f) / 2.0f) * (float)(size - 1);
    const float fl = floorf(f);
    const int i0 = (int)fl;
    const int i1 = i0 + 1;
    t.w0 = ((fl + 1.0f) - f);
    t.w1 = (f - fl);
    if (i0 < 0 || i0 >= size) t.w0 = 0.f;
    if (i1 < 0 || i1 >= size) t.w1 = 0.f;
    t.i0 = min(max(i0, 0), size - 1);
    t.i1 = min(max(i1, 0), size - 1);
    return t;
}

// Parity of the texel a tap pair starts at: index i0 when it is in range, else the one before i1 (scatter walks: parity slots).
__device__ __forceinline__ int tap_parity(const Tap2& tp) { return tp.w0 != 0.f ? (tp.i0 & 1) : ((tp.i1 & 1) ^ 1); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 acc) {
    acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y); acc.z = fmaf(s, a.z, acc.z); acc.w = fmaf(s, a.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float f4_hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
    unsafeAtomicAdd(p + 0, v.x); unsafeAtomicAdd(p + 1, v.y); unsafeAtomicAdd(p + 2, v.z); unsafeAtomicAdd(p + 3, v.w);
}
// XCD-private accumulation.  MI355X has 8 XCDs with private, mutually non-coherent L2s; a device-scope fp32 atomic
// is one fabric transaction to the memory side (measured here: ~40-75 G atomics/s).  An atomic that only has to be
// coherent inside one XCD executes in that XCD's L2.  Each XCD therefore accumulates into its OWN copy of the
// gradient table, selected by the hardware XCC id (read from HW_REG_XCC_ID, so it is correct for any
// workgroup->XCD placement), and a streaming kernel sums the 8 copies afterwards (kernel boundary = L2 write-back).
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7; }
__device__ __forceinline__ void atomic_add4_xcd(float* p, float4 v) {
    __hip_atomic_fetch_add(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// matrix_mode / vector_mode of the reference (tensoRF.py:61-62)
__device__ __forceinline__ void vm_axes(int i, int& a, int& b, int& v) {
    a = (i == 2) ? 1 : 0;
    b = (i == 0) ? 1 : 2;
    v = 2 - i;
}

// Bilinear (plane i) and linear (line i) interpolation of one 4-channel group at channel offset c4.
struct VmTaps {
    Tap2 tx, ty, tz;
};
__device__ __forceinline__ VmTaps vm_taps(const VmP& t, int i, const float xn[3]) {
    int a, b, v;
    vm_axes(i, a, b, v);
    VmTaps o;
    o.tx = make_tap(xn[a], t.res[a]);
    o.ty = make_tap(xn[b], t.res[b]);
    o.tz = make_tap(xn[v], t.res[v]);
    return o;
}
__device__ __forceinline__ float4 vm_plane4(const VmP& t, int i, const VmTaps& k, int c4) {
    int a, b, v;
    vm_axes(i, a, b, v);
    const int W = t.res[a], C = t.comps;
    const float* p = t.plane[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = f4_fma(k.tx.w0 * k.ty.w0, ld4(p + ((size_t)k.ty.i0 * W + k.tx.i0) * C + c4), acc);
    acc = f4_fma(k.tx.w1 * k.ty.w0, ld4(p + ((size_t)k.ty.i0 * W + k.tx.i1) * C + c4), acc);
    acc = f4_fma(k.tx.w0 * k.ty.w1, ld4(p + ((size_t)k.ty.i1 * W + k.tx.i0) * C + c4), acc);
    acc = f4_fma(k.tx.w1 * k.ty.w1, ld4(p + ((size_t)k.ty.i1 * W + k.tx.i1) * C + c4), acc);
    return acc;
}
__device__ __forceinline__ float4 vm_line4(const VmP& t, int i, const VmTaps& k, int c4) {
    const int C = t.comps;
    const float* p = t.line[i];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = f4_fma(k.tz.w0, ld4(p + (size_t)k.tz.i0 * C + c4), acc);
    acc = f4_fma(k.tz.w1, ld4(p + (size_t)k.tz.i1 * C + c4), acc);
    return acc;
}
// scatter d(plane*line) for one 4-channel group: gP = upstream * line value, gL = upstream * plane value.
// Plane taps go to global memory (per-XCD copy when xcd_stride > 0).  Line taps: every sample of the batch hits
// the same few hundred line texels, so global atomics on them serialise thousands-deep; with LDS_LINES the block
// accumulates them in LDS (lds_line = this line's [R][C] slab) and flushes once at the end (scatter_flush_lines).
template <bool XCD, bool LDS_LINES>
__device__ __forceinline__ void vm_scatter4_t(const VmP& t, const VmG& g, int i, const VmTaps& k, int c4, float4 gP, float4 gL, size_t xoff,
                                              float* lds_line) {
    int a, b, v;
    vm_axes(i, a, b, v);
    const int W = t.res[a], C = t.comps;
    float* p = g.plane[i] + xoff;
    const float w00 = k.tx.w0 * k.ty.w0, w10 = k.tx.w1 * k.ty.w0, w01 = k.tx.w0 * k.ty.w1, w11 = k.tx.w1 * k.ty.w1;
#define CLIFT_ADD4(ptr, val) do { if (XCD) atomic_add4_xcd(ptr, val); else atomic_add4(ptr, val); } while (0)
    if (w00 != 0.f) CLIFT_ADD4(p + ((size_t)k.ty.i0 * W + k.tx.i0) * C + c4, f4_scale(w00, gP));
    if (w10 != 0.f) CLIFT_ADD4(p + ((size_t)k.ty.i0 * W + k.tx.i1) * C + c4, f4_scale(w10, gP));
    if (w01 != 0.f) CLIFT_ADD4(p + ((size_t)k.ty.i1 * W + k.tx.i0) * C + c4, f4_scale(w01, gP));
    if (w11 != 0.f) CLIFT_ADD4(p + ((size_t)k.ty.i1 * W + k.tx.i1) * C + c4, f4_scale(w11, gP));
    if (LDS_LINES) {
        if (k.tz.w0 != 0.f) { float* q = lds_line + k.tz.i0 * C + c4; const float4 x = f4_scale(k.tz.w0, gL);
            atomicAdd(q, x.x); atomicAdd(q + 1, x.y); atomicAdd(q + 2, x.z); atomicAdd(q + 3, x.w); }
        if (k.tz.w1 != 0.f) { float* q = lds_line + k.tz.i1 * C + c4; const float4 x = f4_scale(k.tz.w1, gL);
            atomicAdd(q, x.x); atomicAdd(q + 1, x.y); atomicAdd(q + 2, x.z); atomicAdd(q + 3, x.w); }
    } else {
        float* l = g.line[i] + xoff;
        if (k.tz.w0 != 0.f) CLIFT_ADD4(l + (size_t)k.tz.i0 * C + c4, f4_scale(k.tz.w0, gL));
        if (k.tz.w1 != 0.f) CLIFT_ADD4(l + (size_t)k.tz.i1 * C + c4, f4_scale(k.tz.w1, gL));
    }
#undef CLIFT_ADD4
}
template <bool LDS_LINES>
__device__ __forceinline__ void vm_scatter4(const VmP& t, const VmG& g, int i, const VmTaps& k, int c4, float4 gP, float4 gL, size_t xoff,
                                            float* lds_line) {
    if (g.xcd_stride > 0) vm_scatter4_t<true, LDS_LINES>(t, g, i, k, c4, gP, gL, xoff, lds_line);
    else vm_scatter4_t<false, LDS_LINES>(t, g, i, k, c4, gP, gL, xoff, lds_line);
}
// LDS line accumulators: slab i starts at line_lds_offset(t, i) floats, size res[v_i]*comps.
__device__ __forceinline__ int line_lds_offset(const VmP& t, int i) {
    int off = 0;
    for (int j = 0; j < i; ++j) off += t.res[2 - j] * t.comps;
    return off;
}
__host__ __device__ __forceinline__ int line_lds_floats(const int res[3], int comps) { return (res[0] + res[1] + res[2]) * comps; }
// Launch geometry of the persistent scatter kernels: as many resident threads per CU as the LDS line slab allows
// (160 KiB LDS per CU): <= 40 KiB -> 4 x 256 threads, <= 80 KiB -> 2 x 512, <= 160 KiB -> 1 x 1024, else no LDS slab.
static inline bool scatter_geometry(int lds_bytes, int* threads, int* per_cu) {
    if (lds_bytes <= 40 * 1024) { *threads = 256; *per_cu = 4; return true; }
    if (lds_bytes <= 80 * 1024) { *threads = 512; *per_cu = 2; return true; }
    if (lds_bytes <= 160 * 1024 - 512) { *threads = 1024; *per_cu = 1; return true; }
    *threads = 256; *per_cu = 4;
    return false;
}
__device__ __forceinline__ void scatter_zero_lines(float* lds, int n) {
    for (int e = threadIdx.x; e < n; e += blockDim.x) lds[e] = 0.f;
    __syncthreads();
}
__device__ __forceinline__ void scatter_flush_lines(const VmP& t, const VmG& g, float* lds, size_t xoff) {
    __syncthreads();
    int base = 0;
    for (int i = 0; i < 3; ++i) {
        const int n = t.res[2 - i] * t.comps;
        float* dst = g.line[i] + xoff;
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const float v = lds[base + e];
            if (v != 0.f) {
                if (g.xcd_stride > 0) __hip_atomic_fetch_add(dst + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else unsafeAtomicAdd(dst + e, v);
            }
        }
        base += n;
    }
}
