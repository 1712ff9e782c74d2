// heads_io.hip -- appearance VM gather (a9), MLP input assembly (a10) and alpha compositing (a13), fwd + bwd.
#include "clift_dev.h"
#include <type_traits>
CLIFT_ROWS_LIMIT_BINDER(heads_io)
#include <stdlib.h>
#include <string.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ============================================================================ appearance gather
// Thread = (active sample, 4-channel group).  The 3*comps/4 threads of one sample write one contiguous row of
// F and, per tap, read contiguous 16-byte pieces of one channels-last texel.
__global__ __launch_bounds__(256) void k_app_gather_fwd(MarchP m, VmP t, const float* __restrict__ rays, const float* __restrict__ jitter,
                                                         const int* __restrict__ act, long total, float* __restrict__ F,
                                                         float* __restrict__ xa) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g4 = t.comps / 4, G = 3 * g4;
    const int s = (int)(gid / G), j = (int)(gid - (long)s * G);
    if (rows_cut(s)) return;
    const int i = j / g4, c4 = (j - i * g4) * 4;
    const int sid = act[s];
    const int r = sid / m.S, k = sid - r * m.S;
    const RayG g = load_ray(rays, r, m);
    float xn[3];
    sample_xn(g, m, sample_z(g, m, k, jitter ? jitter[r] : 0.f), xn);
    const VmTaps tp = vm_taps(t, i, xn);
    const float4 v = f4_mul(vm_plane4(t, i, tp, c4), vm_line4(t, i, tp, c4));
    *reinterpret_cast<float4*>(F + (size_t)s * (3 * t.comps) + i * t.comps + c4) = v;
    if (j == 0 && xa) *reinterpret_cast<float4*>(xa + (size_t)s * 4) = make_float4(xn[0], xn[1], xn[2], 0.f);
}

extern "C" int clift_app_gather_fwd(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter,
                                    const int* act_idx, int M, float* F, float* xa, clift_stream_t s) {
    CLIFT_REQUIRE(h_app->comps % 4 == 0, "clift_app_gather_fwd: comps must be a multiple of 4");
    if (M <= 0) return 0;
    const long total = (long)M * (3 * h_app->comps / 4);
    k_app_gather_fwd<<<cdiv(total, 256), 256, 0, as_stream(s)>>>(to_dev(h_m), to_dev(h_app), rays, jitter, act_idx, total, F, xa);
    return clift_check_launch("clift_app_gather_fwd");
}

// plane x line products at arbitrary normalised points (reference field API compute_appearance_feature, tensoRF.py:127-137)
__global__ __launch_bounds__(256) void k_vm_products_points(VmP t, const float* __restrict__ xn3, int ldx, long total, float* __restrict__ F) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int g4 = t.comps / 4, G = 3 * g4;
    const long s = gid / G;
    const int j = (int)(gid - s * G);
    const int i = j / g4, c4 = (j - i * g4) * 4;
    const float xn[3] = {xn3[s * ldx + 0], xn3[s * ldx + 1], xn3[s * ldx + 2]};
    const VmTaps tp = vm_taps(t, i, xn);
    *reinterpret_cast<float4*>(F + (size_t)s * (3 * t.comps) + i * t.comps + c4) = f4_mul(vm_plane4(t, i, tp, c4), vm_line4(t, i, tp, c4));
}

extern "C" int clift_vm_products_points(const clift_vm_t* h_vm, const float* xn, int ldx, long n, float* F, clift_stream_t s) {
    CLIFT_REQUIRE(h_vm->comps % 4 == 0, "clift_vm_products_points: comps must be a multiple of 4");
    if (n <= 0) return 0;
    const long total = n * (3 * h_vm->comps / 4);
    k_vm_products_points<<<cdiv(total, 256), 256, 0, as_stream(s)>>>(to_dev(h_vm), xn, ldx, total, F);
    return clift_check_launch("clift_vm_products_points");
}

// MLP input assembly from explicit view directions (reference field API render_appearance_mlp(viewdirs, features))
__global__ __launch_bounds__(256) void k_app_encode_points(const float* __restrict__ feat, int ldf, int nf, int pef, int pev,
                                                            const float* __restrict__ dirs, int ldd, long total, float* __restrict__ X, int ldx) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const long s = gid / ldx;
    const int c = (int)(gid - s * ldx);
    const float* f = feat + (size_t)s * ldf;
    const float* d = dirs + (size_t)s * ldd;
    const int b0 = nf, b1 = b0 + 3, b2 = b1 + nf * pef, b3 = b2 + nf * pef, b4 = b3 + 3 * pev, b5 = b4 + 3 * pev;
    float v = 0.f;
    if (c < b0) v = f[c];
    else if (c < b1) v = d[c - b0];
    else if (c < b2) { const int e = c - b1; v = sinf(f[e / pef] * (float)(1 << (e % pef))); }
    else if (c < b3) { const int e = c - b2; v = cosf(f[e / pef] * (float)(1 << (e % pef))); }
    else if (c < b4) { const int e = c - b3; v = sinf(d[e / pev] * (float)(1 << (e % pev))); }
    else if (c < b5) { const int e = c - b4; v = cosf(d[e / pev] * (float)(1 << (e % pev))); }
    X[gid] = v;
}

extern "C" int clift_app_encode_points(const float* feat, int ldf, int nf, int pe_feat, int pe_view, const float* dirs, int ldd, long n,
                                       float* X, int ldx, clift_stream_t s) {
    CLIFT_REQUIRE(ldx >= nf + 3 + 2 * pe_feat * nf + 2 * pe_view * 3, "clift_app_encode_points: ldx too small");
    if (n <= 0) return 0;
    k_app_encode_points<<<cdiv(n * ldx, 256), 256, 0, as_stream(s)>>>(feat, ldf, nf, pe_feat, pe_view, dirs, ldd, n * ldx, X, ldx);
    return clift_check_launch("clift_app_encode_points");
}

// normalised coordinates of the active samples only (instance / segment passes, renderer.py:204,285)
__global__ __launch_bounds__(256) void k_active_xyz(MarchP m, const float* __restrict__ rays, const float* __restrict__ jitter,
                                                     const int* __restrict__ act, int M, float* __restrict__ xa) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= limit_rows(M)) return;
    const int sid = act[s];
    const int r = sid / m.S, k = sid - r * m.S;
    const RayG g = load_ray(rays, r, m);
    float xn[3];
    sample_xn(g, m, sample_z(g, m, k, jitter ? jitter[r] : 0.f), xn);
    *reinterpret_cast<float4*>(xa + (size_t)s * 4) = make_float4(xn[0], xn[1], xn[2], 0.f);
}

extern "C" int clift_active_xyz(const clift_march_t* h_m, const float* rays, const float* jitter, const int* act_idx, int M, float* xa,
                                clift_stream_t s) {
    if (M <= 0) return 0;
    k_active_xyz<<<cdiv(M, 256), 256, 0, as_stream(s)>>>(to_dev(h_m), rays, jitter, act_idx, M, xa);
    return clift_check_launch("clift_active_xyz");
}

// Backward of the appearance gather.  Persistent blocks, ONE (plane, channel) PER LANE: every atomic instruction of a wave
// covers runs of `comps` consecutive floats of one texel, which the memory pipeline folds into cache-line-granular requests
// (measured 3.8x faster than a float4-per-lane mapping whose lanes are 16 B apart within an instruction).
// Like k_density_bwd (march.hip) a lane WALKS a short run of consecutive active samples -- consecutive samples of one ray
// in the compacted order -- keeping its plane's four current texels in registers (key, table value, gradient sum); a texel
// costs one load when the walk enters it and one atomic when the walk leaves it.  Line gradients accumulate in LDS and are
// flushed once per block; plane gradients go to the per-XCD accumulation copies.
// Measured at 265 k active samples (profiles/r01_scatter_notes.txt): 1 -> 667 us (no merging; the pre-walk kernel: 624), 2 -> 520,
// 4 -> 469, 8 -> 461; with the LINE entries merged in registers as well (round 2, profiles/r02_scatter_notes.txt): 4 -> 359, 8 -> 327,
// 16 -> 318, 32 -> 311.  Used when the caller has no xa or comps > 64; otherwise k_app_gather_bwd_u below (237 us).
constexpr int APP_SEG = 16;

template <bool LDS_LINES>
__global__ __launch_bounds__(1024) void k_app_gather_bwd(MarchP m, VmP t, VmG gr, const float* __restrict__ rays,
                                                            const float* __restrict__ jitter, const int* __restrict__ act, int M,
                                                            const float* __restrict__ dF, int seg_len, const float* __restrict__ xa) {
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    const int nl = line_lds_floats(t.res, t.comps);
    if (LDS_LINES) scatter_zero_lines(lds_lines, nl);
    const size_t xoff = gr.xcd_stride > 0 ? (size_t)xcc_id() * (size_t)gr.xcd_stride : 0;
    const bool xcd = gr.xcd_stride > 0;
    const int C = t.comps, G = 3 * C;
    const long nthreads = (long)gridDim.x * blockDim.x;
    M = limit_rows(M);
    const long total = (long)((M + seg_len - 1) / seg_len) * G;
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += nthreads) {
        const int w = (int)(gid / G), j = (int)(gid - (long)w * G);
        const int i = j / C, c = j - i * C;
        int a, b, v;
        vm_axes(i, a, b, v);
        const int W = t.res[a];
        const float* pp = t.plane[i] + c;
        const float* lp = t.line[i] + c;
        float* gp = gr.plane[i] + xoff + c;
        float* ll = lds_lines + line_lds_offset(t, i) + c;
        float* gl = gr.line[i] + xoff + c;
        int ck[4] = {-1, -1, -1, -1};      // open texels (y * W + x), -1 = empty
        float cv[4] = {0.f, 0.f, 0.f, 0.f}, ca[4] = {0.f, 0.f, 0.f, 0.f};
        // the two open LINE entries, same idea: the LDS float atomics are the slowest thing in this kernel (probe: 491 -> 158 us without
        // them, ~2.3 clocks per lane-op per CU), and a ray stays ~3 samples between two line texels
        int lk[2] = {-1, -1};
        float lacc[2] = {0.f, 0.f};
        auto line_out = [&](int key, float val) {
            if (LDS_LINES) atomicAdd(ll + key * C, val);
            else if (xcd) __hip_atomic_fetch_add(gl + (size_t)key * C, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else unsafeAtomicAdd(gl + (size_t)key * C, val);
        };
        const int s0 = w * seg_len, s1 = min(M, s0 + seg_len);
        for (int s = s0; s < s1; ++s) {
            float xn[3];
            if (xa) {                                // the forward gather left the normalised position of every active sample in xa (M, 4):
                const float4 p4 = *reinterpret_cast<const float4*>(xa + (size_t)s * 4);      // one broadcast 16-byte load instead of an
                xn[0] = p4.x; xn[1] = p4.y; xn[2] = p4.z;                                    // integer division + the ray's slab set-up
            } else {
                const int sid = act[s];
                const int r = sid / m.S, k = sid - r * m.S;
                const RayG g = load_ray(rays, r, m);
                sample_xn(g, m, sample_z(g, m, k, jitter ? jitter[r] : 0.f), xn);
            }
            const Tap2 tx = make_tap(xn[a], t.res[a]), ty = make_tap(xn[b], t.res[b]), tz = make_tap(xn[v], t.res[v]);
            const float w4[4] = {tx.w0 * ty.w0, tx.w1 * ty.w0, tx.w0 * ty.w1, tx.w1 * ty.w1};
            int nk[4] = {ty.i0 * W + tx.i0, ty.i0 * W + tx.i1, ty.i1 * W + tx.i0, ty.i1 * W + tx.i1};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (w4[q] == 0.f) nk[q] = -1;        // clamped out-of-range taps: never loaded, never written
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // leave
                const int key = ck[q];
                if (key >= 0 && key != nk[0] && key != nk[1] && key != nk[2] && key != nk[3]) {
                    if (xcd) __hip_atomic_fetch_add(gp + (size_t)key * C, ca[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else unsafeAtomicAdd(gp + (size_t)key * C, ca[q]);
                }
            }
            float nv[4], na[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // enter
                float val = 0.f, sum = 0.f;
                bool hit = false;
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (nk[q] >= 0 && ck[o] == nk[q]) { val = cv[o]; sum = ca[o]; hit = true; }
                if (!hit && nk[q] >= 0) val = pp[(size_t)nk[q] * C];
                nv[q] = val; na[q] = sum;
            }
            const float P = fmaf(w4[3], nv[3], fmaf(w4[2], nv[2], fmaf(w4[1], nv[1], w4[0] * nv[0])));
            const float L = fmaf(tz.w1, lp[(size_t)tz.i1 * C], tz.w0 * lp[(size_t)tz.i0 * C]);
            const float d = dF[(size_t)s * G + j];
            const float gP = d * L, gL = d * P;
#pragma unroll
            for (int q = 0; q < 4; ++q) { ck[q] = nk[q]; cv[q] = nv[q]; ca[q] = fmaf(w4[q], gP, na[q]); }
            const int n0 = tz.w0 != 0.f ? tz.i0 : -1, n1 = tz.w1 != 0.f ? tz.i1 : -1;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (lk[q] >= 0 && lk[q] != n0 && lk[q] != n1) line_out(lk[q], lacc[q]);
            const float a0 = n0 < 0 ? 0.f : (lk[0] == n0 ? lacc[0] : (lk[1] == n0 ? lacc[1] : 0.f));
            const float a1 = n1 < 0 ? 0.f : (lk[0] == n1 ? lacc[0] : (lk[1] == n1 ? lacc[1] : 0.f));
            lk[0] = n0; lk[1] = n1;
            lacc[0] = fmaf(tz.w0, gL, a0); lacc[1] = fmaf(tz.w1, gL, a1);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (lk[q] >= 0) line_out(lk[q], lacc[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ck[q] >= 0) {
                if (xcd) __hip_atomic_fetch_add(gp + (size_t)ck[q] * C, ca[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else unsafeAtomicAdd(gp + (size_t)ck[q] * C, ca[q]);
            }
    }
    if (LDS_LINES) scatter_flush_lines(t, gr, lds_lines, xoff);
}

// ---------------------------------------------------------------------------- the same walk, one WAVE per (segment, plane)
// With the line entries merged the walk above is bound by its own instruction stream (probes at 265 k samples: 327 us; 315 without the plane
// atomics, 337 without the line output): ~300 VALU instructions per step of tap arithmetic and key matching, identical in all `comps` lanes
// of a plane.  (A float4-of-channels-per-lane form shares that arithmetic -- 132 us without its atomics -- but its plane atomics, four
// scalars per lane 16 bytes apart, cost 580 us: atomics must stay one contiguous run of floats per instruction.)
// Here (comps <= 64) a wave owns ONE (segment of AU_SEG consecutive active samples, plane) pair, lane = channel, and the walk is split
// like k_density_bwd_u (march.hip):
//   phase 1: lane p computes step p of the walk's INDEX work, all steps at once: taps, and the footprint's four texels filed in PARITY
//            SLOTS -- texel (x, y) always sits in slot (x & 1) + 2 (y & 1); a 2 x 2 footprint has one texel of each parity, so a footprint
//            that moves replaces texels IN PLACE and a slot is written out exactly when its key changes (no 4 x 4 key matching, no sums
//            moving between slots); with the keys of step p-1 from the neighbouring lane: one "write out" and one "restart" bit per slot,
//            same for the two line entries (slot = z & 1); a 64-byte record per step in the wave's LDS slot;
//   phase 2: the serial walk: the table values and dF rows of AU_U steps are loaded together (one memory round trip per AU_U steps,
//            nothing but the six gradient sums carried from step to step), then per step <= 6 atomics, 6 restarts, ~12 FMAs.
// Per-sample terms: the bilinear sum runs in slot order instead of corner order (fp32 round-off apart from the walk above).
#ifndef AU_ABL
#define AU_ABL 0          // timing probes (tools/jobs/app_probe.sh): 1 = no plane atomics, 2 = no table loads, 4 = three blocks of eight waves per CU
#endif
constexpr int AU_SEG = 32;
constexpr int AU_U = 4;                // steps whose loads are issued together
struct alignas(16) WalkRec {
    int ks[4];          // texels by parity slot, as BYTE offsets (y * W + x) * comps * 4 into the channels-last table; -1 = tap out of range
    float ws[4];        // their bilinear weights
    int kz[2];          // line entries by parity slot (z & 1), byte offsets z * comps * 4; -1 = out of range
    float wz[2];
    int ctrl;           // bits 0-3 / 4-5: write out plane slot s / line slot s before this step; bits 8-11 / 12-13: restart its sum
    int pad[3];
};

template <bool LDS_LINES>
__global__ __launch_bounds__(1024) void k_app_gather_bwd_u(VmP t, VmG gr, int M, const float* __restrict__ dF, const float* __restrict__ xa,
                                                              int seg_len) {
    // Round 6: a block walks ONE plane (blockIdx.x % 3): its LDS line slab holds that plane's line only -- a third of the three-line slab
    // (24.5 instead of 73.7 KB at 128^3 x 48) -- so two or three blocks fit a CU where one did; the walk is a chain of dependent
    // LDS / memory round trips per wave, i.e. its throughput is the number of resident waves (16 -> 24 per CU: profiles/r06_scatter_probes.txt).
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    const int i = blockIdx.x % 3;
    int a, b, v;
    vm_axes(i, a, b, v);
    const int nl = LDS_LINES ? t.res[v] * t.comps : 0;
    if (LDS_LINES) scatter_zero_lines(lds_lines, nl);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    WalkRec* const recs = reinterpret_cast<WalkRec*>(lds_lines + (nl + 3) / 4 * 4) + wave * AU_SEG;
    const size_t xoff = gr.xcd_stride > 0 ? (size_t)xcc_id() * (size_t)gr.xcd_stride : 0;
    const bool xcd = gr.xcd_stride > 0;
    const int C = t.comps, G = 3 * C;
    const int c = lane < C ? lane : C - 1;           // lanes >= comps idle through phase 2 (they repeat the last channel's loads, never write)
    const bool live = lane < C;
    M = __builtin_amdgcn_readfirstlane(limit_rows(M));
    const int nseg = (M + seg_len - 1) / seg_len, pblocks = gridDim.x / 3;      // (the grid is a multiple of 3 blocks)
    for (int seg = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / 3) * nwaves + wave); seg < nseg; seg += pblocks * nwaves) {
        int a, b, v;
        vm_axes(i, a, b, v);
        const int W = t.res[a];
        const int s0 = seg * seg_len, n = __builtin_amdgcn_readfirstlane(min(seg_len, M - s0));
        // ---------------- phase 1
        {
            const int p = lane;
            int ks[4] = {-1, -1, -1, -1}, kz[2] = {-1, -1};
            float ws[4] = {0.f, 0.f, 0.f, 0.f}, wz[2] = {0.f, 0.f};
            if (p < n) {
                const float4 p4 = *reinterpret_cast<const float4*>(xa + (size_t)(s0 + p) * 4);
                const float xn[3] = {p4.x, p4.y, p4.z};
                const Tap2 tx = make_tap(xn[a], t.res[a]), ty = make_tap(xn[b], t.res[b]), tz = make_tap(xn[v], t.res[v]);
                const float w4[4] = {tx.w0 * ty.w0, tx.w1 * ty.w0, tx.w0 * ty.w1, tx.w1 * ty.w1};
                const int nk[4] = {ty.i0 * W + tx.i0, ty.i0 * W + tx.i1, ty.i1 * W + tx.i0, ty.i1 * W + tx.i1};
                const int par = tap_parity(tx) + 2 * tap_parity(ty);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const int q = sl ^ par;                  // corner (qx, qy) sits in slot ((x0 + qx) & 1, (y0 + qy) & 1)
                    const float wq = q == 0 ? w4[0] : q == 1 ? w4[1] : q == 2 ? w4[2] : w4[3];
                    const int kq = q == 0 ? nk[0] : q == 1 ? nk[1] : q == 2 ? nk[2] : nk[3];
                    ws[sl] = wq;
                    ks[sl] = wq == 0.f ? -1 : kq * (4 * C);  // clamped out-of-range taps: never loaded, never written
                }
                const int pz = tap_parity(tz);
                const int z0 = tz.w0 != 0.f ? tz.i0 * (4 * C) : -1, z1 = tz.w1 != 0.f ? tz.i1 * (4 * C) : -1;
                kz[0] = pz ? z1 : z0; kz[1] = pz ? z0 : z1;
                wz[0] = pz ? tz.w1 : tz.w0; wz[1] = pz ? tz.w0 : tz.w1;
            }
            int ctrl = 0;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                int pk = __shfl_up(ks[sl], 1);
                if (p == 0) pk = -1;
                if (pk != ks[sl]) ctrl |= (pk >= 0 ? (1 << sl) : 0) | (1 << (8 + sl));
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                int pk = __shfl_up(kz[sl], 1);
                if (p == 0) pk = -1;
                if (pk != kz[sl]) ctrl |= (pk >= 0 ? (1 << (4 + sl)) : 0) | (1 << (12 + sl));
            }
            if (p < n) {
                int4* dst = reinterpret_cast<int4*>(recs + p);
                dst[0] = make_int4(ks[0], ks[1], ks[2], ks[3]);
                dst[1] = make_int4(__float_as_int(ws[0]), __float_as_int(ws[1]), __float_as_int(ws[2]), __float_as_int(ws[3]));
                dst[2] = make_int4(kz[0], kz[1], __float_as_int(wz[0]), __float_as_int(wz[1]));
                dst[3] = make_int4(ctrl, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2
        // wave-uniform bases + 32-bit byte offsets (record offset + 4 c): one add per access, no 64-bit multiplies
        auto sptr = [](auto* q) {                        // a pointer the compiler can see is wave-uniform (a table picked by a run-time plane index is not, to it)
            const unsigned long long a = (unsigned long long)(uintptr_t)q;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            return reinterpret_cast<decltype(q)>((uintptr_t)(((unsigned long long)hi << 32) | lo));
        };
        const float* pp = sptr(t.plane[i]);
        const float* lp = sptr(t.line[i]);
        float* gp = sptr(gr.plane[i] + xoff);
        float* ll = lds_lines;                           // (this block's slab is line i)
        float* gl = sptr(gr.line[i] + xoff);
        const float* dcol0 = sptr(dF + (size_t)s0 * G + i * C);       // (+ 4 c bytes per lane)
        const int c4 = 4 * c;
        auto at = [](auto* base, int off) {            // (pointer arithmetic, not integer casts: the address space must stay visible)
            typedef typename std::conditional<std::is_const<typename std::remove_pointer<decltype(base)>::type>::value, const char, char>::type B;
            return reinterpret_cast<decltype(base)>(reinterpret_cast<B*>(base) + (unsigned)off);
        };
        // Round 6: the record of a step is the same for every lane of the wave (lane = channel), so its keys and control bits are taken into
        // SCALAR registers (v_readfirstlane): the "write out" / "restart" tests are scalar branches, a table access is <scalar base + key> +
        // <4 c in a VGPR> -- no vector instruction per address, per test or per clamp (the walk was bound by its vector issue slots:
        // profiles/r05_pmc_tables.txt).  An out-of-range tap has weight 0 and a clamped (valid) load address: its term is an exact + 0.
        auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
        // (hand-issued: the compiler folds <base + 4 c> into a vector pair and adds the scalar key with a vector instruction instead)
        auto ld_s = [](const float* sbase, int voff) {   // global_load_dword <4 c>, <scalar base + key>
            float v;
            asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
            return v;
        };
        auto plane_out = [&](int key, float val) {       // key: scalar.  (XCD-private copies or the table itself: the same instruction either way)
            if (!live || (AU_ABL & 1)) return;
            asm volatile("global_atomic_add_f32 %0, %1, %2" : : "v"(c4), "v"(val), "s"(sptr(at(gp, key))) : "memory");
        };
        auto line_out = [&](int key, float val) {
            if (!live) return;
            if (LDS_LINES) atomicAdd(at(at(ll, key), c4), val);
            else asm volatile("global_atomic_add_f32 %0, %1, %2" : : "v"(c4), "v"(val), "s"(sptr(at(gl, key))) : "memory");
        };
        (void)xcd;
        int ck[4] = {-1, -1, -1, -1}, lk[2] = {-1, -1};
        float ca[4] = {0.f, 0.f, 0.f, 0.f}, lacc[2] = {0.f, 0.f};
        for (int p0 = 0; p0 < n; p0 += AU_U) {
            int ks[AU_U][4], kz[AU_U][2];
            float tv[AU_U][4], tl[AU_U][2], td[AU_U];
#pragma unroll
            for (int u = 0; u < AU_U; ++u) {
                const int4* src = reinterpret_cast<const int4*>(recs + min(p0 + u, n - 1));
                const int4 k4 = src[0];
                const int2 k2 = *reinterpret_cast<const int2*>(src + 2);
                ks[u][0] = uni(k4.x); ks[u][1] = uni(k4.y); ks[u][2] = uni(k4.z); ks[u][3] = uni(k4.w);
                kz[u][0] = uni(k2.x); kz[u][1] = uni(k2.y);
            }
#pragma unroll
            for (int u = 0; u < AU_U; ++u) {             // seven loads per step, in this order (the waits below count them)
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) tv[u][sl] = ld_s(at(pp, (AU_ABL & 2) ? 0 : max(ks[u][sl], 0)), c4);
                tl[u][0] = ld_s(at(lp, max(kz[u][0], 0)), c4); tl[u][1] = ld_s(at(lp, max(kz[u][1], 0)), c4);
                td[u] = ld_s(dcol0 + (size_t)min(p0 + u, n - 1) * G, c4);
            }
#pragma unroll
            for (int u = 0; u < AU_U; ++u) {
                // this step's seven values have landed when at most the 7 (AU_U - 1 - u) younger loads are outstanding (atomics issued since
                // then are younger still: they only make the wait stricter)
                if (u == 0) asm volatile("s_waitcnt vmcnt(21)" : "+v"(tv[0][0]), "+v"(tv[0][1]), "+v"(tv[0][2]), "+v"(tv[0][3]), "+v"(tl[0][0]), "+v"(tl[0][1]), "+v"(td[0]) : : "memory");
                if (u == 1) asm volatile("s_waitcnt vmcnt(14)" : "+v"(tv[1][0]), "+v"(tv[1][1]), "+v"(tv[1][2]), "+v"(tv[1][3]), "+v"(tl[1][0]), "+v"(tl[1][1]), "+v"(td[1]) : : "memory");
                if (u == 2) asm volatile("s_waitcnt vmcnt(7)" : "+v"(tv[2][0]), "+v"(tv[2][1]), "+v"(tv[2][2]), "+v"(tv[2][3]), "+v"(tl[2][0]), "+v"(tl[2][1]), "+v"(td[2]) : : "memory");
                if (u == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(tv[3][0]), "+v"(tv[3][1]), "+v"(tv[3][2]), "+v"(tv[3][3]), "+v"(tl[3][0]), "+v"(tl[3][1]), "+v"(td[3]) : : "memory");
                if (p0 + u < n) {                                              // uniform
                    const int4* src = reinterpret_cast<const int4*>(recs + (p0 + u));
                    const int4 r1 = src[1], r2 = src[2];
                    const int ctrl = uni(src[3].x);
                    const float ws[4] = {__int_as_float(r1.x), __int_as_float(r1.y), __int_as_float(r1.z), __int_as_float(r1.w)};
                    const float wz[2] = {__int_as_float(r2.z), __int_as_float(r2.w)};
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        if (ctrl & (1 << sl)) plane_out(ck[sl], ca[sl]);
                        if (ctrl & (1 << (8 + sl))) ca[sl] = 0.f;
                        ck[sl] = ks[u][sl];
                    }
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        if (ctrl & (1 << (4 + sl))) line_out(lk[sl], lacc[sl]);
                        if (ctrl & (1 << (12 + sl))) lacc[sl] = 0.f;
                        lk[sl] = kz[u][sl];
                    }
                    float P = 0.f;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) P = fmaf(ws[sl], tv[u][sl], P);
                    const float L = fmaf(wz[1], tl[u][1], wz[0] * tl[u][0]);
                    const float gP = td[u] * L, gL = td[u] * P;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) ca[sl] = fmaf(ws[sl], gP, ca[sl]);
                    lacc[0] = fmaf(wz[0], gL, lacc[0]); lacc[1] = fmaf(wz[1], gL, lacc[1]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (lk[q] >= 0) line_out(lk[q], lacc[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ck[q] >= 0) plane_out(ck[q], ca[q]);
        __builtin_amdgcn_wave_barrier();               // the next item's phase 1 overwrites the records
    }
    if (LDS_LINES) {                                     // flush this block's line slab
        __syncthreads();
        float* dst = gr.line[i] + xoff;
        for (int e = threadIdx.x; e < nl; e += blockDim.x) {
            const float val = lds_lines[e];
            if (val != 0.f) unsafeAtomicAdd(dst + e, val);
        }
    }
}

extern "C" int clift_app_gather_bwd(const clift_march_t* h_m, const clift_vm_t* h_app, const clift_vm_grad_t* h_grad,
                                    const float* rays, const float* jitter, const int* act_idx, int M, const float* dF,
                                    const float* xa, clift_stream_t s) {
    CLIFT_REQUIRE(h_app->comps % 4 == 0, "clift_app_gather_bwd: comps must be a multiple of 4");
    if (M <= 0) return 0;
    const int seg = APP_SEG;
    const int lds_bytes = line_lds_floats(h_app->res, h_app->comps) * 4;
    int threads, per_cu;
    const bool use_lds = scatter_geometry(lds_bytes, &threads, &per_cu);
    if (xa != nullptr && h_app->comps <= 64) {          // (otherwise: the lane-per-(plane, channel) walk, the only form without xa / for comps > 64)
        // one wave per (segment, plane): the line slab + a 2 KB record slot per wave; as many waves per CU as that allows
        const int rec_bytes = AU_SEG * (int)sizeof(WalkRec);
        int rmax = h_app->res[0] > h_app->res[1] ? h_app->res[0] : h_app->res[1];
        rmax = rmax > h_app->res[2] ? rmax : h_app->res[2];
        const int slab = (rmax * h_app->comps * 4 + 15) / 16 * 16;      // ONE line per block (the longest: every block gets the same allocation)
        int wpb = (AU_ABL & 32) ? 9 : (AU_ABL & 64) ? 7 : 8, bpc = (AU_ABL & 64) ? 4 : 3;   // waves per block, blocks per CU: 24 resident waves
        bool lds_l = true;
        const int useg = AU_SEG;
        if (3 * (slab + 8 * rec_bytes) > 160 * 1024 - 1536) { wpb = 12; bpc = 2; }
        if (bpc == 2 && 2 * (slab + 12 * rec_bytes) > 160 * 1024 - 1024) { wpb = 16; bpc = 1; }
        if (bpc == 1 && slab + 16 * rec_bytes > 160 * 1024 - 512) { lds_l = false; wpb = 8; bpc = 3; }
        const int nseg = cdiv(M, useg);
        int blocks = 3 * cdiv(nseg, wpb);
        const int cap = (clift_persistent_cus() * bpc) / 3 * 3;
        if (blocks > cap) blocks = cap;
        if (blocks < 3) blocks = 3;
        const int dyn = (lds_l ? slab : 0) + wpb * rec_bytes;
        if (lds_l) {
            if (dyn > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_app_gather_bwd_u<true>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            k_app_gather_bwd_u<true><<<blocks, wpb * 64, dyn, as_stream(s)>>>(to_dev(h_app), to_dev(h_grad), M, dF, xa, useg);
        } else {
            k_app_gather_bwd_u<false><<<blocks, wpb * 64, dyn, as_stream(s)>>>(to_dev(h_app), to_dev(h_grad), M, dF, xa, useg);
        }
        return clift_check_launch("clift_app_gather_bwd");
    }
    const long total = (long)cdiv(M, seg) * 3 * h_app->comps;
    const int want = cdiv(total, threads);
    const int blocks = want < clift_persistent_cus() * per_cu ? want : clift_persistent_cus() * per_cu;
    if (use_lds) {
        if (lds_bytes > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_app_gather_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        k_app_gather_bwd<true><<<blocks, threads, lds_bytes, as_stream(s)>>>(to_dev(h_m), to_dev(h_app), to_dev(h_grad), rays, jitter, act_idx, M, dF, seg, xa);
    } else {
        k_app_gather_bwd<false><<<blocks, threads, 0, as_stream(s)>>>(to_dev(h_m), to_dev(h_app), to_dev(h_grad), rays, jitter, act_idx, M, dF, seg, xa);
    }
    return clift_check_launch("clift_app_gather_bwd");
}

// ============================================================================ appearance MLP input
// Column map (tensoRF.py:401-408,413-418): [feat(nf) | dir(3) | sin(feat_f*2^p) (f-major, p minor) | cos(...) |
// sin(dir_a*2^p) | cos(...) | zero pad].
// Thread = (sample, SOURCE element j): j < nf a feature, nf <= j < nf + 3 a view-direction component, j = nf + 3 the zero pad.
// One sincosf per (element, frequency) yields both encodings (the per-output-column form evaluated every sine and cosine
// separately with a full range reduction each: 226 us -> see profiles/r01_v4 for 265 k samples); neighbouring threads write
// neighbouring columns in every block of the row.
__global__ __launch_bounds__(256) void k_app_encode_fwd(const float* __restrict__ feat, int ldf, int nf, int pef, int pev,
                                                         const float* __restrict__ rays, const int* __restrict__ act, int S, long total,
                                                         float* __restrict__ X, int ldx, int x_bf16) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int J = nf + 4;
    const int s = (int)(gid / J), j = (int)(gid - (long)s * J);
    if (rows_cut(s)) return;
    // the row is written through `put` so that it can be bf16-stored (bf16 mode: it is only ever a matrix-core operand)
    float* xf = X + (size_t)s * ldx;
    unsigned short* xh = reinterpret_cast<unsigned short*>(X) + (size_t)s * ldx;
    auto put = [&](int c, float v) { if (x_bf16) xh[c] = __builtin_bit_cast(unsigned short, (__bf16)v); else xf[c] = v; };
    const int b0 = nf, b1 = b0 + 3, b2 = b1 + nf * pef, b3 = b2 + nf * pef, b4 = b3 + 3 * pev, b5 = b4 + 3 * pev;
    if (j < nf) {
        const float v = feat[(size_t)s * ldf + j];
        put(j, v);
        for (int p = 0; p < pef; ++p) {
            float sn, cs;
            sincosf(v * (float)(1 << p), &sn, &cs);
            put(b1 + j * pef + p, sn);
            put(b2 + j * pef + p, cs);
        }
    } else if (j < nf + 3) {
        const int a = j - nf;
        const float v = rays[(size_t)(act[s] / S) * 8 + 3 + a];
        put(b0 + a, v);
        for (int p = 0; p < pev; ++p) {
            float sn, cs;
            sincosf(v * (float)(1 << p), &sn, &cs);
            put(b3 + a * pev + p, sn);
            put(b4 + a * pev + p, cs);
        }
    } else {
        for (int c = b5; c < ldx; ++c) put(c, 0.f);       // alignment padding of the GEMM operand row stays zero
    }
}

extern "C" int clift_app_encode_fwd(const float* feat, int ldf, int nf, int pe_feat, int pe_view, const float* rays,
                                    const int* act_idx, int S, int M, float* X, int ldx, int x_bf16, clift_stream_t s) {
    CLIFT_REQUIRE(ldx >= nf + 3 + 2 * pe_feat * nf + 2 * pe_view * 3, "clift_app_encode_fwd: ldx %d too small", ldx);
    CLIFT_REQUIRE(pe_feat >= 1 && pe_view >= 1, "clift_app_encode_fwd: pe_feat/pe_view must be >= 1");
    if (M <= 0) return 0;
    const long total = (long)M * (nf + 4);
    k_app_encode_fwd<<<cdiv(total, 256), 256, 0, as_stream(s)>>>(feat, ldf, nf, pe_feat, pe_view, rays, act_idx, S, total, X, ldx, x_bf16);
    return clift_check_launch("clift_app_encode_fwd");
}

__global__ __launch_bounds__(256) void k_app_encode_bwd(const float* __restrict__ feat, int ldf, int nf, int pef,
                                                         const float* __restrict__ dX, int ldx, long total, float* __restrict__ dfeat,
                                                         int lddf) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int s = (int)(gid / lddf), c = (int)(gid - (long)s * lddf);
    if (rows_cut(s)) return;
    float v = 0.f;
    if (c < nf) {
        const float* g = dX + (size_t)s * ldx;
        const float x = feat[(size_t)s * ldf + c];
        v = g[c];
        const int bs = nf + 3, bc = bs + nf * pef;
        for (int p = 0; p < pef; ++p) {
            const float fr = (float)(1 << p);
            float sn, cs;
            sincosf(x * fr, &sn, &cs);
            v += fr * (cs * g[bs + c * pef + p] - sn * g[bc + c * pef + p]);
        }
    }
    dfeat[gid] = v;
}

extern "C" int clift_app_encode_bwd(const float* feat, int ldf, int nf, int pe_feat, const float* dX, int ldx, int M, float* dfeat,
                                    int lddf, clift_stream_t s) {
    CLIFT_REQUIRE(lddf >= nf, "clift_app_encode_bwd: lddf too small");
    if (M <= 0) return 0;
    const long total = (long)M * lddf;
    k_app_encode_bwd<<<cdiv(total, 256), 256, 0, as_stream(s)>>>(feat, ldf, nf, pe_feat, dX, ldx, total, dfeat, lddf);
    return clift_check_launch("clift_app_encode_bwd");
}

// ============================================================================ appearance front end, forward, ONE launch
// gather (plane x line products, a9) -> basis Linear(3 comps -> nf, no bias; tensoRF.py:65,127-134) -> MLP input row [feat | dir | PE] (a10
// input assembly) for a tile of AF_TS = 64 consecutive active samples, the products and the features never leaving the CU.  512 threads:
//   phase 0:  64 threads set up the tile's samples (ray slab, z, normalised position) once -- the unfused gather redid that, six divisions and
//             all, in each of the 36 threads of a sample; positions + view directions go to LDS, xa to memory;
//   phase 0b: thread = (plane, sample): the tap geometry ONCE per pair -- four texel offsets and bilinear weights, two line offsets and weights,
//             a 48-byte record in LDS.  (Timing probes of the first form of this kernel, AF_ABL: the per-thread tap arithmetic, repeated in the 12
//             channel groups of a pair, was 75 of its 190 us -- it is what bounds the unfused gather, too);
//   phase 1:  thread = (plane, sample, 4-channel group), plane wave-uniform: record + six 16-byte table reads -> product into the F tile in LDS
//             (and to memory only when the caller wants F).  64 consecutive samples of the compacted order are one or two rays: their taps
//             overlap, so most of these reads hit the L1;
//   phase 2:  feat = F Wb^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: rows = 32 samples, columns = the <= 28 features padded to 32),
//             waves 0..3 = (sample half, k half); a lane's 36 weights live in registers (loaded at kernel start), its F values come as
//             ds_read_b128 (row stride 4 x odd: conflict-free); the two k halves meet in LDS (upper half stores, barrier, lower half adds);
//   phase 3:  thread = (sample, source element): one sincosf per (element, frequency) as in k_app_encode_fwd, the row assembled in LDS (over the F
//             tile, which is dead by then; odd row stride: a wave writes one column of 64 rows);
//   phase 4:  the tile's rows of X are ONE contiguous run of memory: a wave stores 256 consecutive bytes per instruction.
// Against the three launches it replaces (gather 96 us, 144 -> 27 GEMM 60 us, encode 75 us at 287 k samples) it neither writes and re-reads the
// products (2 x 165 MB) nor re-reads the features.
#ifndef AF_ABL
#define AF_ABL 0           // timing probes (tools/app_probe.sh): bit 0 no gather loads, 1 no contraction, 2 no sincos, 3 no X store (results garbage)
#endif
constexpr int AF_TS = 64;
constexpr int AF_FT = 29;              // floats between the feature rows of the tile (odd: a wave's column accesses are conflict-free)
constexpr int AF_NF = 28;              // feature columns in memory (ldf) = the most features the kernel takes
struct alignas(16) AfRec {
    int o[4];              // texel offsets (floats) into the channels-last plane: corners (x0,y0) (x1,y0) (x0,y1) (x1,y1), clamped
    float w[4];            // bilinear weights (0 where the tap is out of range)
    int z[2];              // line entries
    float wz[2];
};

template <int COMPS>
__global__ __launch_bounds__(512) void k_app_front_fwd(MarchP m, VmP t, const float* __restrict__ rays, const float* __restrict__ jitter,
                                                        const int* __restrict__ act, int M, const float* __restrict__ Wb, int ldb, int nf,
                                                        int pef, int pev, float* __restrict__ xa, float* __restrict__ feat,
                                                        float* __restrict__ X, int ldx, float* __restrict__ F, int x_bf16) {
    constexpr int C = COMPS, NC = 3 * C, G4 = C / 4, FS = 4 * ((NC / 4) | 1), KS = NC / 4;      // KS: MFMA steps (= weights) per lane
    static_assert(C % 16 == 0 && KS % 4 == 0, "the k quarters of phase 2 are read 16 bytes at a time");
    extern __shared__ __attribute__((aligned(16))) float af_lds[];
    float* const pos = af_lds;                                   // [64][8]: xn.xyz, valid flag, dir.xyz, 0
    AfRec* const recs = reinterpret_cast<AfRec*>(pos + AF_TS * 8);   // [3][64]; dead after phase 1:
    float* const ft = pos + AF_TS * 8;                           // feature tile [64][AF_FT] (phase 2 on)
    float* const big = ft + AF_TS * 3 * (int)(sizeof(AfRec) / 4);    // F tile [64][FS], later the X tile [64][ldx | 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    M = limit_rows(M);
    const int s0 = blockIdx.x * AF_TS;
    if (s0 >= M) return;
    // (waves 0..3) this lane's KS basis weights: row j = li, k = (NC / 2) (wave >> 1) + KS lh + 0 .. KS-1
    float4 bw[KS / 4];
    const int li = lane & 31, lh = lane >> 5;
    const int kbase = (NC / 2) * (wave >> 1) + KS * lh;
    if (wave < 4) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q)
            bw[q] = (li < nf && !(AF_ABL & 2)) ? *reinterpret_cast<const float4*>(Wb + (size_t)li * ldb + kbase + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---------------- phase 0
    if (tid < AF_TS) {
        const int s = s0 + tid;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < M) {
            const int sid = act[s];
            const int r = sid / m.S, k = sid - r * m.S;
            const RayG g = load_ray(rays, r, m);
            float xn[3];
            sample_xn(g, m, sample_z(g, m, k, jitter ? jitter[r] : 0.f), xn);
            p = make_float4(xn[0], xn[1], xn[2], 1.f);
            d = make_float4(g.d[0], g.d[1], g.d[2], 0.f);
            if (xa) *reinterpret_cast<float4*>(xa + (size_t)s * 4) = make_float4(xn[0], xn[1], xn[2], 0.f);
        }
        *reinterpret_cast<float4*>(pos + tid * 8) = p;
        *reinterpret_cast<float4*>(pos + tid * 8 + 4) = d;
    }
    __syncthreads();
    // ---------------- phase 0b
    if (tid < 3 * AF_TS) {
        const int i = wave, sl = lane;
        const float4 p = *reinterpret_cast<const float4*>(pos + sl * 8);
        const float xn[3] = {p.x, p.y, p.z};
        const VmTaps tp = vm_taps(t, i, xn);
        int a_, b_, v_;
        vm_axes(i, a_, b_, v_);
        const int W = t.res[a_];
        int4* dst = reinterpret_cast<int4*>(recs + i * AF_TS + sl);
        dst[0] = make_int4((tp.ty.i0 * W + tp.tx.i0) * C, (tp.ty.i0 * W + tp.tx.i1) * C, (tp.ty.i1 * W + tp.tx.i0) * C, (tp.ty.i1 * W + tp.tx.i1) * C);
        dst[1] = make_int4(__float_as_int(tp.tx.w0 * tp.ty.w0), __float_as_int(tp.tx.w1 * tp.ty.w0), __float_as_int(tp.tx.w0 * tp.ty.w1),
                           __float_as_int(tp.tx.w1 * tp.ty.w1));
        dst[2] = make_int4(tp.tz.i0 * C, tp.tz.i1 * C, __float_as_int(tp.tz.w0), __float_as_int(tp.tz.w1));
    }
    __syncthreads();
    // ---------------- phase 1: items in plane-major order (64 x G4 = a whole number of waves per plane: the plane is wave-uniform)
    constexpr int PER_PLANE = AF_TS * G4;
    for (int e = tid; e < 3 * PER_PLANE; e += 512) {
        const int i = __builtin_amdgcn_readfirstlane(e / PER_PLANE);
        const int r_ = e - i * PER_PLANE, sl = r_ / G4, c4 = (r_ - sl * G4) * 4;
        const int4* rp = reinterpret_cast<const int4*>(recs + i * AF_TS + sl);
        const int4 o = rp[0], wi = rp[1], zi = rp[2];
        const float* pp = t.plane[i] + c4;
        const float* lp = t.line[i] + c4;
        float4 v;
        if (AF_ABL & 1) {
            v = make_float4(__int_as_float(wi.x), __int_as_float(wi.y), __int_as_float(wi.z), __int_as_float(zi.z));
        } else {
            float4 acc = f4_scale(__int_as_float(wi.x), ld4(pp + (unsigned)o.x));
            acc = f4_fma(__int_as_float(wi.y), ld4(pp + (unsigned)o.y), acc);
            acc = f4_fma(__int_as_float(wi.z), ld4(pp + (unsigned)o.z), acc);
            acc = f4_fma(__int_as_float(wi.w), ld4(pp + (unsigned)o.w), acc);
            float4 ln = f4_scale(__int_as_float(zi.z), ld4(lp + (unsigned)zi.x));
            ln = f4_fma(__int_as_float(zi.w), ld4(lp + (unsigned)zi.y), ln);
            v = f4_mul(acc, ln);
        }
        *reinterpret_cast<float4*>(big + sl * FS + i * C + c4) = v;
        if (F && s0 + sl < M) *reinterpret_cast<float4*>(F + (size_t)(s0 + sl) * NC + i * C + c4) = v;
    }
    __syncthreads();
    // ---------------- phase 2
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (wave < 4) {
        const float* frow = big + (32 * (wave & 1) + li) * FS + kbase;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 f = *reinterpret_cast<const float4*>(frow + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, bw[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, bw[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, bw[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, bw[q].w, acc, 0, 0, 0);
        }
    }
    // acc[r]: sample 32 (wave & 1) + 8 (r >> 2) + 4 lh + (r & 3), feature li.  (the records under `ft` are dead: every wave passed the barrier above)
    if (wave >= 2 && wave < 4 && li < AF_NF) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ft[(32 * (wave & 1) + 8 * (r >> 2) + 4 * lh + (r & 3)) * AF_FT + li] = acc[r];
    }
    __syncthreads();
    if (wave < 2 && li < AF_NF) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* q = ft + (32 * (wave & 1) + 8 * (r >> 2) + 4 * lh + (r & 3)) * AF_FT + li;
            *q = acc[r] + *q;
        }
    }
    __syncthreads();
    // ---------------- phase 3 (the X tile overwrites the F tile)
    const int rows = min(AF_TS, M - s0);
    for (int e = tid; e < rows * AF_NF; e += 512) {                     // the features themselves (the encode backward reads them), pad columns zero
        const int sl = e / AF_NF, j = e - sl * AF_NF;
        feat[(size_t)s0 * AF_NF + e] = j < nf ? ft[sl * AF_FT + j] : 0.f;
    }
    const int XS = ldx | 1;
    {
        const int sl = lane;
        float* xr = big + sl * XS;
        const int J = nf + 3;
        const int b0 = nf, b1 = b0 + 3, b2 = b1 + nf * pef, b3 = b2 + nf * pef, b4 = b3 + 3 * pev, b5 = b4 + 3 * pev;
        for (int j = wave; j < J; j += 8) {                              // (wave-uniform j: no divergence between the feature and direction forms)
            if (j < nf) {
                const float v = ft[sl * AF_FT + j];
                xr[j] = v;
                for (int p = 0; p < pef; ++p) {
                    float sn, cs;
                    if (AF_ABL & 4) { sn = v; cs = v + 1.f; } else
                    sincosf(v * (float)(1 << p), &sn, &cs);
                    xr[b1 + j * pef + p] = sn;
                    xr[b2 + j * pef + p] = cs;
                }
            } else {
                const int a = j - nf;
                const float v = pos[sl * 8 + 4 + a];
                xr[b0 + a] = v;
                for (int p = 0; p < pev; ++p) {
                    float sn, cs;
                    if (AF_ABL & 4) { sn = v; cs = v + 1.f; } else
                    sincosf(v * (float)(1 << p), &sn, &cs);
                    xr[b3 + a * pev + p] = sn;
                    xr[b4 + a * pev + p] = cs;
                }
            }
        }
        for (int c = b5 + wave; c < ldx; c += 8) xr[c] = 0.f;           // alignment padding of the GEMM operand row stays zero
    }
    __syncthreads();
    // ---------------- phase 4
    if (x_bf16) {           // bf16 mode: the MLP input is bf16-STORED (ldx even: two columns per lane)
        for (int r = wave; r < rows; r += 8) {
            unsigned* dst = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(X) + (size_t)(s0 + r) * ldx);
            const float* src = big + r * XS;
            for (int c = lane; 2 * c < ldx; c += 64)
                dst[c] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)src[2 * c]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)src[2 * c + 1]) << 16);
        }
        return;
    }
    for (int r = wave; r < ((AF_ABL & 8) ? 1 : rows); r += 8) {
        float* dst = X + (size_t)(s0 + r) * ldx;
        const float* src = big + r * XS;
        for (int c = lane; c < ldx; c += 64) dst[c] = src[c];
    }
}

extern "C" int clift_app_front_fwd(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter, const int* act_idx,
                                   int M, const float* Wb, int ldb, int nf, int pe_feat, int pe_view, float* xa, float* feat, int ldf, float* X,
                                   int ldx, float* F, clift_stream_t s) {
    return clift_app_front_fwd_x(h_m, h_app, rays, jitter, act_idx, M, Wb, ldb, nf, pe_feat, pe_view, xa, feat, ldf, X, ldx, F, 0, s);
}

// ... with the choice of X's storage: x_bf16 = 1 writes X as (M, ldx) bf16 (bf16 mode, ABI 17); everything else as above.
extern "C" int clift_app_front_fwd_x(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter, const int* act_idx,
                                     int M, const float* Wb, int ldb, int nf, int pe_feat, int pe_view, float* xa, float* feat, int ldf, void* Xv,
                                     int ldx, float* F, int x_bf16, clift_stream_t s) {
    float* X = reinterpret_cast<float*>(Xv);
    const int C = h_app->comps, nc = 3 * C;
    CLIFT_REQUIRE(C == 16 || C == 32 || C == 48, "clift_app_front_fwd: comps must be 16, 32 or 48 (got %d)", C);
    CLIFT_REQUIRE(nf >= 1 && nf <= AF_NF && ldf == AF_NF, "clift_app_front_fwd: 1 <= n_features <= %d, ldf == %d", AF_NF, AF_NF);
    CLIFT_REQUIRE(pe_feat >= 1 && pe_view >= 1, "clift_app_front_fwd: pe_feat/pe_view must be >= 1");
    CLIFT_REQUIRE(ldx % 4 == 0 && ldx >= nf + 3 + 2 * pe_feat * nf + 2 * pe_view * 3, "clift_app_front_fwd: ldx %d too small or not a multiple of 4", ldx);
    CLIFT_REQUIRE(ldb % 4 == 0 && ldb >= nc && (((uintptr_t)Wb) & 15) == 0, "clift_app_front_fwd: the basis matrix needs 16-byte aligned rows of >= 3 * comps floats");
    if (M <= 0) return 0;
    const int fs = 4 * ((nc / 4) | 1), xs = ldx | 1;
    const int dyn = (AF_TS * 8 + AF_TS * 3 * (int)(sizeof(AfRec) / 4) + AF_TS * (fs > xs ? fs : xs)) * 4;
    CLIFT_REQUIRE(dyn <= 160 * 1024 - 512, "clift_app_front_fwd: tile does not fit in LDS (ldx %d)", ldx);
    const dim3 grid(cdiv(M, AF_TS));
    hipStream_t st = as_stream(s);
#define CLIFT_AF(CC)                                                                                                                            \
    do {                                                                                                                                        \
        if (dyn > 48 * 1024)                                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_app_front_fwd<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);    \
        k_app_front_fwd<CC><<<grid, 512, dyn, st>>>(to_dev(h_m), to_dev(h_app), rays, jitter, act_idx, M, Wb, ldb, nf, pe_feat, pe_view, xa,  \
                                                    feat, X, ldx, F, x_bf16);                                                                  \
    } while (0)
    if (C == 48) CLIFT_AF(48);
    else if (C == 32) CLIFT_AF(32);
    else CLIFT_AF(16);
#undef CLIFT_AF
    return clift_check_launch("clift_app_front_fwd");
}

// ============================================================================ compositing forward
// Thread = (ray, output channel); channel ranges [0,3) rgb, [3,3+C) semantics, [3+C,3+C+D) instances.
// Each thread walks the ray's contiguous slice of the compacted sample list (same summation order as a
// sequential sum over samples).
__global__ __launch_bounds__(256) void k_composite_sum(const float* __restrict__ w, const int* __restrict__ start, const int* __restrict__ act,
                                                        int N, int C, int D, const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                        const float* __restrict__ inst_s, float* __restrict__ rgb_raw,
                                                        float* __restrict__ sem_raw, float* __restrict__ inst_map) {
    const int CH = 3 + C + D;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)N * CH) return;
    const int r = (int)(gid / CH), c = (int)(gid - (long)r * CH);
    const float* src;
    int ld, cc;
    float* dst;
    if (c < 3) { src = rgb_s; ld = 3; cc = c; dst = rgb_raw ? rgb_raw + (size_t)r * 3 + c : nullptr; }
    else if (c < 3 + C) { src = sem_s; ld = C; cc = c - 3; dst = sem_raw ? sem_raw + (size_t)r * C + cc : nullptr; }
    else { src = inst_s; ld = D; cc = c - 3 - C; dst = inst_map ? inst_map + (size_t)r * D + cc : nullptr; }
    if (!src || !dst) return;
    // the sum stays sequential (one fmaf chain in sample order), but eight samples' loads are in flight together: the walk is a chain of
    // dependent loads (act -> w) ~60 long per ray, and one at a time it took 35 us for 1024 rays (41 us for 4096: latency, not work)
    float acc = 0.f;
    int i = start[r];
    const int e = start[r + 1];
    for (; i + 8 <= e; i += 8) {
        int a[8];
        float wv[8], sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = act[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { wv[u] = w[a[u]]; sv[u] = src[(size_t)(i + u) * ld + cc]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(wv[u], sv[u], acc);
    }
    for (; i < e; ++i) acc = fmaf(w[act[i]], src[(size_t)i * ld + cc], acc);
    *dst = acc;
}

__global__ __launch_bounds__(256) void k_composite_finish(int N, int C, const float* __restrict__ ray_out, int softmax_mode, int white_bg,
                                                           float* __restrict__ rgb_raw, float* __restrict__ rgb_map,
                                                           const float* __restrict__ sem_raw, float* __restrict__ sem_map) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    if (rgb_raw && rgb_map) {
        const float add = white_bg ? (1.f - ray_out[(size_t)r * 8]) : 0.f;
        for (int c = 0; c < 3; ++c) {
            const float v = rgb_raw[(size_t)r * 3 + c] + add;
            rgb_raw[(size_t)r * 3 + c] = v;  // pre-clamp value, kept for the clamp mask of the backward
            rgb_map[(size_t)r * 3 + c] = fminf(fmaxf(v, 0.f), 1.f);
        }
    }
    if (sem_raw && sem_map) {
        const float* sr = sem_raw + (size_t)r * C;
        float* sm = sem_map + (size_t)r * C;
        if (softmax_mode) {
            float t = 0.f;
            for (int c = 0; c < C; ++c) t += sr[c];
            t += 1e-8f;
            for (int c = 0; c < C; ++c) sm[c] = logf(sr[c] / t + 1e-8f);
        } else {
            for (int c = 0; c < C; ++c) sm[c] = sr[c];
        }
    }
}

extern "C" int clift_composite_fwd(const float* w, const int* ray_start, const int* act_idx, int N, int C, int D, const float* rgb_s,
                                   const float* sem_s, const float* inst_s, const float* ray_out, int softmax_mode, int white_bg,
                                   float* rgb_raw, float* rgb_map, float* sem_raw, float* sem_map, float* inst_map, clift_stream_t s) {
    if (N <= 0) return 0;
    const long total = (long)N * (3 + C + D);
    k_composite_sum<<<cdiv(total, 256), 256, 0, as_stream(s)>>>(w, ray_start, act_idx, N, C, D, rgb_s, sem_s, inst_s, rgb_raw, sem_raw, inst_map);
    int rc = clift_check_launch("clift_composite_fwd(sum)");
    if (rc) return rc;
    // (a chunk without a single active sample passes NULL heads: its sums are the caller's zeros and are finished all the same --
    // white background, clamp, log-normalisation of an all-zero semantic sum, renderer.py:160-167)
    k_composite_finish<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(N, C, ray_out, softmax_mode, white_bg, rgb_raw, rgb_map, C > 0 ? sem_raw : nullptr, sem_map);
    return clift_check_launch("clift_composite_fwd(finish)");
}

// ============================================================================ compositing backward
// Stage 1 (per ray): fold clamp / white background / log-normalisation into effective per-ray gradients
// ge (N, 3+C+D) and g_opacity.  Stage 2 (per active sample): d head outputs = w * ge, g_w = <head output, ge>.
__global__ __launch_bounds__(256) void k_composite_bwd_ray(int N, int C, int D, const float* __restrict__ rgb_raw, const float* __restrict__ sem_raw,
                                                            int softmax_mode, int white_bg, const float* __restrict__ g_rgb,
                                                            const float* __restrict__ g_sem, const float* __restrict__ g_inst,
                                                            float* __restrict__ ge, float* __restrict__ g_opacity) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const int CH = 3 + C + D;
    float* e = ge + (size_t)r * CH;
    float gop = 0.f;
    for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (g_rgb && rgb_raw) {
            const float raw = rgb_raw[(size_t)r * 3 + c];
            v = (raw >= 0.f && raw <= 1.f) ? g_rgb[(size_t)r * 3 + c] : 0.f;
            if (white_bg) gop -= v;
        }
        e[c] = v;
    }
    if (g_sem && sem_raw) {
        const float* sr = sem_raw + (size_t)r * C;
        const float* gs = g_sem + (size_t)r * C;
        if (softmax_mode) {
            float t = 0.f;
            for (int c = 0; c < C; ++c) t += sr[c];
            t += 1e-8f;
            float dot = 0.f;
            for (int c = 0; c < C; ++c) dot += (gs[c] / (sr[c] / t + 1e-8f)) * sr[c];
            for (int c = 0; c < C; ++c) e[3 + c] = (gs[c] / (sr[c] / t + 1e-8f)) / t - dot / (t * t);
        } else {
            for (int c = 0; c < C; ++c) e[3 + c] = gs[c];
        }
    } else {
        for (int c = 0; c < C; ++c) e[3 + c] = 0.f;
    }
    for (int c = 0; c < D; ++c) e[3 + C + c] = g_inst ? g_inst[(size_t)r * D + c] : 0.f;
    if (g_opacity) g_opacity[r] = gop;
}

__global__ __launch_bounds__(256) void k_composite_bwd_sample(const float* __restrict__ w, const int* __restrict__ act, int S, int M, int C, int D,
                                                               const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                               const float* __restrict__ inst_s, const float* __restrict__ ge, int stop_grad,
                                                               float* __restrict__ d_rgb_s, float* __restrict__ d_sem_s,
                                                               float* __restrict__ d_inst_s, float* __restrict__ g_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= limit_rows(M)) return;
    const int sid = act[i];
    const int r = sid / S;
    const float wv = w[sid];
    const float* e = ge + (size_t)r * (3 + C + D);
    float gw = 0.f;
    if (rgb_s) {
        for (int c = 0; c < 3; ++c) {
            if (d_rgb_s) d_rgb_s[(size_t)i * 3 + c] = wv * e[c];
            gw = fmaf(rgb_s[(size_t)i * 3 + c], e[c], gw);
        }
    }
    if (sem_s) {
        for (int c = 0; c < C; ++c) {
            if (d_sem_s) d_sem_s[(size_t)i * C + c] = wv * e[3 + c];
            if (!stop_grad) gw = fmaf(sem_s[(size_t)i * C + c], e[3 + c], gw);
        }
    }
    if (inst_s) {
        for (int c = 0; c < D; ++c) {
            if (d_inst_s) d_inst_s[(size_t)i * D + c] = wv * e[3 + C + c];
            if (!stop_grad) gw = fmaf(inst_s[(size_t)i * D + c], e[3 + C + c], gw);
        }
    }
    if (g_w) g_w[sid] = gw;
}

// Stage 2 with the heads' output activations folded in (ABI 16): what leaves is the gradient w.r.t. the PRE-activation outputs of the last
// layers, in the zero-padded rows their backward kernels take (row pitches ld_*, multiples of 4) -- the separate clift_rows_act_bwd launches (one
// per head: a read of the head output and of these gradients, a write of the same rows again) are gone.  rgb: sigmoid (d o (1 - o)); semantics:
// sem_kind 2 = softmax over the row (o (g - <g, o>), the dot product summed in class order), 0 = identity; instances: identity, columns
// [0, E) to dpre_i0 and [E, 2 E) to dpre_i1 (each nullable: the slow half is detached in training, T:268).  A thread owns a sample's whole row.
__global__ __launch_bounds__(256) void k_composite_bwd_sample_act(const float* __restrict__ w, const int* __restrict__ act, int S, int M, int C, int D,
                                                                   const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                                   const float* __restrict__ inst_s, const float* __restrict__ ge, int stop_grad,
                                                                   int sem_kind, float* __restrict__ dpre_rgb, int ld_rgb, float* __restrict__ dpre_sem,
                                                                   int ld_sem, float* __restrict__ dpre_i0, float* __restrict__ dpre_i1, int ld_inst,
                                                                   int E, float* __restrict__ g_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= limit_rows(M)) return;
    const int sid = act[i];
    const int r = sid / S;
    const float wv = w[sid];
    const float* e = ge + (size_t)r * (3 + C + D);
    float gw = 0.f;
    if (rgb_s) {
        float* dr = dpre_rgb ? dpre_rgb + (size_t)i * ld_rgb : nullptr;
        for (int c = 0; c < 3; ++c) {
            const float o = rgb_s[(size_t)i * 3 + c];
            if (dr) { const float g = wv * e[c]; dr[c] = g * o * (1.f - o); }
            gw = fmaf(o, e[c], gw);
        }
        if (dr) for (int c = 3; c < ld_rgb; ++c) dr[c] = 0.f;
    }
    if (sem_s) {
        const float* so = sem_s + (size_t)i * C;
        float* ds = dpre_sem ? dpre_sem + (size_t)i * ld_sem : nullptr;
        float dot = 0.f;
        if (ds && sem_kind == 2)
            for (int c = 0; c < C; ++c) dot += (wv * e[3 + c]) * so[c];
        for (int c = 0; c < C; ++c) {
            const float o = so[c];
            if (ds) { const float g = wv * e[3 + c]; ds[c] = sem_kind == 2 ? o * (g - dot) : g; }
            if (!stop_grad) gw = fmaf(o, e[3 + c], gw);
        }
        if (ds) for (int c = C; c < ld_sem; ++c) ds[c] = 0.f;
    }
    if (inst_s) {
        float* d0 = dpre_i0 ? dpre_i0 + (size_t)i * ld_inst : nullptr;
        float* d1 = dpre_i1 ? dpre_i1 + (size_t)i * ld_inst : nullptr;
        for (int c = 0; c < D; ++c) {
            const float g = wv * e[3 + C + c];
            if (c < E) { if (d0) d0[c] = g; }
            else if (c < 2 * E) { if (d1) d1[c - E] = g; }
            if (!stop_grad) gw = fmaf(inst_s[(size_t)i * D + c], e[3 + C + c], gw);
        }
        for (int c = E; c < ld_inst; ++c) { if (d0) d0[c] = 0.f; if (d1) d1[c] = 0.f; }
    }
    if (g_w) g_w[sid] = gw;
}

// The same, SIXTEEN lanes per sample (C <= 48, D <= 8): lane 0 = the colours, lanes 1 .. 12 = four semantic classes each, lanes 13 / 14 = four
// instance dimensions each; the row's two sums (the softmax dot product, g_w) fold over the 16 lanes with xor shuffles.  A wave covers four
// consecutive samples: its loads and stores are runs of whole rows.  (The one-thread-per-sample form above walks a 100-byte row per lane, every
// access of a wave 44 cache lines wide: 63 us per launch where this data moves in ~15.)
__global__ __launch_bounds__(256) void k_composite_bwd_sample_act16(const float* __restrict__ w, const int* __restrict__ act, int S, int M, int C, int D,
                                                                     const float* __restrict__ rgb_s, const float* __restrict__ sem_s,
                                                                     const float* __restrict__ inst_s, const float* __restrict__ ge, int stop_grad,
                                                                     int sem_kind, float* __restrict__ dpre_rgb, int ld_rgb, float* __restrict__ dpre_sem,
                                                                     int ld_sem, float* __restrict__ dpre_i0, float* __restrict__ dpre_i1, int ld_inst,
                                                                     int E, float* __restrict__ g_w) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(gid >> 4), l = (int)(gid & 15);
    const bool row = i < limit_rows(M);              // (whole 16-lane groups: the shuffles below stay inside a group)
    const int ii = row ? i : 0;
    const int sid = act[ii];
    const int r = sid / S;
    const float wv = w[sid];
    const float* e = ge + (size_t)r * (3 + C + D);
    float gw = 0.f, dot = 0.f;
    float o[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
    int c0 = 0, n = 0;                                // this lane's channels [c0, c0 + n) of its head
    if (l == 0) {
        if (rgb_s) {
            n = 3;
            for (int k = 0; k < 3; ++k) { o[k] = rgb_s[(size_t)ii * 3 + k]; g[k] = wv * e[k]; gw = fmaf(o[k], e[k], gw); }
        }
    } else if (l <= 12) {
        c0 = 4 * (l - 1);
        if (sem_s && c0 < C) {
            n = min(4, C - c0);
            for (int k = 0; k < n; ++k) {
                o[k] = sem_s[(size_t)ii * C + c0 + k]; g[k] = wv * e[3 + c0 + k];
                dot += g[k] * o[k];
                if (!stop_grad) gw = fmaf(o[k], e[3 + c0 + k], gw);
            }
        }
    } else if (l <= 14) {
        c0 = 4 * (l - 13);
        if (inst_s && c0 < D) {
            n = min(4, D - c0);
            for (int k = 0; k < n; ++k) {
                g[k] = wv * e[3 + C + c0 + k];
                if (!stop_grad) gw = fmaf(inst_s[(size_t)ii * D + c0 + k], e[3 + C + c0 + k], gw);
            }
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) { gw += __shfl_xor(gw, d); dot += __shfl_xor(dot, d); }
    if (!row) return;
    if (l == 0) {
        if (dpre_rgb && rgb_s) {
            float* dr = dpre_rgb + (size_t)i * ld_rgb;
            for (int k = 0; k < ld_rgb; ++k) dr[k] = k < 3 ? g[k] * o[k] * (1.f - o[k]) : 0.f;
        }
        if (g_w) g_w[sid] = gw;
    } else if (l <= 12) {
        if (dpre_sem && sem_s && c0 < ld_sem) {
            float* ds = dpre_sem + (size_t)i * ld_sem + c0;
            for (int k = 0; k < min(4, ld_sem - c0); ++k) ds[k] = k < n ? (sem_kind == 2 ? o[k] * (g[k] - dot) : g[k]) : 0.f;
        }
    } else if (l <= 14) {
        if (inst_s)
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;                 // instance column: [0, E) -> dpre_i0, [E, 2E) -> dpre_i1, pads zero
                if (c < E) { if (dpre_i0) dpre_i0[(size_t)i * ld_inst + c] = g[k]; }
                else if (c < 2 * E && c < D) { if (dpre_i1) dpre_i1[(size_t)i * ld_inst + c - E] = g[k]; }
            }
        if (l == 13 && inst_s)
            for (int c = E; c < ld_inst; ++c) { if (dpre_i0) dpre_i0[(size_t)i * ld_inst + c] = 0.f; if (dpre_i1) dpre_i1[(size_t)i * ld_inst + c] = 0.f; }
    }
}

extern "C" int clift_composite_bwd_act(const float* w, const int* act_idx, int N, int S, int M, int C, int D, const float* rgb_s, const float* sem_s,
                                       const float* inst_s, const float* rgb_raw, const float* sem_raw, int softmax_mode, int white_bg, int stop_grad,
                                       const float* g_rgb, const float* g_sem, const float* g_inst, float* ge_work, int sem_kind, float* dpre_rgb,
                                       int ld_rgb, float* dpre_sem, int ld_sem, float* dpre_i0, float* dpre_i1, int ld_inst, int E, float* g_w,
                                       float* g_opacity, clift_stream_t s) {
    CLIFT_REQUIRE(sem_kind == 0 || sem_kind == 2, "clift_composite_bwd_act: sem_kind must be 0 (identity) or 2 (softmax)");
    CLIFT_REQUIRE((!dpre_rgb || ld_rgb >= 3) && (!dpre_sem || ld_sem >= C) && ((!dpre_i0 && !dpre_i1) || (ld_inst >= E && E >= 1 && 2 * E >= D) || D == 0),
                  "clift_composite_bwd_act: row pitches smaller than the rows (ld_rgb %d, ld_sem %d / C %d, ld_inst %d / E %d / D %d)", ld_rgb, ld_sem, C,
                  ld_inst, E, D);
    if (N <= 0) return 0;
    k_composite_bwd_ray<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(N, C, D, rgb_raw, sem_raw, softmax_mode, white_bg, g_rgb, g_sem, g_inst,
                                                                 ge_work, g_opacity);
    int rc = clift_check_launch("clift_composite_bwd_act(ray)");
    if (rc || M <= 0) return rc;
    if (C <= 48 && D <= 8 && ld_sem <= 48)
        k_composite_bwd_sample_act16<<<cdiv(16L * M, 256), 256, 0, as_stream(s)>>>(w, act_idx, S, M, C, D, rgb_s, sem_s, inst_s, ge_work, stop_grad, sem_kind,
                                                                                    dpre_rgb, ld_rgb, dpre_sem, ld_sem, dpre_i0, dpre_i1, ld_inst, E, g_w);
    else
        k_composite_bwd_sample_act<<<cdiv(M, 256), 256, 0, as_stream(s)>>>(w, act_idx, S, M, C, D, rgb_s, sem_s, inst_s, ge_work, stop_grad, sem_kind, dpre_rgb,
                                                                            ld_rgb, dpre_sem, ld_sem, dpre_i0, dpre_i1, ld_inst, E, g_w);
    return clift_check_launch("clift_composite_bwd_act(sample)");
}

extern "C" int clift_composite_bwd(const float* w, const int* ray_start, const int* act_idx, int N, int S, int M, int C, int D,
                                   const float* rgb_s, const float* sem_s, const float* inst_s, const float* rgb_raw,
                                   const float* sem_raw, int softmax_mode, int white_bg, int stop_grad, const float* g_rgb,
                                   const float* g_sem, const float* g_inst, float* ge_work, float* d_rgb_s, float* d_sem_s,
                                   float* d_inst_s, float* g_w, float* g_opacity, clift_stream_t s) {
    (void)ray_start;
    if (N <= 0) return 0;
    k_composite_bwd_ray<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(N, C, D, rgb_raw, sem_raw, softmax_mode, white_bg, g_rgb, g_sem, g_inst,
                                                                 ge_work, g_opacity);
    int rc = clift_check_launch("clift_composite_bwd(ray)");
    if (rc || M <= 0) return rc;
    k_composite_bwd_sample<<<cdiv(M, 256), 256, 0, as_stream(s)>>>(w, act_idx, S, M, C, D, rgb_s, sem_s, inst_s, ge_work, stop_grad, d_rgb_s,
                                                                    d_sem_s, d_inst_s, g_w);
    return clift_check_launch("clift_composite_bwd(sample)");
}
