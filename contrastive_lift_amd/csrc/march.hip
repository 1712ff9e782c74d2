// march.hip -- ray marching: density VM lookup, transmittance scan, compaction, and their backward.
// Reference rows: a4 a5 a6 a7 a8.
#include "clift_dev.h"
#include <type_traits>
#include <string.h>
#include <stdlib.h>
CLIFT_ROWS_LIMIT_BINDER(march)

// ============================================================================ density forward
// 4 lanes per sample, lane q owns channels [4q, 4q+4) (+16 per extra pass) of all three plane/line pairs:
// a tap of one sample is one contiguous 64-byte segment of the channels-last table, read by 4 adjacent
// lanes.  Samples are consecutive along a ray, so a wave (16 samples) walks a short 3-D segment and its
// taps mostly share cache lines.
__global__ __launch_bounds__(256) void k_density_fwd(MarchP m, VmP t, const float* __restrict__ rays,
                                                      const float* __restrict__ jitter, long total, float* __restrict__ sigma) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = gid >> 2;
    const int q = (int)(gid & 3);
    if (s >= total) return;
    const int r = (int)(s / m.S), k = (int)(s - (long)r * m.S);
    const RayG g = load_ray(rays, r, m);
    const float jit = jitter ? jitter[r] : 0.f;
    float xn[3];
    const bool in = sample_xn(g, m, sample_z(g, m, k, jit), xn);
    float acc = 0.f;
    if (in) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const VmTaps tp = vm_taps(t, i, xn);
            for (int c4 = q * 4; c4 < t.comps; c4 += 16) acc += f4_hsum(f4_mul(vm_plane4(t, i, tp, c4), vm_line4(t, i, tp, c4)));
        }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) {
        float out = 0.f;
        if (in) {
            const float x = acc + m.shift;
            out = (x > 20.f) ? x : log1pf(expf(x));
        }
        sigma[s] = out;
    }
}

// The same values (bit for bit: same taps, same FMA order, same quad sum) from ONE WAVE PER SWEEP of 64 consecutive samples of a ray.  The per-thread form above is bound by its
// index arithmetic, not by the table reads: every one of the four lanes of a sample redoes the ray's slab set-up (six IEEE divisions) and the tap
// geometry of all three planes, and samples behind the box exit pay for the set-up too.  Here the ray is set up once per lane and sweep; lane =
// sample computes position and tap records (four texel offsets + weights, two line offsets + weights per plane: 48
// bytes) into the wave's LDS slot, then four passes of 16 samples x 4 lanes read the records back (broadcast inside a quad) and do nothing but
// the 18 table reads and their FMAs.  In-box samples of a ray are a prefix (the march starts at the slab entry): a sweep with no sample in the
// box stores zeros and ends.  56 -> see profiles/r04_density_fwd.txt.
struct alignas(16) DfRec {
    int o[4];
    float w[4];
    int z[2];
    float wz[2];
};
constexpr int DFR_WAVES = 4;

__global__ __launch_bounds__(64 * DFR_WAVES) void k_density_fwd_ray(MarchP m, VmP t, const float* __restrict__ rays, const float* __restrict__ jitter, int N,
                                                                    float* __restrict__ sigma) {
    __shared__ DfRec recs_all[DFR_WAVES][3][64];
    __shared__ float sig_all[DFR_WAVES][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsw = (m.S + 63) / 64;                         // sweeps per ray
    const long item = (long)blockIdx.x * DFR_WAVES + wave;
    if (item >= (long)N * nsw) return;
    const int r = (int)(item / nsw);
    DfRec (*recs)[64] = recs_all[wave];
    float* sig = sig_all[wave];
    const RayG g = load_ray(rays, r, m);
    const float jit = jitter ? jitter[r] : 0.f;
    const int C = t.comps;
    float* out = sigma + (size_t)r * m.S;
    {
        const int k0 = (int)(item - (long)r * nsw) * 64;
        const int k = k0 + lane;
        float xn[3];
        const bool in = k < m.S && sample_xn(g, m, sample_z(g, m, k, jit), xn);
        if (__ballot(in) == 0) {                          // wave-uniform
            if (k < m.S) out[k] = 0.f;
            return;
        }
        if (in) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const VmTaps tp = vm_taps(t, i, xn);
                int a_, b_, v_;
                vm_axes(i, a_, b_, v_);
                const int W = t.res[a_];
                int4* dst = reinterpret_cast<int4*>(&recs[i][lane]);
                dst[0] = make_int4((tp.ty.i0 * W + tp.tx.i0) * C, (tp.ty.i0 * W + tp.tx.i1) * C, (tp.ty.i1 * W + tp.tx.i0) * C, (tp.ty.i1 * W + tp.tx.i1) * C);
                dst[1] = make_int4(__float_as_int(tp.tx.w0 * tp.ty.w0), __float_as_int(tp.tx.w1 * tp.ty.w0), __float_as_int(tp.tx.w0 * tp.ty.w1),
                                   __float_as_int(tp.tx.w1 * tp.ty.w1));
                dst[2] = make_int4(tp.tz.i0 * C, tp.tz.i1 * C, __float_as_int(tp.tz.w0), __float_as_int(tp.tz.w1));
            }
        }
        sig[lane] = 0.f;
        const unsigned long long inmask = __ballot(in);
        __builtin_amdgcn_wave_barrier();
        const int q = lane & 3;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int sl = pass * 16 + (lane >> 2);
            const bool on = (inmask >> sl) & 1ull;
            if ((inmask >> (pass * 16)) & 0xffffull) {                     // wave-uniform: some sample of this pass is in the box
                float acc = 0.f;
                if (on) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int4* rp = reinterpret_cast<const int4*>(&recs[i][sl]);
                        const int4 o = rp[0], wi = rp[1], zi = rp[2];
                        for (int c4 = q * 4; c4 < C; c4 += 16) {
                            const float* pp = t.plane[i] + c4;
                            const float* lp = t.line[i] + c4;
                            float4 pa = make_float4(0.f, 0.f, 0.f, 0.f);
                            pa = f4_fma(__int_as_float(wi.x), ld4(pp + (unsigned)o.x), pa);
                            pa = f4_fma(__int_as_float(wi.y), ld4(pp + (unsigned)o.y), pa);
                            pa = f4_fma(__int_as_float(wi.z), ld4(pp + (unsigned)o.z), pa);
                            pa = f4_fma(__int_as_float(wi.w), ld4(pp + (unsigned)o.w), pa);
                            float4 la = make_float4(0.f, 0.f, 0.f, 0.f);
                            la = f4_fma(__int_as_float(zi.z), ld4(lp + (unsigned)zi.x), la);
                            la = f4_fma(__int_as_float(zi.w), ld4(lp + (unsigned)zi.y), la);
                            acc += f4_hsum(f4_mul(pa, la));
                        }
                    }
                }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (q == 0 && on) {
                    const float x = acc + m.shift;
                    sig[sl] = (x > 20.f) ? x : log1pf(expf(x));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (k < m.S) out[k] = sig[lane];
    }
}

extern "C" int clift_density_fwd(const clift_march_t* h_m, const clift_vm_t* h_dens, const float* rays,
                                 const float* jitter, int N, float* sigma, clift_stream_t s) {
    CLIFT_REQUIRE(h_dens->comps % 4 == 0, "clift_density_fwd: comps must be a multiple of 4 (got %d)", h_dens->comps);
    if (N <= 0) return 0;
    const char* form = getenv("CLIFT_DENS_FWD");                         // test hook: "thread" = the per-thread form
    const bool per_thread = form && !strcmp(form, "thread");
    if (per_thread) {
        const long total = (long)N * h_m->n_samples;
        k_density_fwd<<<cdiv(total * 4, 256), 256, 0, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), rays, jitter, total, sigma);
    } else {
        k_density_fwd_ray<<<cdiv((long)N * cdiv(h_m->n_samples, 64), DFR_WAVES), 64 * DFR_WAVES, 0, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), rays, jitter, N, sigma);
    }
    return clift_check_launch("clift_density_fwd");
}

// ============================================================================ density at arbitrary points
// tensoRF.py:114-125 on explicit normalised coordinates (dense alpha grid of the bbox shrink, renderer.py:717-754;
// reference field API compute_density).  Same 4-lanes-per-point mapping as k_density_fwd; no in-box mask (taps outside
// the table are zero-padded exactly like grid_sample).
__global__ __launch_bounds__(256) void k_density_points(VmP t, const float* __restrict__ xn3, int ldx, long total, float shift, int activation,
                                                         float* __restrict__ out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = gid >> 2;
    const int q = (int)(gid & 3);
    if (s >= total) return;
    const float xn[3] = {xn3[s * ldx + 0], xn3[s * ldx + 1], xn3[s * ldx + 2]};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const VmTaps tp = vm_taps(t, i, xn);
        for (int c4 = q * 4; c4 < t.comps; c4 += 16) acc += f4_hsum(f4_mul(vm_plane4(t, i, tp, c4), vm_line4(t, i, tp, c4)));
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) {
        const float x = acc + shift;
        out[s] = activation ? ((x > 20.f) ? x : log1pf(expf(x))) : x;
    }
}

extern "C" int clift_density_points(const clift_vm_t* h_dens, const float* xn, int ldx, long n, float shift, int activation, float* out,
                                    clift_stream_t s) {
    CLIFT_REQUIRE(h_dens->comps % 4 == 0, "clift_density_points: comps must be a multiple of 4");
    CLIFT_REQUIRE(ldx >= 3, "clift_density_points: ldx must be >= 3");
    if (n <= 0) return 0;
    k_density_points<<<cdiv(n * 4, 256), 256, 0, as_stream(s)>>>(to_dev(h_dens), xn, ldx, n, shift, activation, out);
    return clift_check_launch("clift_density_points");
}

// ============================================================================ alpha-mask bounding box (renderer.py:669-761)
// The epoch-boundary shrink asks: which voxels of the (g0, g1, g2) lattice of the current box have alpha >= threshold after a 3^3
// max-pool, and what is their index bounding box?  Pass 1 evaluates alpha on the lattice (point p_a = lo_a (1 - s) + hi_a s with
// s = the caller's linspace value of that axis, normalised like renderer.py:633, density + softplus like tensoRF.py:114-125,
// alpha = 1 - exp(-sigma step), R:752-754) -- 4 lanes per voxel as in k_density_points, fully coalesced.  Pass 2 pools, thresholds
// and reduces to six integers (min / max lattice index per axis) with wave reductions and one atomic pair per wave and axis.
__global__ __launch_bounds__(256) void k_alpha_lattice(VmP t, float3 lo, float3 hi, float3 inv_ext2, const float* __restrict__ s0,
                                                        const float* __restrict__ s1, const float* __restrict__ s2, int g0, int g1, int g2,
                                                        float shift, float step, float* __restrict__ alpha) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long v = gid >> 2, total = (long)g0 * g1 * g2;
    const int q = (int)(gid & 3);
    if (v >= total) return;
    const int i2 = (int)(v % g2), i1 = (int)((v / g2) % g1), i0 = (int)(v / ((long)g1 * g2));
    const float a = s0[i0], b = s1[i1], c = s2[i2];
    const float p[3] = {lo.x * (1.f - a) + hi.x * a, lo.y * (1.f - b) + hi.y * b, lo.z * (1.f - c) + hi.z * c};
    const float xn[3] = {(p[0] - lo.x) * inv_ext2.x - 1.f, (p[1] - lo.y) * inv_ext2.y - 1.f, (p[2] - lo.z) * inv_ext2.z - 1.f};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const VmTaps tp = vm_taps(t, i, xn);
        for (int c4 = q * 4; c4 < t.comps; c4 += 16) acc += f4_hsum(f4_mul(vm_plane4(t, i, tp, c4), vm_line4(t, i, tp, c4)));
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) {
        const float x = acc + shift;
        const float sigma = (x > 20.f) ? x : log1pf(expf(x));
        alpha[v] = 1.f - expf(-sigma * step);
    }
}

__global__ void k_box_init(int* box6) {
    if (threadIdx.x < 3) box6[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) box6[threadIdx.x] = -1;
    else if (threadIdx.x == 6) box6[6] = 0;
}

__global__ __launch_bounds__(256) void k_alpha_pool_box(const float* __restrict__ alpha, int g0, int g1, int g2, float thr, int* __restrict__ box6) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x, total = (long)g0 * g1 * g2;
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1}, cnt = 0;
    if (v < total) {
        const int i2 = (int)(v % g2), i1 = (int)((v / g2) % g1), i0 = (int)(v / ((long)g1 * g2));
        float best = 0.f;                                       // values are clamped to [0, 1] first (R:672), so 0 is the pool's floor
        for (int d0 = max(i0 - 1, 0); d0 <= min(i0 + 1, g0 - 1); ++d0)
            for (int d1 = max(i1 - 1, 0); d1 <= min(i1 + 1, g1 - 1); ++d1)
                for (int d2 = max(i2 - 1, 0); d2 <= min(i2 + 1, g2 - 1); ++d2)
                    best = fmaxf(best, fminf(fmaxf(alpha[((long)d0 * g1 + d1) * g2 + d2], 0.f), 1.f));
        if (best >= thr) { mn[0] = mx[0] = i0; mn[1] = mx[1] = i1; mn[2] = mx[2] = i2; cnt = 1; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d > 0; d >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], d)); mx[a] = max(mx[a], __shfl_xor(mx[a], d)); }
    cnt = wave_sum_i(cnt);
    if (lane_id() == 0 && cnt > 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(box6 + a, mn[a]); atomicMax(box6 + 3 + a, mx[a]); }
        atomicAdd(box6 + 6, cnt);
    }
}

extern "C" int clift_alpha_bbox(const clift_vm_t* h_dens, const float* h_lo3, const float* h_hi3, const float* h_inv_ext2, const float* s0,
                                const float* s1, const float* s2, int g0, int g1, int g2, float shift, float step, float threshold,
                                float* alpha_work, int* box7, clift_stream_t s) {
    CLIFT_REQUIRE(h_dens->comps % 4 == 0, "clift_alpha_bbox: comps must be a multiple of 4");
    CLIFT_REQUIRE(g0 >= 1 && g1 >= 1 && g2 >= 1, "clift_alpha_bbox: lattice dimensions must be positive");
    const long total = (long)g0 * g1 * g2;
    hipStream_t st = as_stream(s);
    k_box_init<<<1, 64, 0, st>>>(box7);
    k_alpha_lattice<<<cdiv(total * 4, 256), 256, 0, st>>>(to_dev(h_dens), make_float3(h_lo3[0], h_lo3[1], h_lo3[2]), make_float3(h_hi3[0], h_hi3[1], h_hi3[2]),
                                                           make_float3(h_inv_ext2[0], h_inv_ext2[1], h_inv_ext2[2]), s0, s1, s2, g0, g1, g2, shift, step,
                                                           alpha_work);
    k_alpha_pool_box<<<cdiv(total, 256), 256, 0, st>>>(alpha_work, g0, g1, g2, threshold, box7);
    return clift_check_launch("clift_alpha_bbox");
}

// ============================================================================ density backward
// Persistent blocks; `comps` lanes form a GROUP with ONE CHANNEL PER LANE (comps = 4..64, power of two; 16 in the
// reference configs => 4 groups per wave), so every atomic instruction covers whole 64-byte texels.
// The kernel is bound by the plane atomics (measured: 1.11 ms with, 0.32 ms without them at 1.8 M samples), and
// consecutive samples of a ray move ~0.3 texel per step in each plane, so a group WALKS a segment of consecutive samples
// of one ray and keeps, per plane, the current sample's four texels in registers (key, table value, gradient sum).
// A texel is written to memory (one atomic) only when the walk leaves it, and read only when the walk enters it:
// ~1.1 instead of 4 atomics and loads per plane per sample.  Line gradients accumulate in LDS and are flushed once per
// block; plane gradients go to the per-XCD accumulation copies (clift_dev.h).
// Samples per walk segment.  Measured at 1.8 M samples (profiles/r01_scatter_notes.txt): 32 -> 535 us, 16 -> 430, 8 -> 421,
// 4 -> 389 (one sample per group, no merging: 1114): longer walks save more atomics but serialise more steps per group
// and diverge more between the four groups of a wave.
constexpr int DENS_SEG_DEFAULT = 4;

template <bool LDS_LINES>
__device__ __forceinline__ void line_add(const VmG& gr, int i, size_t xoff, float* lds_line, int off, float v) {
    if (LDS_LINES) atomicAdd(lds_line + off, v);
    else if (gr.xcd_stride > 0) __hip_atomic_fetch_add(gr.line[i] + xoff + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else unsafeAtomicAdd(gr.line[i] + off, v);
}

template <bool LDS_LINES>
__global__ __launch_bounds__(1024) void k_density_bwd(MarchP m, VmP t, VmG gr, const float* __restrict__ rays,
                                                       const float* __restrict__ jitter, int N, int lg_c,
                                                       const float* __restrict__ dsigma, int DENS_SEG) {
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    const int nl = line_lds_floats(t.res, t.comps);
    if (LDS_LINES) scatter_zero_lines(lds_lines, nl);
    const size_t xoff = gr.xcd_stride > 0 ? (size_t)xcc_id() * (size_t)gr.xcd_stride : 0;
    const bool xcd = gr.xcd_stride > 0;
    const int C = t.comps;
    const int c = threadIdx.x & (C - 1);
    const long ngroups = ((long)gridDim.x * blockDim.x) >> lg_c;
    const int nchunk = (m.S + DENS_SEG - 1) / DENS_SEG;
    const long nseg = (long)N * nchunk;
    for (long seg = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> lg_c; seg < nseg; seg += ngroups) {
        const int r = (int)(seg / nchunk);
        const int k0 = (int)(seg - (long)r * nchunk) * DENS_SEG, k1 = min(m.S, k0 + DENS_SEG);
        const RayG g = load_ray(rays, r, m);
        const float jit = jitter ? jitter[r] : 0.f;
        const float* dsr = dsigma + (size_t)r * m.S;
        int ck[3][4];            // texel index (y * W + x) of the open entries, -1 = empty
        float cv[3][4], ca[3][4];  // table value / gradient sum of this lane's channel
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ck[i][q] = -1; cv[i][q] = 0.f; ca[i][q] = 0.f; }
        }
        float ds_next = dsr[k0];
        for (int k = k0; k < k1; ++k) {
            const float ds = ds_next;
            if (k + 1 < k1) ds_next = dsr[k + 1];
            if (ds == 0.f) continue;     // uniform over the lanes of a group
            float xn[3];
            if (!sample_xn(g, m, sample_z(g, m, k, jit), xn)) continue;
            float P[3], L[3], w4[3][4], wz[3][2];
            int oz[3][2];
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const VmTaps tp = vm_taps(t, i, xn);
                int a, b, v;
                vm_axes(i, a, b, v);
                const int W = t.res[a];
                int nk[4];
                w4[i][0] = tp.tx.w0 * tp.ty.w0; w4[i][1] = tp.tx.w1 * tp.ty.w0; w4[i][2] = tp.tx.w0 * tp.ty.w1; w4[i][3] = tp.tx.w1 * tp.ty.w1;
                nk[0] = tp.ty.i0 * W + tp.tx.i0; nk[1] = tp.ty.i0 * W + tp.tx.i1; nk[2] = tp.ty.i1 * W + tp.tx.i0; nk[3] = tp.ty.i1 * W + tp.tx.i1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (w4[i][q] == 0.f) nk[q] = -1;       // clamped out-of-range taps: never loaded, never written
                float* gp = gr.plane[i] + xoff;
                const float* pp = t.plane[i];
                // leave: open entries that are not part of this sample
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int key = ck[i][q];
                    if (key >= 0 && key != nk[0] && key != nk[1] && key != nk[2] && key != nk[3]) {
                        if (xcd) __hip_atomic_fetch_add(gp + (size_t)key * C + c, ca[i][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else unsafeAtomicAdd(gp + (size_t)key * C + c, ca[i][q]);
                    }
                }
                // enter: keep matching entries, load the new ones
                float nv[4], na[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float val = 0.f, sum = 0.f;
                    bool hit = false;
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (nk[q] >= 0 && ck[i][o] == nk[q]) { val = cv[i][o]; sum = ca[i][o]; hit = true; }
                    if (!hit && nk[q] >= 0) val = pp[(size_t)nk[q] * C + c];
                    nv[q] = val; na[q] = sum;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { ck[i][q] = nk[q]; cv[i][q] = nv[q]; ca[i][q] = na[q]; }
                oz[i][0] = tp.tz.i0 * C + c; oz[i][1] = tp.tz.i1 * C + c;
                wz[i][0] = tp.tz.w0; wz[i][1] = tp.tz.w1;
                const float* lp = t.line[i];
                P[i] = fmaf(w4[i][3], nv[3], fmaf(w4[i][2], nv[2], fmaf(w4[i][1], nv[1], w4[i][0] * nv[0])));
                L[i] = fmaf(wz[i][1], lp[oz[i][1]], wz[i][0] * lp[oz[i][0]]);
                acc = fmaf(P[i], L[i], acc);
            }
            for (int d = 1; d < C; d <<= 1) acc += __shfl_xor(acc, d);
            const float x = acc + m.shift;
            const float up = ds * ((x > 20.f) ? 1.f : 1.f / (1.f + expf(-x)));
            int loff = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float gP = up * L[i], gL = up * P[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) ca[i][q] = fmaf(w4[i][q], gP, ca[i][q]);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (wz[i][q] != 0.f) line_add<LDS_LINES>(gr, i, xoff, lds_lines + loff, oz[i][q], wz[i][q] * gL);
                loff += t.res[2 - i] * C;
            }
        }
        // end of the segment: write out whatever is still open
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float* gp = gr.plane[i] + xoff;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ck[i][q] >= 0) {
                    if (xcd) __hip_atomic_fetch_add(gp + (size_t)ck[i][q] * C + c, ca[i][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else unsafeAtomicAdd(gp + (size_t)ck[i][q] * C + c, ca[i][q]);
                }
        }
    }
    if (LDS_LINES) scatter_flush_lines(t, gr, lds_lines, xoff);
}

// ---------------------------------------------------------------------------- the same walk, one WAVE per (ray, 32-sample chunk)
// The walk above spends ~700 VALU instructions per step, nearly all of it tap arithmetic and key matching that is identical in the
// `comps` lanes of a group.  Here (comps <= 16) a wave owns one chunk of DU_SEG consecutive samples of one ray, lane = (plane, channel)
// (3 comps live lanes), and the walk is split in two:
//   phase 1: lane p handles sample k0 + p: gradient, position, in-box test; the active samples are compacted (ballot) and for each of
//            the three planes the lane computes the taps and files the four texels of the footprint in PARITY SLOTS -- texel (x, y) always
//            sits in slot (x & 1) + 2 (y & 1), and a 2 x 2 footprint has exactly one texel of each parity, so a footprint that moves by a
//            texel replaces the texels of one parity IN PLACE: no matching of four old against four new keys, no moving of sums between
//            slots; a slot is written out exactly when its key changes.  With the keys of the previous ACTIVE sample (fetched from that
//            lane) the lane sets one "write out" and one "restart" bit per slot (same for the two line entries, slot = z & 1) and parks a
//            64-byte record per (step, plane) in the wave's LDS slot.  With sigma from the forward it also computes the softplus derivative
//            1 - exp(-sigma) there, once per sample;
//   phase 2: the serial walk over the active samples: a lane reads its plane's record; the table values of DU_U steps are loaded together
//            (one memory round trip per DU_U steps; nothing but the six gradient sums is carried from step to step), then per step:
//            <= 6 atomics, 6 restarts, ~12 FMAs.
// Per-sample terms: the bilinear sum runs in slot order instead of corner order and (with sigma) the derivative is 1 - exp(-softplus)
// instead of the sigmoid: fp32 round-off apart from the walk above.
#ifndef DU_ABL
#define DU_ABL 0          // timing probes: 1 = 16-sample chunks, 12 waves x 2 blocks per CU; 2 = no plane atomics; 4 = 16-sample chunks, 8 waves x 3 blocks
#endif
constexpr int DU_SEG = (DU_ABL & 5) ? 16 : 32;
constexpr int DU_U = 4;                // steps whose loads are issued together
struct alignas(16) DensRec {
    int ks[4];          // texels by parity slot, as BYTE offsets (y * W + x) * comps * 4 into the channels-last table; -1 = tap out of range
    float ws[4];        // their weights
    int kz[2];          // line entries by parity slot (z & 1), byte offsets z * comps * 4; -1 = out of range
    float wz[2];
    int ctrl;           // bits 0-3 / 4-5: write out plane slot s / line slot s before this step; bits 8-11 / 12-13: restart its sum
    float up;           // dL/dsigma (x the softplus derivative when sigma is given)
    int pad[2];
};

template <bool LDS_LINES>
__global__ __launch_bounds__(1024) void k_density_bwd_u(MarchP m, VmP t, VmG gr, const float* __restrict__ rays, const float* __restrict__ jitter,
                                                           int N, int lg_c, const float* __restrict__ dsigma, const float* __restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    const int nl = LDS_LINES ? line_lds_floats(t.res, t.comps) : 0;
    if (LDS_LINES) scatter_zero_lines(lds_lines, nl);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    DensRec* const recs = reinterpret_cast<DensRec*>(lds_lines + (nl + 3) / 4 * 4) + wave * (3 * DU_SEG);
    const size_t xoff = gr.xcd_stride > 0 ? (size_t)xcc_id() * (size_t)gr.xcd_stride : 0;
    const bool xcd = gr.xcd_stride > 0;
    const int C = t.comps;
    const bool live = lane < 3 * C;
    const int pi = live ? (lane >> lg_c) : 2, c = lane & (C - 1);            // this lane's plane and channel
    const float* pp = (pi == 0 ? t.plane[0] : pi == 1 ? t.plane[1] : t.plane[2]) + c;
    const float* lp = (pi == 0 ? t.line[0] : pi == 1 ? t.line[1] : t.line[2]) + c;
    float* gp = (pi == 0 ? gr.plane[0] : pi == 1 ? gr.plane[1] : gr.plane[2]) + xoff + c;
    float* gl = (pi == 0 ? gr.line[0] : pi == 1 ? gr.line[1] : gr.line[2]) + xoff + c;
    float* ll = lds_lines + (pi == 0 ? 0 : pi == 1 ? t.res[2] * C : (t.res[2] + t.res[1]) * C) + c;
    const int nchunk = (m.S + DU_SEG - 1) / DU_SEG;
    const long items = (long)N * nchunk;
    for (long it = (long)blockIdx.x * nwaves + wave; it < items; it += (long)gridDim.x * nwaves) {
        const int r = (int)(it / nchunk), k0 = (int)(it - (long)r * nchunk) * DU_SEG;
        // ---------------- phase 1
        const int p = lane, k = k0 + p;
        const bool valid = p < DU_SEG && k < m.S;
        const float ds = valid ? dsigma[(size_t)r * m.S + k] : 0.f;
        if (__ballot(ds != 0.f) == 0ull) continue;                          // nothing flows into this chunk (most chunks behind a surface)
        const RayG g = load_ray(rays, r, m);
        float xn[3] = {0.f, 0.f, 0.f};
        bool active = false;
        if (ds != 0.f) active = sample_xn(g, m, sample_z(g, m, k, jitter ? jitter[r] : 0.f), xn);
        const unsigned long long mask = __ballot(active);
        const int na = __popcll(mask);
        if (na == 0) continue;
        {
            const unsigned long long below = mask & ((1ull << p) - 1ull);
            const int j = __popcll(below);
            const int pl = below ? 63 - __clzll(below) : p;                 // lane of the previous active sample
            float up = ds;
            if (sigma && active) up = ds * -expm1f(-sigma[(size_t)r * m.S + k]);       // d softplus(x) / dx = sigmoid(x) = 1 - exp(-softplus(x))
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int a, b, v;
                vm_axes(i, a, b, v);
                const int W = t.res[a];
                int ks[4] = {-1, -1, -1, -1}, kz[2] = {-1, -1};
                float ws[4] = {0.f, 0.f, 0.f, 0.f}, wz[2] = {0.f, 0.f};
                if (active) {
                    const VmTaps tp = vm_taps(t, i, xn);
                    const float w4[4] = {tp.tx.w0 * tp.ty.w0, tp.tx.w1 * tp.ty.w0, tp.tx.w0 * tp.ty.w1, tp.tx.w1 * tp.ty.w1};
                    const int nk[4] = {tp.ty.i0 * W + tp.tx.i0, tp.ty.i0 * W + tp.tx.i1, tp.ty.i1 * W + tp.tx.i0, tp.ty.i1 * W + tp.tx.i1};
                    const int par = tap_parity(tp.tx) + 2 * tap_parity(tp.ty);
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int q = sl ^ par;                              // corner (qx, qy) sits in slot ((x0 + qx) & 1, (y0 + qy) & 1)
                        const float wq = q == 0 ? w4[0] : q == 1 ? w4[1] : q == 2 ? w4[2] : w4[3];
                        const int kq = q == 0 ? nk[0] : q == 1 ? nk[1] : q == 2 ? nk[2] : nk[3];
                        ws[sl] = wq;
                        ks[sl] = wq == 0.f ? -1 : kq * (4 * C);              // clamped out-of-range taps: never loaded, never written
                    }
                    const int pz = tap_parity(tp.tz);
                    const int z0 = tp.tz.w0 != 0.f ? tp.tz.i0 * (4 * C) : -1, z1 = tp.tz.w1 != 0.f ? tp.tz.i1 * (4 * C) : -1;
                    kz[0] = pz ? z1 : z0; kz[1] = pz ? z0 : z1;
                    wz[0] = pz ? tp.tz.w1 : tp.tz.w0; wz[1] = pz ? tp.tz.w0 : tp.tz.w1;
                }
                int ctrl = 0;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    int pk = __shfl(ks[sl], pl);
                    if (j == 0) pk = -1;
                    if (pk != ks[sl]) ctrl |= (pk >= 0 ? (1 << sl) : 0) | (1 << (8 + sl));
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    int pk = __shfl(kz[sl], pl);
                    if (j == 0) pk = -1;
                    if (pk != kz[sl]) ctrl |= (pk >= 0 ? (1 << (4 + sl)) : 0) | (1 << (12 + sl));
                }
                if (active) {
                    int4* dst = reinterpret_cast<int4*>(recs + (j * 3 + i));
                    dst[0] = make_int4(ks[0], ks[1], ks[2], ks[3]);
                    dst[1] = make_int4(__float_as_int(ws[0]), __float_as_int(ws[1]), __float_as_int(ws[2]), __float_as_int(ws[3]));
                    dst[2] = make_int4(kz[0], kz[1], __float_as_int(wz[0]), __float_as_int(wz[1]));
                    dst[3] = make_int4(ctrl, __float_as_int(up), 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---------------- phase 2
        // per-lane bases (plane and channel folded in) + the records' 32-bit byte offsets: one 64-bit add per access, no multiplies
        auto at = [](auto* base, int off) {            // (pointer arithmetic, not integer casts: the address space must stay visible)
            typedef typename std::conditional<std::is_const<typename std::remove_pointer<decltype(base)>::type>::value, const char, char>::type B;
            return reinterpret_cast<decltype(base)>(reinterpret_cast<B*>(base) + (unsigned)off);
        };
        auto plane_out = [&](int key, float val) {
            if (!live || (DU_ABL & 2)) return;
            if (xcd) __hip_atomic_fetch_add(at(gp, key), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else unsafeAtomicAdd(at(gp, key), val);
        };
        auto line_out = [&](int key, float val) {
            if (!live) return;
            if (LDS_LINES) atomicAdd(at(ll, key), val);
            else if (xcd) __hip_atomic_fetch_add(at(gl, key), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else unsafeAtomicAdd(at(gl, key), val);
        };
        int ck[4] = {-1, -1, -1, -1}, lk[2] = {-1, -1};
        float ca[4] = {0.f, 0.f, 0.f, 0.f}, lacc[2] = {0.f, 0.f};
        for (int j0 = 0; j0 < na; j0 += DU_U) {
            int4 kq[DU_U];
            int2 kzz[DU_U];
            float tv[DU_U][4], tl[DU_U][2];
#pragma unroll
            for (int u = 0; u < DU_U; ++u) {
                const int4* src = reinterpret_cast<const int4*>(recs + (min(j0 + u, na - 1) * 3 + pi));
                kq[u] = src[0];
                kzz[u] = *reinterpret_cast<const int2*>(src + 2);
            }
#pragma unroll
            for (int u = 0; u < DU_U; ++u) {
                tv[u][0] = *at(pp, max(kq[u].x, 0)); tv[u][1] = *at(pp, max(kq[u].y, 0));
                tv[u][2] = *at(pp, max(kq[u].z, 0)); tv[u][3] = *at(pp, max(kq[u].w, 0));
                tl[u][0] = *at(lp, max(kzz[u].x, 0)); tl[u][1] = *at(lp, max(kzz[u].y, 0));
            }
#pragma unroll
            for (int u = 0; u < DU_U; ++u) {
                if (j0 + u >= na) break;                                       // uniform
                const int4* src = reinterpret_cast<const int4*>(recs + ((j0 + u) * 3 + pi));
                const int4 r1 = src[1], r2 = src[2], r3 = src[3];
                const int ks[4] = {kq[u].x, kq[u].y, kq[u].z, kq[u].w}, kz[2] = {kzz[u].x, kzz[u].y};
                const float ws[4] = {__int_as_float(r1.x), __int_as_float(r1.y), __int_as_float(r1.z), __int_as_float(r1.w)};
                const float wz[2] = {__int_as_float(r2.z), __int_as_float(r2.w)};
                const int ctrl = r3.x;
                float up = __int_as_float(r3.y);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    if (ctrl & (1 << sl)) plane_out(ck[sl], ca[sl]);
                    if (ctrl & (1 << (8 + sl))) ca[sl] = 0.f;
                    ck[sl] = ks[sl];
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    if (ctrl & (1 << (4 + sl))) line_out(lk[sl], lacc[sl]);
                    if (ctrl & (1 << (12 + sl))) lacc[sl] = 0.f;
                    lk[sl] = kz[sl];
                }
                float P = 0.f;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) P = fmaf(ws[sl], ks[sl] >= 0 ? tv[u][sl] : 0.f, P);
                const float L = fmaf(wz[1], kz[1] >= 0 ? tl[u][1] : 0.f, wz[0] * (kz[0] >= 0 ? tl[u][0] : 0.f));
                if (!sigma) {
                    // no sigma from the forward: the sample's feature again -- per channel the planes in order 0, 1, 2, then the channel butterfly
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc = fmaf(__shfl(P, i * C + c), __shfl(L, i * C + c), acc);
                    for (int dlt = 1; dlt < C; dlt <<= 1) acc += __shfl_xor(acc, dlt);
                    const float x = acc + m.shift;
                    up *= (x > 20.f) ? 1.f : 1.f / (1.f + expf(-x));
                }
                const float gP = up * L, gL = up * P;
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) ca[sl] = fmaf(ws[sl], gP, ca[sl]);
                lacc[0] = fmaf(wz[0], gL, lacc[0]); lacc[1] = fmaf(wz[1], gL, lacc[1]);
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
    if (LDS_LINES) scatter_flush_lines(t, gr, lds_lines, xoff);
}

extern "C" int clift_density_bwd(const clift_march_t* h_m, const clift_vm_t* h_dens, const clift_vm_grad_t* h_grad,
                                 const float* rays, const float* jitter, int N, const float* dsigma, const float* sigma, clift_stream_t s) {
    const int Cc = h_dens->comps;
    CLIFT_REQUIRE(Cc >= 4 && Cc <= 64 && (Cc & (Cc - 1)) == 0, "clift_density_bwd: comps must be a power of two in [4,64] (got %d)", Cc);
    if (N <= 0) return 0;
    int lg = 0;
    while ((1 << lg) < Cc) ++lg;
    const int DENS_SEG = DENS_SEG_DEFAULT;
    const char* mode = getenv("CLIFT_DENS_SCATTER");                  // test hook: "walk" = the group-per-segment walk (the only form for comps > 16) for any comps
    if (Cc <= 16 && !(mode && strcmp(mode, "walk") == 0)) {
        const int slab = (line_lds_floats(h_dens->res, Cc) * 4 + 15) / 16 * 16;
        const int rec_bytes = 3 * DU_SEG * (int)sizeof(DensRec);     // 6 KB per wave
        int wpb = 16, bpc = 1;
        bool lds_l = true;
        if (slab + 16 * rec_bytes > 160 * 1024 - 512) { lds_l = false; wpb = 8; bpc = 3; }
        else if (DU_ABL & 1) { wpb = 12; bpc = 2; }
        else if (DU_ABL & 4) { wpb = 8; bpc = 3; }
        const long items = (long)N * cdiv(h_m->n_samples, DU_SEG);
        const long want = cdiv(items, wpb);
        const int blocks = (int)(want < clift_persistent_cus() * bpc ? want : clift_persistent_cus() * bpc);
        const int dyn = (lds_l ? slab : 0) + wpb * rec_bytes;
        if (lds_l) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_density_bwd_u<true>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            k_density_bwd_u<true><<<blocks, wpb * 64, dyn, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), to_dev(h_grad), rays, jitter, N, lg, dsigma, sigma);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_density_bwd_u<false>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            k_density_bwd_u<false><<<blocks, wpb * 64, dyn, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), to_dev(h_grad), rays, jitter, N, lg, dsigma, sigma);
        }
        return clift_check_launch("clift_density_bwd");
    }
    const long nseg = (long)N * cdiv(h_m->n_samples, DENS_SEG);
    const int lds_bytes = line_lds_floats(h_dens->res, Cc) * 4;
    int threads, per_cu;
    const bool use_lds = scatter_geometry(lds_bytes, &threads, &per_cu);
    const long want = cdiv(nseg << lg, (long)threads);
    const int blocks = (int)(want < clift_persistent_cus() * per_cu ? want : clift_persistent_cus() * per_cu);
    if (use_lds) {
        if (lds_bytes > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_density_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        k_density_bwd<true><<<blocks, threads, lds_bytes, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), to_dev(h_grad), rays, jitter, N, lg, dsigma, DENS_SEG);
    } else {
        k_density_bwd<false><<<blocks, threads, 0, as_stream(s)>>>(to_dev(h_m), to_dev(h_dens), to_dev(h_grad), rays, jitter, N, lg, dsigma, DENS_SEG);
    }
    return clift_check_launch("clift_density_bwd");
}

// ============================================================================ transmittance scan (forward)
// One wavefront per ray; the S samples are swept in chunks of 64 lanes with a wave prefix product for
// T = cumprod(1 - alpha + 1e-10) and prefix sums for the distortion loss.
__global__ __launch_bounds__(256) void k_march_fwd(MarchP m, const float* __restrict__ rays, const float* __restrict__ jitter, int N,
                                                    const float* __restrict__ sigma, float* __restrict__ alpha_o, float* __restrict__ T_o,
                                                    float* __restrict__ w_o, float* __restrict__ ray_out, int* __restrict__ n_active) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int l = lane_id();
    const RayG g = load_ray(rays, r, m);
    const float jit = jitter ? jitter[r] : 0.f;
    const int S = m.S;
    const float z_last_mid = sample_z(g, m, max(S - 2, 0), jit);
    float carryT = 1.f, carryW = 0.f, carryWM = 0.f;
    float opacity = 0.f, depth = 0.f, dl = 0.f;
    int cnt = 0;
    for (int base = 0; base < S; base += 64) {
        const int k = base + l;
        const bool valid = k < S;
        const float z0 = sample_z(g, m, k, jit);
        const float z1 = sample_z(g, m, k + 1, jit);
        const float delta = (valid && k < S - 1) ? __fsub_rn(z1, z0) : 0.f;
        const float sg = valid ? sigma[(size_t)r * S + k] : 0.f;
        const float a = 1.f - expf(-(sg * (delta * m.dist_scale)));
        const float om = valid ? ((1.f - a) + 1e-10f) : 1.f;
        const float incl = wave_incl_prod(om);
        float excl = __shfl_up(incl, 1);
        if (l == 0) excl = 1.f;
        const float T = carryT * excl;
        const float w = valid ? a * T : 0.f;
        carryT *= __shfl(incl, 63);
        // distortion loss (public eff_distloss formula; midpoints renderer.py:84)
        const float mid = (k < S - 1) ? (z1 + z0) * 0.5f : z_last_mid;
        const float wm = w * mid;
        const float iw = wave_incl_sum(w), iwm = wave_incl_sum(wm);
        const float wpre = carryW + (iw - w), wmpre = carryWM + (iwm - wm);
        dl += (1.f / 3.f) * delta * w * w + 2.f * w * (mid * wpre - wmpre);
        carryW += __shfl(iw, 63);
        carryWM += __shfl(iwm, 63);
        opacity += w;
        depth += w * z0;
        cnt += (valid && w > m.thres) ? 1 : 0;
        if (valid) {
            const size_t o = (size_t)r * S + k;
            alpha_o[o] = a; T_o[o] = T; w_o[o] = w;
        }
    }
    opacity = wave_sum(opacity);
    depth = wave_sum(depth);
    dl = wave_sum(dl);
    cnt = wave_sum_i(cnt);
    if (l == 0) {
        float* ro = ray_out + (size_t)r * 8;
        ro[0] = opacity; ro[1] = depth; ro[2] = carryT; ro[3] = carryW; ro[4] = carryWM; ro[5] = dl; ro[6] = g.tmin; ro[7] = 0.f;
        n_active[r] = cnt;
    }
}

extern "C" int clift_march_fwd(const clift_march_t* h_m, const float* rays, const float* jitter, int N, const float* sigma,
                               float* alpha, float* T, float* w, float* ray_out, int* n_active, clift_stream_t s) {
    if (N <= 0) return 0;
    k_march_fwd<<<cdiv(N, 4), 256, 0, as_stream(s)>>>(to_dev(h_m), rays, jitter, N, sigma, alpha, T, w, ray_out, n_active);
    return clift_check_launch("clift_march_fwd");
}

// ============================================================================ transmittance scan (backward)
// Reverse sweep with suffix sums.  g_k = g_w[k] + g_opacity + g_dist * d(dist)/dw_k;
// dL/dalpha_k = g_k T_k - (sum_{j>k} g_j w_j) / (1 - alpha_k + 1e-10);  dL/dsigma_k = dL/dalpha_k (1-alpha_k) delta_k scale.
__global__ __launch_bounds__(256) void k_march_bwd(MarchP m, const float* __restrict__ rays, const float* __restrict__ jitter, int N,
                                                    const float* __restrict__ alpha_i, const float* __restrict__ T_i, const float* __restrict__ w_i,
                                                    const float* __restrict__ ray_out, const float* __restrict__ g_w,
                                                    const float* __restrict__ g_opacity, const float* __restrict__ g_dist,
                                                    float* __restrict__ dsigma) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int l = lane_id();
    const RayG g = load_ray(rays, r, m);
    const float jit = jitter ? jitter[r] : 0.f;
    const int S = m.S;
    const float z_last_mid = sample_z(g, m, max(S - 2, 0), jit);
    const float gop = g_opacity ? g_opacity[r] : 0.f;
    const float gd = g_dist ? g_dist[0] / (float)N : 0.f;
    const float Wtot = ray_out[(size_t)r * 8 + 3], WMtot = ray_out[(size_t)r * 8 + 4];
    float sufW = 0.f, sufWM = 0.f, sufGW = 0.f;  // sums over samples strictly after the current chunk
    const int nchunk = (S + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int k = c * 64 + l;
        const bool valid = k < S;
        const size_t o = (size_t)r * S + (valid ? k : 0);
        const float a = valid ? alpha_i[o] : 0.f, T = valid ? T_i[o] : 0.f, w = valid ? w_i[o] : 0.f;
        const float z0 = sample_z(g, m, k, jit), z1 = sample_z(g, m, k + 1, jit);
        const float delta = (valid && k < S - 1) ? __fsub_rn(z1, z0) : 0.f;
        const float mid = (k < S - 1) ? (z1 + z0) * 0.5f : z_last_mid;
        const float wm = w * mid;
        const float sw = wave_incl_suffix_sum(w), swm = wave_incl_suffix_sum(wm);
        const float Wsuf = sufW + (sw - w), WMsuf = sufWM + (swm - wm);
        const float Wpre = Wtot - Wsuf - w, WMpre = WMtot - WMsuf - wm;
        float gk = (valid ? (g_w ? g_w[o] : 0.f) : 0.f) + gop;
        gk += gd * ((2.f / 3.f) * delta * w + 2.f * (mid * (Wpre - Wsuf) + (WMsuf - WMpre)));
        if (!valid) gk = 0.f;
        const float gw = gk * w;
        const float sgw = wave_incl_suffix_sum(gw);
        const float GWsuf = sufGW + (sgw - gw);
        const float om = (1.f - a) + 1e-10f;
        const float dalpha = gk * T - GWsuf / om;
        if (valid) dsigma[o] = dalpha * (1.f - a) * (delta * m.dist_scale);
        sufW += __shfl(sw, 0);
        sufWM += __shfl(swm, 0);
        sufGW += __shfl(sgw, 0);
    }
}

extern "C" int clift_march_bwd(const clift_march_t* h_m, const float* rays, const float* jitter, int N, const float* alpha,
                               const float* T, const float* w, const float* ray_out, const float* g_w, const float* g_opacity,
                               const float* g_dist, float* dsigma, clift_stream_t s) {
    if (N <= 0) return 0;
    k_march_bwd<<<cdiv(N, 4), 256, 0, as_stream(s)>>>(to_dev(h_m), rays, jitter, N, alpha, T, w, ray_out, g_w, g_opacity, g_dist, dsigma);
    return clift_check_launch("clift_march_bwd");
}

// ============================================================================ compaction
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ cnt, int N, int* __restrict__ start) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int b = t * per, e = min(b + per, N);
    int s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) { start[i] = run; run += cnt[i]; }
    if (t == 1023) start[N] = part[1023];
}

// Capped form for sync-free steps: the compacted buffers hold `cap` rows.  Offsets are clamped to cap (rays past the cap simply lose
// their samples -- memory-safe, and reported), limit_out[0] = min(total, cap) is the row count every per-sample kernel clamps to
// (clift_bind_rows_limit), overflow[0] = max(overflow[0], total) when total > cap.
__global__ __launch_bounds__(1024) void k_scan_counts_capped(const int* __restrict__ cnt, int N, int* __restrict__ start, int cap,
                                                              int* __restrict__ limit_out, int* __restrict__ overflow) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int b = t * per, e = min(b + per, N);
    int s = 0;
    for (int i = b; i < e; ++i) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) { start[i] = min(run, cap); run += cnt[i]; }
    if (t == 1023) {
        const int total = part[1023];
        start[N] = min(total, cap);
        limit_out[0] = min(total, cap);
        if (total > cap) atomicMax(overflow, total);
    }
}

extern "C" int clift_scan_counts_capped(const int* n_active, int N, int* ray_start, int cap, int* limit_out, int* overflow, clift_stream_t s) {
    CLIFT_REQUIRE(N >= 0 && cap >= 0, "clift_scan_counts_capped: negative N or cap");
    k_scan_counts_capped<<<1, 1024, 0, as_stream(s)>>>(n_active, N, ray_start, cap, limit_out, overflow);
    return clift_check_launch("clift_scan_counts_capped");
}

extern "C" int clift_scan_counts(const int* n_active, int N, int* ray_start, clift_stream_t s) {
    CLIFT_REQUIRE(N >= 0, "clift_scan_counts: negative N");
    k_scan_counts<<<1, 1024, 0, as_stream(s)>>>(n_active, N, ray_start);
    return clift_check_launch("clift_scan_counts");
}

__global__ __launch_bounds__(256) void k_compact_fill(const float* __restrict__ w, const int* __restrict__ start, int N, int S,
                                                       float thres, int* __restrict__ act, int cap) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int l = lane_id();
    int run = start[r];
    for (int base = 0; base < S; base += 64) {
        const int k = base + l;
        const bool on = (k < S) && (w[(size_t)r * S + k] > thres);
        const unsigned long long bal = __ballot(on);
        const int pre = __popcll(bal & ((1ull << l) - 1ull));
        if (on && run + pre < cap) act[run + pre] = r * S + k;
        run += __popcll(bal);
    }
}

extern "C" int clift_compact_fill(const float* w, const int* ray_start, int N, int S, float thres, int* act_idx, clift_stream_t s) {
    if (N <= 0) return 0;
    CLIFT_REQUIRE((long)N * S < 2147483647L, "clift_compact_fill: N*S overflows int32 sample ids");
    k_compact_fill<<<cdiv(N, 4), 256, 0, as_stream(s)>>>(w, ray_start, N, S, thres, act_idx, 2147483647);
    return clift_check_launch("clift_compact_fill");
}

extern "C" int clift_compact_fill_capped(const float* w, const int* ray_start, int N, int S, float thres, int* act_idx, int cap, clift_stream_t s) {
    if (N <= 0) return 0;
    CLIFT_REQUIRE((long)N * S < 2147483647L, "clift_compact_fill_capped: N*S overflows int32 sample ids");
    k_compact_fill<<<cdiv(N, 4), 256, 0, as_stream(s)>>>(w, ray_start, N, S, thres, act_idx, cap);
    return clift_check_launch("clift_compact_fill_capped");
}

// ============================================================================ fold the 8 per-XCD accumulation copies
__global__ __launch_bounds__(256) void k_xcd_reduce(float* __restrict__ work, long stride, long n4, float* __restrict__ dst) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 acc = *reinterpret_cast<const float4*>(dst + i * 4);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float4* w = reinterpret_cast<float4*>(work + (size_t)x * stride + i * 4);
            const float4 v = *w;
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            *w = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4*>(dst + i * 4) = acc;
    }
}

extern "C" int clift_xcd_reduce(float* work, long xcd_stride, long n, float* dst, clift_stream_t s) {
    CLIFT_REQUIRE(n % 4 == 0 && xcd_stride % 4 == 0, "clift_xcd_reduce: n and xcd_stride must be multiples of 4");
    if (n <= 0) return 0;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    k_xcd_reduce<<<blocks, 256, 0, as_stream(s)>>>(work, xcd_stride, n4, dst);
    return clift_check_launch("clift_xcd_reduce");
}
