// losses.hip -- TV regulariser (a18), pixel losses (a19), contrastive (a16) and slow-fast (a17) instance losses,
// Adam / EMA plumbing.  All are small or purely HBM-streaming kernels.
#include "clift_dev.h"

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return t;  // valid on thread 0
}

// ============================================================================ TV (model/loss/loss.py:14-22)
// x is channels-last (H, W, C).  loss = 2 (sum_h (x[h+1]-x[h])^2 / cnt_h + sum_w (x[w+1]-x[w])^2 / cnt_w),
// cnt_h = C (H-1) W + 1e-4, cnt_w = C H (W-1) + 1e-4 (batch = 1).  One streaming pass produces value and gradient.
__global__ __launch_bounds__(256) void k_tv(const float* __restrict__ x, int H, int W, int C, float weight, float* __restrict__ grad,
                                             float* __restrict__ loss) {
    __shared__ float sh[4];
    const long total = (long)H * W * C;
    const float ch = 2.f / ((float)C * (H - 1) * W + 1e-4f), cw = 2.f / ((float)C * H * (W - 1) + 1e-4f);
    float part = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e / ((long)W * C)), w = (int)((e / C) % W);
        const float v = x[e];
        float g = 0.f;
        if (h + 1 < H) { const float d = x[e + (long)W * C] - v; part += ch * d * d; g -= ch * 2.f * d; }
        if (h > 0) g += ch * 2.f * (v - x[e - (long)W * C]);
        if (w + 1 < W) { const float d = x[e + C] - v; part += cw * d * d; g -= cw * 2.f * d; }
        if (w > 0) g += cw * 2.f * (v - x[e - C]);
        if (grad) grad[e] += weight * g;
    }
    const float t = block_sum_256(part, sh);
    if (threadIdx.x == 0 && loss) unsafeAtomicAdd(loss, weight * t);
}

// all planes of a step in one launch: blockIdx.y selects the plane.  Thread = four channels of one pixel (channels-last planes,
// C % 4 == 0): one 32-bit division per 16 bytes instead of two 64-bit divisions per element, float4 neighbours.
__global__ __launch_bounds__(256) void k_tv_multi(clift_tv_set_t set, float* __restrict__ loss) {
    __shared__ float sh[4];
    const int i = blockIdx.y;
    const float* __restrict__ x = set.plane[i];
    float* __restrict__ grad = set.grad[i];
    const int H = set.H[i], W = set.W[i], C = set.C[i];
    const float weight = set.weight[i];
    const float ch = 2.f / ((float)C * (H - 1) * W + 1e-4f), cw = 2.f / ((float)C * H * (W - 1) + 1e-4f);
    float part = 0.f;
    if ((C & 3) == 0) {
        const int cq = C >> 2, total = H * W * cq;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
            const int pix = e / cq, c4 = (e - pix * cq) * 4;
            const int h = pix / W, w = pix - h * W;
            const size_t o = (size_t)pix * C + c4;
            const float4 v = ld4(x + o);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h + 1 < H) {
                const float4 n = ld4(x + o + (size_t)W * C);
                const float4 d = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
                part += ch * ((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
                g.x -= ch * 2.f * d.x; g.y -= ch * 2.f * d.y; g.z -= ch * 2.f * d.z; g.w -= ch * 2.f * d.w;
            }
            if (h > 0) {
                const float4 n = ld4(x + o - (size_t)W * C);
                g.x += ch * 2.f * (v.x - n.x); g.y += ch * 2.f * (v.y - n.y); g.z += ch * 2.f * (v.z - n.z); g.w += ch * 2.f * (v.w - n.w);
            }
            if (w + 1 < W) {
                const float4 n = ld4(x + o + C);
                const float4 d = make_float4(n.x - v.x, n.y - v.y, n.z - v.z, n.w - v.w);
                part += cw * ((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
                g.x -= cw * 2.f * d.x; g.y -= cw * 2.f * d.y; g.z -= cw * 2.f * d.z; g.w -= cw * 2.f * d.w;
            }
            if (w > 0) {
                const float4 n = ld4(x + o - C);
                g.x += cw * 2.f * (v.x - n.x); g.y += cw * 2.f * (v.y - n.y); g.z += cw * 2.f * (v.z - n.z); g.w += cw * 2.f * (v.w - n.w);
            }
            if (grad) {
                float4 go = ld4(grad + o);
                go.x += weight * g.x; go.y += weight * g.y; go.z += weight * g.z; go.w += weight * g.w;
                *reinterpret_cast<float4*>(grad + o) = go;
            }
        }
    } else {
        const long total = (long)H * W * C;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const int h = (int)(e / ((long)W * C)), w = (int)((e / C) % W);
            const float v = x[e];
            float g = 0.f;
            if (h + 1 < H) { const float d = x[e + (long)W * C] - v; part += ch * d * d; g -= ch * 2.f * d; }
            if (h > 0) g += ch * 2.f * (v - x[e - (long)W * C]);
            if (w + 1 < W) { const float d = x[e + C] - v; part += cw * d * d; g -= cw * 2.f * d; }
            if (w > 0) g += cw * 2.f * (v - x[e - C]);
            if (grad) grad[e] += weight * g;
        }
    }
    const float t = block_sum_256(part, sh);
    if (threadIdx.x == 0 && loss && t != 0.f) unsafeAtomicAdd(loss, weight * t);
}

extern "C" int clift_tv_fwd_bwd_multi(const clift_tv_set_t* set, float* loss_accum, clift_stream_t s) {
    CLIFT_REQUIRE(set->n >= 0 && set->n <= CLIFT_TV_MAX, "clift_tv_fwd_bwd_multi: n must be in [0,%d]", CLIFT_TV_MAX);
    if (set->n == 0) return 0;
    long most = 0;
    for (int i = 0; i < set->n; ++i) {
        CLIFT_REQUIRE(set->H[i] > 0 && set->W[i] > 0 && set->C[i] > 0 && set->plane[i], "clift_tv_fwd_bwd_multi: bad plane %d", i);
        CLIFT_REQUIRE((long)set->H[i] * set->W[i] * set->C[i] < 2147483647L, "clift_tv_fwd_bwd_multi: plane %d too large", i);
        CLIFT_REQUIRE((((uintptr_t)set->plane[i]) & 15) == 0 && (((uintptr_t)set->grad[i]) & 15) == 0, "clift_tv_fwd_bwd_multi: plane %d not 16-byte aligned", i);
        const long t = (long)set->H[i] * set->W[i] * set->C[i] / 4;
        most = t > most ? t : most;
    }
    // few, long-running blocks: every block ends in ONE atomic on the same loss word, and same-address atomics serialise (~5 ns each:
    // 12 k blocks spent 60 us there)
    const int bx = (int)((most + 255) / 256 < 128 ? (most + 255) / 256 : 128);
    k_tv_multi<<<dim3(bx, set->n), 256, 0, as_stream(s)>>>(*set, loss_accum);
    return clift_check_launch("clift_tv_fwd_bwd_multi");
}

extern "C" int clift_tv_fwd_bwd(const float* plane, int H, int W, int C, float weight, float* grad, float* loss_accum, clift_stream_t s) {
    CLIFT_REQUIRE(H > 0 && W > 0 && C > 0, "clift_tv_fwd_bwd: bad shape");
    const long total = (long)H * W * C;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    k_tv<<<blocks, 256, 0, as_stream(s)>>>(plane, H, W, C, weight, grad, loss_accum);
    return clift_check_launch("clift_tv_fwd_bwd");
}

// ============================================================================ pixel losses (trainer T:160,177-178)
// Per-row semantic loss and its gradient.  sce == 0: CrossEntropyLoss(weight=cw, reduction='none') with soft targets (T:75,177):
//   l = -sum_c cw_c p_c log_softmax(x)_c.  sce != 0: SCELoss (model/loss/loss.py:36-59): l = alpha * CE + beta * RCE with
//   RCE = -sum_c clamp(softmax(x . cw), 1e-8, 1)_c * log(clamp(p_c, 1e-8, 1)) * cw_c   (the prediction is re-softmaxed over the
//   class-weighted logits, L:50-52).  gx (nullable) receives k * d l / d x.
__device__ __forceinline__ float semantic_row_loss(const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ cw, int C,
                                                   int sce, float alpha, float beta, float k, float* __restrict__ gx) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float lse = mx + logf(se);
    float ce = 0.f, pw = 0.f;
    for (int c = 0; c < C; ++c) {
        const float wc = cw ? cw[c] : 1.f;
        ce -= wc * p[c] * (x[c] - lse);
        pw += wc * p[c];
    }
    if (!sce) {
        if (gx)
            for (int c = 0; c < C; ++c) gx[c] = k * (expf(x[c] - lse) * pw - (cw ? cw[c] : 1.f) * p[c]);
        return ce;
    }
    // reverse cross entropy over q = softmax(cw . x)
    float my = -INFINITY;
    for (int c = 0; c < C; ++c) my = fmaxf(my, (cw ? cw[c] : 1.f) * x[c]);
    float sy = 0.f;
    for (int c = 0; c < C; ++c) sy += expf((cw ? cw[c] : 1.f) * x[c] - my);
    float rce = 0.f, aq = 0.f;          // aq = sum_c a_c q_c with a_c = d rce / d q_c (zero where the clamp is active)
    for (int c = 0; c < C; ++c) {
        const float wc = cw ? cw[c] : 1.f;
        const float q = expf(wc * x[c] - my) / sy;
        const float L = logf(fminf(fmaxf(p[c], 1e-8f), 1.0f));
        rce -= fminf(fmaxf(q, 1e-8f), 1.0f) * L * wc;
        if (q >= 1e-8f && q <= 1.0f) aq += (-L * wc) * q;
    }
    if (gx)
        for (int c = 0; c < C; ++c) {
            const float wc = cw ? cw[c] : 1.f;
            const float q = expf(wc * x[c] - my) / sy;
            const float L = logf(fminf(fmaxf(p[c], 1e-8f), 1.0f));
            const float a = (q >= 1e-8f && q <= 1.0f) ? -L * wc : 0.f;
            gx[c] = k * (alpha * (expf(x[c] - lse) * pw - wc * p[c]) + beta * wc * q * (a - aq));
        }
    return alpha * ce + beta * rce;
}

__global__ __launch_bounds__(256) void k_pixel_losses(const float* __restrict__ rgb, const float* __restrict__ gt, const float* __restrict__ sem,
                                                       const float* __restrict__ probs, const float* __restrict__ conf,
                                                       const float* __restrict__ cw, const float* __restrict__ maskf, int N, int C,
                                                       float w_rgb, float w_sem, int sce, float alpha, float beta,
                                                       float* __restrict__ out2, float* __restrict__ g_rgb, float* __restrict__ g_sem) {
    __shared__ float sh[4];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float l_rgb = 0.f, l_sem = 0.f;
    if (r < N) {
        const float mk = maskf ? maskf[r] : 1.f;
        if (rgb && gt) {
            for (int c = 0; c < 3; ++c) {
                const float d = mk * (rgb[(size_t)r * 3 + c] - gt[(size_t)r * 3 + c]);
                l_rgb += d * d;
                if (g_rgb) g_rgb[(size_t)r * 3 + c] = w_rgb * 2.f * d * mk / (3.f * N);
            }
        }
        if (sem && probs) {
            const float cf = (conf ? conf[r] : 1.f) * mk;
            l_sem = cf * semantic_row_loss(sem + (size_t)r * C, probs + (size_t)r * C, cw, C, sce, alpha, beta, w_sem * cf / (float)N,
                                           g_sem ? g_sem + (size_t)r * C : nullptr);
        }
    }
    const float a = block_sum_256(l_rgb, sh);
    const float b = block_sum_256(l_sem, sh);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(out2 + 0, a / (3.f * N));
        unsafeAtomicAdd(out2 + 1, b / (float)N);
    }
}

extern "C" int clift_pixel_losses(const float* rgb, const float* rgb_gt, const float* sem, const float* probs, const float* conf,
                                  const float* class_w, const float* maskf, int N, int C, float w_rgb, float w_sem, float* out2,
                                  float* g_rgb, float* g_sem, clift_stream_t s) {
    if (N <= 0) return 0;
    k_pixel_losses<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(rgb, rgb_gt, sem, probs, conf, class_w, maskf, N, C, w_rgb, w_sem, 0, 1.f, 0.f, out2, g_rgb, g_sem);
    return clift_check_launch("clift_pixel_losses");
}

extern "C" int clift_pixel_losses_sce(const float* rgb, const float* rgb_gt, const float* sem, const float* probs, const float* conf,
                                      const float* class_w, const float* maskf, int N, int C, float w_rgb, float w_sem, float alpha,
                                      float beta, float* out2, float* g_rgb, float* g_sem, clift_stream_t s) {
    if (N <= 0) return 0;
    k_pixel_losses<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(rgb, rgb_gt, sem, probs, conf, class_w, maskf, N, C, w_rgb, w_sem, 1, alpha, beta, out2, g_rgb, g_sem);
    return clift_check_launch("clift_pixel_losses_sce");
}

// Per-row form of the semantic losses (the loss callables of the reference return one value per pixel, reduction='none'):
// loss_rows (N), grad_rows (N, C; nullable) = d loss_rows[i] / d pred[i, :].
__global__ __launch_bounds__(256) void k_semantic_rows(const float* __restrict__ pred, const float* __restrict__ probs, const float* __restrict__ cw,
                                                        int N, int C, int sce, float alpha, float beta, float* __restrict__ loss_rows,
                                                        float* __restrict__ grad_rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    loss_rows[r] = semantic_row_loss(pred + (size_t)r * C, probs + (size_t)r * C, cw, C, sce, alpha, beta, 1.f,
                                     grad_rows ? grad_rows + (size_t)r * C : nullptr);
}

extern "C" int clift_semantic_loss_rows(const float* pred, const float* probs, const float* class_w, int N, int C, int sce, float alpha,
                                        float beta, float* loss_rows, float* grad_rows, clift_stream_t s) {
    CLIFT_REQUIRE(C >= 1, "clift_semantic_loss_rows: C must be positive");
    if (N <= 0) return 0;
    k_semantic_rows<<<cdiv(N, 256), 256, 0, as_stream(s)>>>(pred, probs, class_w, N, C, sce, alpha, beta, loss_rows, grad_rows);
    return clift_check_launch("clift_semantic_loss_rows");
}

// ============================================================================ segment consistency (T:185-197)
// The per-segment mean of the rendered semantic features (torch_scatter.scatter_mean) picks one class per 2D segment; every
// ray of the segment is pulled towards it with the class-weighted, confidence-weighted cross entropy.
__global__ __launch_bounds__(256) void k_segment_sums(const float* __restrict__ f, int ld, const int* __restrict__ group, int B, int C,
                                                       float* __restrict__ sums, float* __restrict__ counts) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * C) return;
    const int i = (int)(gid / C), c = (int)(gid - (long)i * C);
    const int g = group[i];
    unsafeAtomicAdd(sums + (size_t)g * C + c, f[(size_t)i * ld + c]);
    if (c == 0) unsafeAtomicAdd(counts + g, 1.f);
}

__global__ __launch_bounds__(256) void k_segment_ce(const float* __restrict__ f, int ld, const int* __restrict__ group, const float* __restrict__ conf,
                                                     const float* __restrict__ cw, int B, int C, const float* __restrict__ sums, float scale,
                                                     float* __restrict__ loss, float* __restrict__ grad, int ldg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float li = 0.f;
    if (i < B) {
        const float* x = f + (size_t)i * ld;
        const float* m = sums + (size_t)group[i] * C;      // argmax of the sums = argmax of the means (first maximum, like torch)
        int t = 0;
        float best = m[0], mx = x[0];
        for (int c = 1; c < C; ++c) {
            if (m[c] > best) { best = m[c]; t = c; }
            mx = fmaxf(mx, x[c]);
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        const float w = cw[t] * conf[i];
        li = -w * (x[t] - lse);
        if (grad) {
            const float k = scale * w / (float)B;
            float* gr = grad + (size_t)i * ldg;
            for (int c = 0; c < C; ++c) gr[c] = k * (expf(x[c] - lse) - (c == t ? 1.f : 0.f));
        }
    }
    // block reduction of the loss
    __shared__ float sh[4];
    for (int d = 32; d > 0; d >>= 1) li += __shfl_xor(li, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = li;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (sh[0] + sh[1] + sh[2] + sh[3]) / (float)B);
}

extern "C" int clift_segment_loss(const float* feats, int ld, const int* group, const float* conf, const float* class_w, int B, int C, int G,
                                  float scale, float* work, float* loss, float* grad, int ldg, clift_stream_t s) {
    CLIFT_REQUIRE(C >= 1 && G >= 1, "clift_segment_loss: C and G must be positive");
    if (B <= 0) return 0;
    hipStream_t st = as_stream(s);
    CLIFT_REQUIRE(hipMemsetAsync(work, 0, sizeof(float) * ((size_t)G * C + G), st) == hipSuccess, "clift_segment_loss: memset of the work buffer failed");
    k_segment_sums<<<cdiv((long)B * C, 256), 256, 0, st>>>(feats, ld, group, B, C, work, work + (size_t)G * C);
    k_segment_ce<<<cdiv(B, 256), 256, 0, st>>>(feats, ld, group, conf, class_w, B, C, work, scale, loss, grad, ldg);
    return clift_check_launch("clift_segment_loss");
}

// ============================================================================ contrastive_loss (model/loss/loss.py:62-82)
// l_ij = exp(exp(-d2_ij / tau_ij)), tau = temperature for positive pairs (same label, i != j) and 1 otherwise;
// p_i = sum_j l_ij [pos], Z_i = sum_j l_ij (diagonal included); loss = -sum_{p_i != 0} log(p_i / Z_i) / B.
// O(B^2 E) with B <= 1024 and E = 3: a K=3 "GEMM" -- VALU, one block per row.
#define CL_MAXE 32

__global__ __launch_bounds__(256) void k_contrastive_rows(const float* __restrict__ f, const int* __restrict__ y, int B, int E, float temp,
                                                           float* __restrict__ pz) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    float fi[CL_MAXE];
    for (int e = 0; e < E; ++e) fi[e] = f[(size_t)i * E + e];
    const int yi = y[i];
    float p = 0.f, Z = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        float d2 = 0.f;
        for (int e = 0; e < E; ++e) { const float d = fi[e] - f[(size_t)j * E + e]; d2 = fmaf(d, d, d2); }
        const bool pos = (y[j] == yi) && (j != i);
        const float l = expf(expf(-d2 / (pos ? temp : 1.f)));
        Z += l;
        if (pos) p += l;
    }
    const float ps = block_sum_256(p, sh);
    const float zs = block_sum_256(Z, sh);
    if (threadIdx.x == 0) { pz[i] = ps; pz[B + i] = zs; }
}

__global__ __launch_bounds__(256) void k_contrastive_grad(const float* __restrict__ f, const int* __restrict__ y, int B, int E, float temp,
                                                           const float* __restrict__ pz, float* __restrict__ loss, float* __restrict__ gf) {
    __shared__ float sh[4];
    const int a = blockIdx.x;
    float fa[CL_MAXE], ga[CL_MAXE];
    for (int e = 0; e < E; ++e) { fa[e] = f[(size_t)a * E + e]; ga[e] = 0.f; }
    const int ya = y[a];
    const float pa = pz[a], Za = pz[B + a];
    const float invB = 1.f / (float)B;
    for (int j = threadIdx.x; j < B; j += 256) {
        if (j == a) continue;
        float d2 = 0.f;
        for (int e = 0; e < E; ++e) { const float d = fa[e] - f[(size_t)j * E + e]; d2 = fmaf(d, d, d2); }
        const bool pos = (y[j] == ya);
        const float tau = pos ? temp : 1.f;
        const float sk = expf(-d2 / tau), l = expf(sk);
        const float dl_dd2 = l * sk * (-1.f / tau);
        // row a (valid iff p_a != 0): dL/dl_aj ; row j (valid iff p_j != 0): dL/dl_ja  (l is symmetric)
        float G = 0.f;
        if (pa != 0.f) G += -invB * ((pos ? 1.f / pa : 0.f) - 1.f / Za);
        const float pj = pz[j], Zj = pz[B + j];
        if (pj != 0.f) G += -invB * ((pos ? 1.f / pj : 0.f) - 1.f / Zj);
        G *= dl_dd2;
        for (int e = 0; e < E; ++e) ga[e] = fmaf(2.f * G, fa[e] - f[(size_t)j * E + e], ga[e]);
    }
    for (int e = 0; e < E; ++e) {
        const float t = block_sum_256(ga[e], sh);
        if (threadIdx.x == 0 && gf) gf[(size_t)a * E + e] = t;
    }
    if (threadIdx.x == 0 && pa != 0.f) unsafeAtomicAdd(loss, -logf(pa / Za) * invB);
}

__global__ void k_zero1(float* p) { p[0] = 0.f; }

extern "C" int clift_contrastive(const float* feat, const int* labels, int B, int E, float temperature, float* loss, float* g_feat,
                                 float* work, clift_stream_t s) {
    CLIFT_REQUIRE(E >= 1 && E <= CL_MAXE, "clift_contrastive: E must be in [1,%d]", CL_MAXE);
    k_zero1<<<1, 1, 0, as_stream(s)>>>(loss);
    if (B <= 0) return clift_check_launch("clift_contrastive");
    k_contrastive_rows<<<B, 256, 0, as_stream(s)>>>(feat, labels, B, E, temperature, work);
    k_contrastive_grad<<<B, 256, 0, as_stream(s)>>>(feat, labels, B, E, temperature, work, loss, g_feat);
    return clift_check_launch("clift_contrastive");
}

// ============================================================================ slow-fast loss (trainer T:261-309)
// fast set = rays [0, B/2) with fast features inst[:, :E]; slow set = rays [B/2, B) with (detached) slow
// features inst[:, E:].  Per fast row i (label l): n_f, n_s, slow centroid c_l, concentration term
// -exp(-|f_i - c_l|^2) conf_i / n_f (if n_s > 0), contrastive numerators/denominators over the slow set with
// the UN-squared Euclidean distance (torch.cdist, T:304).
// work layout (floats), H = B/2 rows: [num(H) | Z(H) | conc(H) | nf(H) | ns(H) | cent(H*E) | scalars(4)].
__global__ __launch_bounds__(256) void k_sf_rows(const float* __restrict__ inst, const int* __restrict__ y, const float* __restrict__ conf, int B,
                                                  int E, float* __restrict__ work) {
    __shared__ float sh[4];
    const int H = B / 2, i = blockIdx.x;
    float fi[CL_MAXE], cs[CL_MAXE];
    for (int e = 0; e < E; ++e) { fi[e] = inst[(size_t)i * 2 * E + e]; cs[e] = 0.f; }
    const int yi = y[i];
    float num = 0.f, Z = 0.f, ns = 0.f, nf = 0.f;
    for (int j = H + threadIdx.x; j < B; j += 256) {
        const float* sj = inst + (size_t)j * 2 * E + E;
        float d2 = 0.f;
        for (int e = 0; e < E; ++e) { const float d = fi[e] - sj[e]; d2 = fmaf(d, d, d2); }
        const float l = expf(expf(-sqrtf(d2)));
        Z += l;
        if (y[j] == yi) {
            num += l; ns += 1.f;
            for (int e = 0; e < E; ++e) cs[e] += sj[e];
        }
    }
    for (int j = threadIdx.x; j < H; j += 256) nf += (y[j] == yi) ? 1.f : 0.f;
    num = block_sum_256(num, sh); Z = block_sum_256(Z, sh); ns = block_sum_256(ns, sh); nf = block_sum_256(nf, sh);
    float cent[CL_MAXE];
    for (int e = 0; e < E; ++e) cent[e] = block_sum_256(cs[e], sh);
    if (threadIdx.x == 0) {
        float conc = 0.f;
        if (ns > 0.f) {
            float d2 = 0.f;
            for (int e = 0; e < E; ++e) { cent[e] /= ns; const float d = fi[e] - cent[e]; d2 = fmaf(d, d, d2); }
            conc = -expf(-d2) * conf[i] / nf;
        }
        work[i] = num; work[H + i] = Z; work[2 * H + i] = conc; work[3 * H + i] = nf; work[4 * H + i] = ns;
        for (int e = 0; e < E; ++e) work[5 * H + (size_t)i * E + e] = cent[e];
    }
}

__global__ __launch_bounds__(256) void k_sf_reduce(int B, int E, float* __restrict__ work, float* __restrict__ loss) {
    __shared__ float sh[4];
    const int H = B / 2;
    float lcap = 0.f, nvalid = 0.f, conc = 0.f, lg = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
        const float num = work[i], Z = work[H + i], nf = work[3 * H + i], ns = work[4 * H + i];
        if (ns > 0.f) { lcap += 1.f / nf; conc += work[2 * H + i]; }
        if (num != 0.f) { nvalid += 1.f; lg += -logf(num / Z); }
    }
    lcap = block_sum_256(lcap, sh); nvalid = block_sum_256(nvalid, sh); conc = block_sum_256(conc, sh); lg = block_sum_256(lg, sh);
    if (threadIdx.x == 0) {
        float* sc = work + 5 * H + (size_t)H * E;
        sc[0] = lcap; sc[1] = nvalid;
        // number of intersecting labels = round(lcap) (sum over rows of 1/n_f)
        const float nl = rintf(lcap);
        sc[2] = nl;
        loss[0] = (nl > 0.f ? conc / nl : 0.f) + lg / nvalid;  // 0/0 -> NaN like torch's mean of an empty tensor
    }
}

__global__ __launch_bounds__(64) void k_sf_grad(const float* __restrict__ inst, const int* __restrict__ y, const float* __restrict__ conf, int B,
                                                 int E, const float* __restrict__ work, float* __restrict__ g) {
    const int H = B / 2, i = blockIdx.x, l = threadIdx.x;
    float* gi = g + (size_t)i * 2 * E;
    if (i >= H) {  // slow ray set: no gradient at all
        for (int e = l; e < 2 * E; e += 64) gi[e] = 0.f;
        return;
    }
    const float* sc = work + 5 * H + (size_t)H * E;
    const float nvalid = sc[1], nl = sc[2];
    float fi[CL_MAXE], ga[CL_MAXE];
    for (int e = 0; e < E; ++e) { fi[e] = inst[(size_t)i * 2 * E + e]; ga[e] = 0.f; }
    const int yi = y[i];
    const float num = work[i], Z = work[H + i], nf = work[3 * H + i], ns = work[4 * H + i];
    if (num != 0.f) {
        for (int j = H + l; j < B; j += 64) {
            const float* sj = inst + (size_t)j * 2 * E + E;
            float d2 = 0.f;
            for (int e = 0; e < E; ++e) { const float d = fi[e] - sj[e]; d2 = fmaf(d, d, d2); }
            const float r = sqrtf(d2);
            if (r == 0.f) continue;  // torch.cdist backward: zero gradient at zero distance
            const float sk = expf(-r), lij = expf(sk);
            const float dL = -(1.f / nvalid) * (((y[j] == yi) ? 1.f / num : 0.f) - 1.f / Z);
            const float co = dL * lij * sk * (-1.f) / r;
            for (int e = 0; e < E; ++e) ga[e] = fmaf(co, fi[e] - sj[e], ga[e]);
        }
    }
    for (int e = 0; e < E; ++e) ga[e] = wave_sum(ga[e]);
    if (l == 0) {
        if (ns > 0.f && nl > 0.f) {
            const float* c = work + 5 * H + (size_t)i * E;
            float d2 = 0.f;
            for (int e = 0; e < E; ++e) { const float d = fi[e] - c[e]; d2 = fmaf(d, d, d2); }
            const float co = expf(-d2) * conf[i] / nf / nl;
            for (int e = 0; e < E; ++e) ga[e] = fmaf(co * 2.f, fi[e] - c[e], ga[e]);
        }
        for (int e = 0; e < E; ++e) { gi[e] = ga[e]; gi[E + e] = 0.f; }
    }
}

extern "C" int clift_slow_fast(const float* inst, const int* labels, const float* conf, int B, int E, float* loss, float* g_inst,
                               float* work, clift_stream_t s) {
    CLIFT_REQUIRE(E >= 1 && E <= CL_MAXE, "clift_slow_fast: E must be in [1,%d]", CL_MAXE);
    k_zero1<<<1, 1, 0, as_stream(s)>>>(loss);
    const int H = B / 2;
    if (H >= 1) {
        k_sf_rows<<<H, 256, 0, as_stream(s)>>>(inst, labels, conf, B, E, work);
        k_sf_reduce<<<1, 256, 0, as_stream(s)>>>(B, E, work, loss);
    }
    if (g_inst && B > 0) {
        if (H >= 1) k_sf_grad<<<B, 64, 0, as_stream(s)>>>(inst, labels, conf, B, E, work, g_inst);
        else (void)hipMemsetAsync(g_inst, 0, sizeof(float) * (size_t)B * 2 * E, as_stream(s));
    }
    return clift_check_launch("clift_slow_fast");
}

// ============================================================================ Adam / EMA
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                               float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = g[i] + wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv; v[i] = vv;
        p[i] = pv - (lr / bc1) * mv / (sqrtf(vv) / bc2s + eps);
    }
}

extern "C" int clift_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, clift_stream_t s) {
    CLIFT_REQUIRE(step >= 1, "clift_adam: step must be >= 1");
    if (n <= 0) return 0;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    k_adam<<<blocks, 256, 0, as_stream(s)>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
    return clift_check_launch("clift_adam");
}

__global__ __launch_bounds__(256) void k_ema(float* __restrict__ slow, const float* __restrict__ fast, long n, float mom) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        slow[i] = slow[i] * mom + (1.f - mom) * fast[i];
}

extern "C" int clift_ema(float* slow, const float* fast, long n, float momentum, clift_stream_t s) {
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    k_ema<<<blocks, 256, 0, as_stream(s)>>>(slow, fast, n, momentum);
    return clift_check_launch("clift_ema");
}

// ============================================================================ nearest-centroid assignment (RP:371-419)
// labels[i] = argmin_c |feat_i - centroid_c|  for points with valid[i] != 0 (else -1); K, E small (tens, 3).
__global__ __launch_bounds__(256) void k_nearest_centroid(const float* __restrict__ feat, int ldf, int E, const float* __restrict__ cent, int K,
                                                           const unsigned char* __restrict__ valid, long n, int* __restrict__ labels) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (valid && !valid[i]) { labels[i] = -1; return; }
    const float* f = feat + i * ldf;
    float best = INFINITY;
    int arg = -1;
    for (int c = 0; c < K; ++c) {
        float d2 = 0.f;
        for (int e = 0; e < E; ++e) { const float d = f[e] - cent[c * E + e]; d2 = fmaf(d, d, d2); }
        if (d2 < best) { best = d2; arg = c; }
    }
    labels[i] = arg;
}

extern "C" int clift_nearest_centroid(const float* feat, int ldf, int E, const float* centroids, int K, const unsigned char* valid, long n,
                                      int* labels, clift_stream_t s) {
    CLIFT_REQUIRE(K >= 1 && E >= 1, "clift_nearest_centroid: need K >= 1 and E >= 1");
    if (n <= 0) return 0;
    k_nearest_centroid<<<cdiv(n, 256), 256, 0, as_stream(s)>>>(feat, ldf, E, centroids, K, valid, n, labels);
    return clift_check_launch("clift_nearest_centroid");
}
