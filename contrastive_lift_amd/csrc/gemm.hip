// gemm.hip -- nn.Linear forward / dgrad / wgrad on the gfx950 matrix cores, fp32 in / fp32 accumulate.
//
//   C[m][n] (+)= act( sum_k A(m,k) B(n,k) + bias[n] ) * (mask[m][n] > 0)
//
// v_mfma_f32_32x32x2_f32: per wave a 32x32 tile, lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// exact fp32 (bit-identical to an fmaf chain), 64 cycles per instruction per SIMD = 256 FLOP/clk/CU.
// Block = WM x WN waves, each owning a (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles.  Operand tiles
// are staged global -> registers -> LDS (images described at lds_floats below: 16-byte stores for both operand
// orientations, 16-byte fragment reads for row-major operands) with the next k-tile's global loads in flight under the
// MFMAs of the current one; full tiles load without bounds checks or clamps.
// Measured (profiles/r01_gemm_pmc_notes.txt): the k-loop sustains ~128 TFLOP/s (4096^3); at the K = 256 of the MLP
// layers a large K-independent share remained, traced to the EPILOGUE being store-issue-bound: 64 four-byte stores per
// lane.  The forward/dgrad kernels therefore issue the MFMAs with swapped operands (SWAP) so that every lane owns four
// consecutive output columns and stores / mask-loads 16 bytes at a time (+10..20 % per launch).
//
// Used for (reference tensoRF.py): basis Linear :65, appearance MLP :393-397, instance MLPs :475-491,
// semantic MLP :576-582 -- forward (A = activations, B = weight (out,in)), dgrad (B transposed), wgrad
// (both transposed, reduction over the sample dimension split over blockIdx.z with atomic accumulation).
#include <cstring>
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(gemm)

constexpr int BK = 32;

// LDS image of one operand tile (ROWS = BM or BN, BK = 32 deep):
//   trans operand (memory [k][row]):  [k][ROWS + 4]   -- 16-byte stores along the row index, fragment = 4 scalar reads
//   plain operand (memory [row][k]):  [row][BK + 4]   -- 16-byte stores AND 16-byte fragment reads (pitch 36 floats: the
//                                                        16 lanes of a ds_read_b128 group land on 16 distinct 4-bank slots)
// Both serve the same k-interleaved MFMA order: within a group of 8 consecutive k, step s (0..3) multiplies k = s on
// lanes 0-31 and k = 4 + s on lanes 32-63, so a lane's four fragment values are 4 consecutive k of its row.
template <bool TR>
__host__ __device__ constexpr int lds_floats(int rows) { return TR ? BK * (rows + 4) : rows * (BK + 4); }

// Stage one operand tile (ROWS = BM or BN along m, BK along k) from global into registers.
//   non-trans: element (m,k) at P[m*ld + k]  -> float4 along k
//   trans    : element (m,k) at P[k*ld + m]  -> float4 along m
template <int ROWS, int NT, bool TR>
struct Stager {
    static constexpr int NV = (ROWS * BK / 4 + NT - 1) / NT;  // float4 per thread
    float4 v[NV];

    // full tiles (k0 + BK <= klim and m0 + ROWS <= mlim): no bounds checks and no index clamps -- a clamp per load costs
    // a 64-bit address each and pushed the forward kernel past 128 VGPRs.
    static __device__ __forceinline__ bool fast_ok(int m0, int mlim, int k0, int klim) {
        return (k0 + BK <= klim) && (m0 + ROWS <= mlim);
    }
    __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int m0, int mlim, int k0, int tid) {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int e = tid + p * NT;
            if ((ROWS * BK / 4) % NT == 0 || e < ROWS * BK / 4) {
                if (!TR) {
                    const int m = m0 + e / (BK / 4), k = k0 + (e % (BK / 4)) * 4;
                    v[p] = *reinterpret_cast<const float4*>(P + (size_t)m * ld + k);
                } else {
                    const int k = k0 + e / (ROWS / 4), m = m0 + (e % (ROWS / 4)) * 4;
                    v[p] = *reinterpret_cast<const float4*>(P + (size_t)k * ld + m);
                }
            }
        }
    }
    __device__ __forceinline__ void load_any(const float* __restrict__ P, int ld, int m0, int mlim, int k0, int klim, int tid) {
        if (fast_ok(m0, mlim, k0, klim)) load_fast(P, ld, m0, mlim, k0, tid);
        else load(P, ld, m0, mlim, k0, klim, tid);
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int m0, int mlim, int k0, int klim, int tid) {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int e = tid + p * NT;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < ROWS * BK / 4) {
                if (!TR) {
                    const int m = m0 + e / (BK / 4), k = k0 + (e % (BK / 4)) * 4;
                    if (m < mlim) {
                        const float* q = P + (size_t)m * ld + k;
                        if (k + 3 < klim) x = *reinterpret_cast<const float4*>(q);
                        else {
                            if (k < klim) x.x = q[0];
                            if (k + 1 < klim) x.y = q[1];
                            if (k + 2 < klim) x.z = q[2];
                        }
                    }
                } else {
                    const int k = k0 + e / (ROWS / 4), m = m0 + (e % (ROWS / 4)) * 4;
                    if (k < klim) {
                        const float* q = P + (size_t)k * ld + m;
                        if (m + 3 < mlim) x = *reinterpret_cast<const float4*>(q);
                        else {
                            if (m < mlim) x.x = q[0];
                            if (m + 1 < mlim) x.y = q[1];
                            if (m + 2 < mlim) x.z = q[2];
                        }
                    }
                }
            }
            v[p] = x;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ L, int tid) const {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int e = tid + p * NT;
            if ((ROWS * BK / 4) % NT == 0 || e < ROWS * BK / 4) {
                if (!TR) {
                    const int m = e / (BK / 4), k = (e % (BK / 4)) * 4;
                    *reinterpret_cast<float4*>(L + m * (BK + 4) + k) = v[p];
                } else {
                    const int k = e / (ROWS / 4), m = (e % (ROWS / 4)) * 4;
                    *reinterpret_cast<float4*>(L + k * (ROWS + 4) + m) = v[p];
                }
            }
        }
    }
    // fragment of row `row` for the 8-k group starting at kk8, lane half h: values for MFMA steps s = 0..3
    static __device__ __forceinline__ float4 frag(const float* __restrict__ L, int row, int kk8, int h) {
        if (!TR) return *reinterpret_cast<const float4*>(L + row * (BK + 4) + kk8 + 4 * h);
        const float* q = L + (kk8 + 4 * h) * (ROWS + 4) + row;
        return make_float4(q[0], q[ROWS + 4], q[2 * (ROWS + 4)], q[3 * (ROWS + 4)]);
    }
};

template <int BM, int BN, int WM, int WN, bool AT, bool BT, bool SWAP>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8 ? 4 : 1)) void k_gemm(GemmP g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LSA = BM + 4;      // k-row pitch of a transposed A image (column sums read it)
    __shared__ __attribute__((aligned(16))) float lds[lds_floats<AT>(BM) + lds_floats<BT>(BN)];
    float* As = lds;
    float* Bs = lds + lds_floats<AT>(BM);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // sample rows run along K in the weight-gradient form (A transposed), along M otherwise: clamp them to the true count of a
    // sync-free step (clift_dev.h, g_rows_limit); tiles / splits past it find nothing to do
    if (AT) g.K = limit_rows(g.K); else g.M = limit_rows(g.M);
    const int ntn = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const bool do_colsum = AT && g.colsum != nullptr && n0 == 0;
    float csum = 0.f;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    Stager<BM, NT, AT> sa;
    Stager<BN, NT, BT> sb;
    sa.load_any(g.A, g.lda, m0, g.M, kbeg, kend, tid);
    sb.load_any(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        sa.store(As, tid);
        sb.store(Bs, tid);
        __syncthreads();
        if (AT && do_colsum && tid < BM) {
#pragma unroll 8
            for (int kk = 0; kk < BK; ++kk) csum += As[kk * LSA + tid];
        }
        if (k0 + BK < kend) {
            sa.load_any(g.A, g.lda, m0, g.M, k0 + BK, kend, tid);
            sb.load_any(g.B, g.ldb, n0, g.N, k0 + BK, kend, tid);
        }
        const int arow = wm * (BM / WM) + li, brow = wn * (BN / WN) + li;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            float4 a[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = Stager<BM, NT, AT>::frag(As, arow + t * 32, kk, lh);
#pragma unroll
            for (int y = 0; y < TN; ++y) {           // one B fragment live at a time (register pressure)
                const float4 b = Stager<BN, NT, BT>::frag(Bs, brow + y * 32, kk, lh);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                    for (int x = 0; x < TM; ++x) {
                        const float av = sidx == 0 ? a[x].x : sidx == 1 ? a[x].y : sidx == 2 ? a[x].z : a[x].w;
                        const float bv = sidx == 0 ? b.x : sidx == 1 ? b.y : sidx == 2 ? b.z : b.w;
                        acc[x][y] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[x][y], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x][y], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    gemm_epilogue<BM, BN, WM, WN, SWAP>(g, acc, m0, n0, wm, wn, li, lh, tid, do_colsum, csum);
}

template <int BM, int BN, int WM, int WN, bool AT, bool BT>
static int launch_one(const GemmP& p, int splits, hipStream_t st) {
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, splits);
    const bool vec = gemm_vector_epilogue_ok(p);
    if (!vec) k_gemm<BM, BN, WM, WN, AT, BT, false><<<grid, dim3(WM * WN * 64), 0, st>>>(p);
    else k_gemm<BM, BN, WM, WN, AT, BT, true><<<grid, dim3(WM * WN * 64), 0, st>>>(p);
    return clift_check_launch("clift_gemm");
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmP& p, int a_trans, int b_trans, int splits, hipStream_t st) {
    if (!a_trans && !b_trans) return launch_one<BM, BN, WM, WN, false, false>(p, splits, st);
    if (!a_trans && b_trans) return launch_one<BM, BN, WM, WN, false, true>(p, splits, st);
    if (a_trans && b_trans) return launch_one<BM, BN, WM, WN, true, true>(p, splits, st);
    return launch_one<BM, BN, WM, WN, true, false>(p, splits, st);
}

extern "C" long clift_gemm_workspace_bytes(int N, int K) { return clift_gemm_split_workspace_bytes(N, K); }

extern "C" int clift_gemm(const clift_gemm_t* h, clift_stream_t s) {
    CLIFT_REQUIRE(h->M >= 0 && h->N >= 0 && h->K >= 0, "clift_gemm: negative dimension");
    if (h->M == 0 || h->N == 0) return 0;
    CLIFT_REQUIRE((h->a_bf16 || h->lda % 4 == 0) && (h->b_bf16 || h->ldb % 4 == 0), "clift_gemm: lda/ldb of fp32 operands must be multiples of 4 (got %d, %d)", h->lda, h->ldb);
    CLIFT_REQUIRE(((uintptr_t)h->A & 15) == 0 && ((uintptr_t)h->B & 15) == 0, "clift_gemm: A/B must be 16-byte aligned");
    int splits = h->split_k > 1 ? h->split_k : 1;
    CLIFT_REQUIRE(splits == 1 || h->accumulate, "clift_gemm: split_k > 1 requires accumulate");
    CLIFT_REQUIRE(splits == 1 || (!h->bias && !h->act && !h->mask), "clift_gemm: split_k > 1 excludes bias/act/mask");
    CLIFT_REQUIRE(!h->colsum || h->a_trans, "clift_gemm: colsum requires a_trans (wgrad form)");
    GemmP p;
    p.M = h->M; p.N = h->N; p.K = h->K;
    p.A = h->A; p.lda = h->lda; p.B = h->B; p.ldb = h->ldb; p.C = h->C; p.ldc = h->ldc;
    p.bias = h->bias; p.act = h->act; p.mask = h->mask; p.ldmask = h->ldmask; p.accumulate = h->accumulate;
    p.c_trans = h->c_trans; p.colsum = h->colsum;
    p.a_bf16 = h->a_bf16; p.b_bf16 = h->b_bf16; p.c_bf16 = h->c_bf16; p.mask_bf16 = h->mask_bf16;
    p.sign_bits = (unsigned char*)h->sign_bits;
    CLIFT_REQUIRE(!h->sign_bits || (h->precision == 2 && !h->a_trans && h->N == 256 && h->K == 256 && splits == 1 && !h->accumulate && !h->c_trans &&
                                     h->lda % 4 == 0 && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 && getenv("CLIFT_X6_TILED") == nullptr &&
                                     ((!h->b_trans && !h->mask) || (h->b_trans && !h->mask && !h->bias && h->act == 0))),
                  "clift_gemm: sign_bits goes with the persistent fp32x6 256 x 256 forms only (forward: written; dgrad with mask = NULL: read)");
    CLIFT_REQUIRE(h->precision == 1 || !(h->a_bf16 || h->b_bf16 || h->c_bf16 || h->mask_bf16),
                  "clift_gemm: bf16-stored tensors (a/b/c/mask_bf16) are only supported with precision 1");
    CLIFT_REQUIRE(!h->c_bf16 || (!h->accumulate && !h->c_trans), "clift_gemm: a bf16-stored output cannot be accumulated into or transposed");
    int kper = cdiv(cdiv(h->K, splits), BK) * BK;
    if (kper < BK) kper = BK;
    splits = cdiv(h->K, kper);
    if (splits < 1) splits = 1;
    p.k_per_split = kper;
    hipStream_t st = as_stream(s);
    CLIFT_REQUIRE(h->precision >= 0 && h->precision <= 2, "clift_gemm: precision must be 0 (fp32), 1 (bf16 operands) or 2 (fp32x6 split), got %d", h->precision);
    if (h->precision == 1) return clift_gemm_bf16_launch(p, h->a_trans, h->b_trans, splits, st);
    // fp32x6, the 256 x 256 hidden layers: persistent split kernels (layer_x6.hip), every M (a row's bits must not depend on its launch)
    if (h->precision == 2 && !h->a_trans && h->N == 256 && h->K == 256 && splits == 1 && !h->accumulate && !h->c_trans && h->lda % 4 == 0 &&
        h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 && getenv("CLIFT_X6_TILED") == nullptr &&
        ((!h->b_trans && !h->mask) ||
         (h->b_trans && h->mask && !h->bias && h->act == 0 && h->ldmask % 4 == 0 && (((uintptr_t)h->mask) & 15) == 0) ||
         (h->b_trans && !h->mask && h->sign_bits && !h->bias && h->act == 0)))
        return clift_layer_x6_launch(p, h->b_trans, st);
    // fp32x6 weight gradient of the 256 x 256 layers (layer_x6w.hip)
    if (h->precision == 2 && h->a_trans && h->b_trans && h->M == 256 && h->N == 256 && h->K >= 4096 && h->accumulate && !h->c_trans && !h->bias && !h->mask &&
        h->act == 0 && h->lda % 4 == 0 && h->ldb % 4 == 0 && (((uintptr_t)h->A) & 15) == 0 && (((uintptr_t)h->B) & 15) == 0)
        return clift_wgrad_x6_launch(p, st);
    // fp32x6, the 128-wide appearance layers (layer_n6.hip): forward K = 128 / 160 -> 128 (bias, ReLU), masked input gradient 128 -> 128, unmasked
    // input gradient 128 -> 160, every M (a row's bits must not depend on its launch); and their weight gradients 128 x {128, 160}
    if (h->precision == 2 && !h->a_trans && splits == 1 && !h->accumulate && !h->c_trans && h->lda % 4 == 0 && h->lda >= h->K && h->ldc % 4 == 0 &&
        (((uintptr_t)h->C) & 15) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr && getenv("CLIFT_X6_TILED") == nullptr) {
        const bool fwd = !h->b_trans && h->N == 128 && (h->K == 128 || h->K == 160) && !h->mask && h->ldb >= h->K && h->ldc >= 128 && (h->act == 0 || h->act == 1);
        const bool dg_m = h->b_trans && h->N == 128 && h->K == 128 && h->mask && !h->bias && h->act == 0 && h->ldmask % 4 == 0 && h->ldmask >= 128 &&
                          (((uintptr_t)h->mask) & 15) == 0 && h->ldb >= 128 && h->ldc >= 128;
        const bool dg_u = h->b_trans && h->N == 160 && h->K == 128 && !h->mask && !h->bias && h->act == 0 && h->ldb >= 160 && h->ldc >= 160;
        if (fwd || dg_m || dg_u) return clift_layer_n6_launch(p, h->b_trans, st);
    }
    if (h->precision == 2 && h->a_trans && h->b_trans && h->M == 128 && (h->N == 128 || h->N == 160) && h->K >= 1 && h->accumulate && !h->c_trans && !h->bias &&
        !h->mask && h->act == 0 && h->lda % 4 == 0 && h->lda >= 128 && h->ldb % 4 == 0 && h->ldb >= h->N && h->ldc >= h->N &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr && getenv("CLIFT_X6_TILED") == nullptr)
        return clift_wgrad_n6_launch(p, st);
    // fp32x6: forward / dgrad forms (row-major A, one weight-sized B); any other shape takes the tiled split kernel
    if (h->precision == 2 && !h->a_trans && !h->accumulate && splits == 1 && !h->c_trans && (long)h->N * h->K <= (1L << 22)) {
        const long need = clift_gemm_split_workspace_bytes(h->N, h->K);
        CLIFT_REQUIRE(h->workspace != nullptr && h->workspace_bytes >= need && (((uintptr_t)h->workspace & 15) == 0),
                      "clift_gemm: precision 2 needs a 16-byte aligned workspace of %ld bytes (got %ld)", need, h->workspace_bytes);
        return clift_gemm_split_launch(p, h->a_trans, h->b_trans, h->workspace, st);
    }
    // (every M: a row's result must not depend on how many rows share its launch -- a frame rendered in row tiles over several GPUs
    // has to be bit-identical to the unsharded render, and the tiled kernel sums k in a different order)
    if (h->precision == 0 && !h->a_trans && !h->b_trans && h->N == 256 && h->K == 256 && splits == 1 && !h->accumulate && !h->c_trans &&
        !h->mask && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_layer_f32_launch(p, 0, st);                    // persistent forward layer: weights in registers, LDS-DMA row stream
    if (h->precision == 0 && !h->a_trans && h->b_trans && h->N == 256 && h->K == 256 && h->M >= 4096 && splits == 1 && !h->accumulate && !h->c_trans &&
        h->mask && !h->bias && h->act == 0 && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 && h->ldmask % 4 == 0 && (((uintptr_t)h->mask) & 15) == 0 &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_layer_f32_launch(p, 1, st);                    // persistent masked dgrad of the same layers
    if (h->precision == 0 && !h->a_trans && h->b_trans && (h->N == 256 || h->N == 128) && h->K <= 32 && h->K <= h->lda && h->lda <= 32 && h->M >= 4096 && splits == 1 &&
        !h->accumulate && !h->c_trans && h->mask && !h->bias && h->act == 0 && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 && h->ldmask % 4 == 0 &&
        (((uintptr_t)h->mask) & 15) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_dgrad_narrow_stream_launch(p, 0, st);          // output-layer dgrad: a stream over the mask and the result
    if (h->precision == 0 && !h->a_trans && h->b_trans && h->N <= 256 && h->N % 4 == 0 && h->N > 32 && h->K <= 32 && h->K <= h->lda && h->lda <= 32 && h->M >= 4096 &&
        splits == 1 && !h->accumulate && !h->c_trans && !h->mask && !h->bias && h->act == 0 && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_dgrad_narrow_stream_launch(p, 0, st);          // unmasked narrow-K dgrad (appearance basis 27 -> 144): a stream over the result
    // (up to ~150 k rows: beyond that the split-K tiled launch, whose k-loop is long by then, is as fast or faster: 304 vs 332 us at 249 k)
    if (h->precision == 0 && h->a_trans && h->b_trans && h->M == 256 && h->N == 256 && h->K >= 4096 && h->accumulate && !h->c_trans && !h->bias && !h->mask &&
        h->act == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        // persistent 2-D weight gradient: 64 row ranges x four 128 x 128 quadrants, 64-row tiles (layer_n128.hip) -- 92 / 318 us at 62 k / 249 k rows
        // (measured against 256 x 64 slices with 32-row tiles: 95 / 340, and the split-K tiled launch: 127 / 349; profiles/r02_*)
        return clift_wgrad_f32_quads_launch(p, st);
    // weight gradients of the 128-wide appearance layers (128 x 128 and 128 x 160 results): persistent row-range stream
    if (h->precision == 0 && h->a_trans && h->b_trans && h->M == 128 && (h->N == 128 || h->N == 160) && h->ldb >= h->N && h->lda >= 128 && h->K >= 4096 &&
        h->accumulate && !h->c_trans && !h->bias && !h->mask && h->act == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_wgrad_n128_stream_launch(p, st);
    // the 128-wide layers of the appearance MLP (forward K = 128 / 160 with bias, masked dgrad K = 128): persistent, every M (see above)
    if (h->precision == 0 && !h->a_trans && h->N == 128 && ((h->K == 128) || (h->K == 160 && !h->b_trans)) && h->lda >= h->K && splits == 1 &&
        !h->accumulate && !h->c_trans && h->ldc % 4 == 0 && (((uintptr_t)h->C) & 15) == 0 &&
        ((!h->b_trans && !h->mask && h->ldb >= h->K) ||
         (h->b_trans && h->mask && !h->bias && h->act == 0 && h->ldmask % 4 == 0 && (((uintptr_t)h->mask) & 15) == 0 && h->ldb >= 128)) &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_layer_n128_launch(p, h->b_trans, st);
    // dX of the first appearance layer, (M x 128) (128 x 160), no mask: columns 0..127 through the persistent 128-wide dgrad kernel,
    // columns 128..159 through the tiled 32-column kernel (one tiled 128 x 256 launch: 183 us at 265 k rows; the pair: see DESIGN 5b)
    if (h->precision == 0 && !h->a_trans && h->b_trans && h->N == 160 && h->K == 128 && h->M >= 4096 && splits == 1 && !h->accumulate && !h->c_trans &&
        !h->mask && !h->bias && h->act == 0 && h->lda >= 128 && h->lda % 4 == 0 && h->ldb >= 160 && h->ldb % 4 == 0 && h->ldc >= 160 && h->ldc % 4 == 0 &&
        (((uintptr_t)h->C) & 15) == 0 && (((uintptr_t)h->A) & 15) == 0 && (((uintptr_t)h->B) & 15) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr) {
        GemmP p1 = p;
        p1.N = 128; p1.mask = p1.A; p1.ldmask = p1.lda; p1.act = 7;
        const int rc = clift_layer_n128_launch(p1, 1, st);
        if (rc) return rc;
        GemmP p2 = p;
        p2.N = 32; p2.B = p.B + 128; p2.C = p.C + 128;
        return launch_gemm<256, 32, 4, 1>(p2, 0, 1, 1, st);
    }
    if (h->N > 128) {
        return launch_gemm<128, 256, 2, 4>(p, h->a_trans, h->b_trans, splits, st);
    }
    if (h->N > 32) return launch_gemm<128, 128, 2, 2>(p, h->a_trans, h->b_trans, splits, st);
    return launch_gemm<256, 32, 4, 1>(p, h->a_trans, h->b_trans, splits, st);
}

// ============================================================================ K = 3 first layers
// out[m][n] = act(W[n][0] x + W[n][1] y + W[n][2] z + b[n]); pure store-bandwidth kernel (tensoRF.py:475,576).
// Thread = four consecutive output columns, kept for MANY rows: the 12 weights and 4 biases live in registers and a block strides
// over the rows with four rows in flight per thread (one broadcast 16-byte load of x and one 16-byte store per row).  The first
// version took one (row, column-quad) per thread and re-fetched its 16 coefficients every time: 18 memory instructions per
// 16-byte store, 3 TB/s of output.
__global__ __launch_bounds__(256) void k_linear_k3_fwd(const float* __restrict__ x4, const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ b, int M, int Nout, int relu,
                                                        float* __restrict__ out, int ldo, int out_bf16) {
    M = limit_rows(M);
    const int nq = Nout / 4, rpi = 256 / nq;                 // column quads per row; rows per block iteration (nq divides 256)
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, n = q * 4;
    float w0[4], w1[4], w2[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* w = W + (size_t)(n + j) * ldw;
        w0[j] = w[0]; w1[j] = w[1]; w2[j] = w[2]; bb[j] = b[n + j];
    }
    auto emit = [&](int m, const float4 x) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = fmaf(w2[j], x.z, fmaf(w1[j], x.y, fmaf(w0[j], x.x, bb[j])));
            o[j] = relu ? fmaxf(v, 0.f) : v;
        }
        if (out_bf16) {      // bf16-stored hidden activation (bf16 mode)
            const unsigned lo = (unsigned)float_to_bf16_bits(o[0]) | ((unsigned)float_to_bf16_bits(o[1]) << 16);
            const unsigned hi = (unsigned)float_to_bf16_bits(o[2]) | ((unsigned)float_to_bf16_bits(o[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + (size_t)m * ldo + n) = make_uint2(lo, hi);
        } else {
            *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    const long step = (long)gridDim.x * rpi;
    long m = (long)blockIdx.x * rpi + rl;
    for (; m + 3 * step < M; m += 4 * step) {
        const float4 xa = ld4(x4 + (size_t)m * 4), xb = ld4(x4 + (size_t)(m + step) * 4);
        const float4 xc = ld4(x4 + (size_t)(m + 2 * step) * 4), xd = ld4(x4 + (size_t)(m + 3 * step) * 4);
        emit((int)m, xa); emit((int)(m + step), xb); emit((int)(m + 2 * step), xc); emit((int)(m + 3 * step), xd);
    }
    for (; m < M; m += step) emit((int)m, ld4(x4 + (size_t)m * 4));
}

// any width (Nout / 4 not a divisor of 256): one (row, column quad) per thread
__global__ __launch_bounds__(256) void k_linear_k3_fwd_any(const float* __restrict__ x4, const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ b, int M, int Nout, int relu,
                                                            float* __restrict__ out, int ldo, int out_bf16) {
    const int nq = Nout / 4;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)limit_rows(M) * nq) return;
    const int m = (int)(gid / nq), n = (int)(gid % nq) * 4;
    const float4 x = ld4(x4 + (size_t)m * 4);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* w = W + (size_t)(n + j) * ldw;
        const float v = fmaf(w[2], x.z, fmaf(w[1], x.y, fmaf(w[0], x.x, b[n + j])));
        o[j] = relu ? fmaxf(v, 0.f) : v;
    }
    if (out_bf16) {
        const unsigned lo = (unsigned)float_to_bf16_bits(o[0]) | ((unsigned)float_to_bf16_bits(o[1]) << 16);
        const unsigned hi = (unsigned)float_to_bf16_bits(o[2]) | ((unsigned)float_to_bf16_bits(o[3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + (size_t)m * ldo + n) = make_uint2(lo, hi);
    } else {
        *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int clift_linear_k3_fwd(const float* x4, const float* W, int ldw, const float* b, int M, int Nout, int relu,
                                   float* out, int ldo, int out_bf16, clift_stream_t s) {
    CLIFT_REQUIRE(Nout % 4 == 0 && ldo % 4 == 0, "clift_linear_k3_fwd: Nout and ldo must be multiples of 4");
    if (M <= 0) return 0;
    if (Nout > 1024 || 256 % (Nout / 4) != 0) {
        k_linear_k3_fwd_any<<<cdiv((long)M * (Nout / 4), 256), 256, 0, as_stream(s)>>>(x4, W, ldw, b, M, Nout, relu, out, ldo, out_bf16);
        return clift_check_launch("clift_linear_k3_fwd");
    }
    const int rpi = 256 / (Nout / 4);
    const long want = ((long)M + rpi - 1) / rpi;
    const int blocks = (int)(want < 2048 ? want : 2048);             // 8 blocks per CU, each thread keeps its coefficients for ~M / 8192 rows
    k_linear_k3_fwd<<<blocks, 256, 0, as_stream(s)>>>(x4, W, ldw, b, M, Nout, relu, out, ldo, out_bf16);
    return clift_check_launch("clift_linear_k3_fwd");
}

// dW[n][0..2] += sum_m dH[m][n] x[m][:], db[n] += sum_m dH[m][n].
// Block = 256 columns x 4 row slabs (1024 threads): thread (slab, n) streams every 4th 64-row group of the block's rows with
// 8 independent loads in flight, the four slabs are folded through LDS and ONE thread per column issues the atomics.  (A
// 256-thread block per 512 rows left 8 waves per CU: one dependent 4-byte load per thread in flight, 95 us for 271 MB.)
__global__ __launch_bounds__(1024) void k_linear_k3_bwd(const float* __restrict__ x4, const float* __restrict__ dH, int ldh, int M,
                                                         int Nout, int rows_per_block, float* __restrict__ dW, int ldw, float* __restrict__ db,
                                                         int dh_bf16) {
    const int col = threadIdx.x & 255, slab = threadIdx.x >> 8;
    const int n = blockIdx.y * 256 + col;
    M = limit_rows(M);
    const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
    __shared__ float4 red[3][256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < Nout) {
        auto ldh_at = [&](int m) {
            return dh_bf16 ? bf16_bits_to_float(reinterpret_cast<const unsigned short*>(dH)[(size_t)m * ldh + n]) : dH[(size_t)m * ldh + n];
        };
        for (int mc = mb + slab * 8; mc < me; mc += 32) {      // 8 consecutive rows per step, the 4 slabs interleaved
            float d[8];
            float4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool ok = mc + q < me;
                d[q] = ok ? ldh_at(mc + q) : 0.f;
                x[q] = ok ? ld4(x4 + (size_t)(mc + q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);      // uniform per slab: broadcast load
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { a0 = fmaf(d[q], x[q].x, a0); a1 = fmaf(d[q], x[q].y, a1); a2 = fmaf(d[q], x[q].z, a2); a3 += d[q]; }
        }
    }
    if (slab > 0) red[slab - 1][col] = make_float4(a0, a1, a2, a3);
    __syncthreads();
    if (slab == 0 && n < Nout) {
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) { const float4 r = red[s_][col]; a0 += r.x; a1 += r.y; a2 += r.z; a3 += r.w; }
        unsafeAtomicAdd(dW + (size_t)n * ldw + 0, a0);
        unsafeAtomicAdd(dW + (size_t)n * ldw + 1, a1);
        unsafeAtomicAdd(dW + (size_t)n * ldw + 2, a2);
        if (db) unsafeAtomicAdd(db + n, a3);
    }
}

extern "C" int clift_linear_k3_bwd(const float* x4, const float* dH, int ldh, int M, int Nout, float* dW, int ldw, float* db,
                                   int dh_bf16, clift_stream_t s) {
    if (M <= 0) return 0;
    if (Nout == 256 && M >= 4096 && db && (((uintptr_t)x4) & 15) == 0 && (((uintptr_t)dH) & 15) == 0 && ldh % (dh_bf16 ? 8 : 4) == 0 &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_k3_bwd_stream_launch(x4, dH, ldh, M, dW, ldw, db, dh_bf16, as_stream(s));      // matrix-core stream over dH (narrow_stream.hip)
    const int rpb = 512;
    k_linear_k3_bwd<<<dim3(cdiv(M, rpb), cdiv(Nout, 256)), 1024, 0, as_stream(s)>>>(x4, dH, ldh, M, Nout, rpb, dW, ldw, db, dh_bf16);
    return clift_check_launch("clift_linear_k3_bwd");
}

// ============================================================================ narrow wgrad (out_features <= 32)
// gW[c][j] += sum_s dY[s][c] X[s][j],  gb[c] += sum_s dY[s][c]   for c < NO <= 32, j < ni.
// A 22 x 256 output over 265 k samples is a streaming reduction (it reads X once: 271 MB), not matrix-core work: an MFMA
// tile would be 83 % padding.  Thread = input feature j (coalesced 1 KB rows of X), the NO partial sums live in
// registers, dY rows are broadcast from LDS; one atomic per (c, j) per block.
template <int NO>
__global__ __launch_bounds__(256) void k_wgrad_narrow(const float* __restrict__ dY, int ldd, int no, const float* __restrict__ X, int ldx, int ni,
                                                       int M, int rows_per_block, float* __restrict__ gW, int ldw, float* __restrict__ gb, int x_bf16) {
    __shared__ __attribute__((aligned(16))) float ds[64 * NO];
    const unsigned short* X16 = reinterpret_cast<const unsigned short*>(X);
    const int j = blockIdx.y * 256 + threadIdx.x;
    M = limit_rows(M);
    const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
    float acc[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) acc[c] = 0.f;
    float bsum = 0.f;      // thread c < no of blockIdx.y == 0 also accumulates the bias gradient
    for (int mc = mb; mc < me; mc += 64) {
        const int lim = min(64, me - mc);
        __syncthreads();
        for (int e = threadIdx.x; e < lim * NO; e += 256) {
            const int r = e / NO, c = e - r * NO;
            ds[e] = (c < no) ? dY[(size_t)(mc + r) * ldd + c] : 0.f;
        }
        __syncthreads();
        if (j < ni) {
            int r = 0;
            for (; r + 8 <= lim; r += 8) {        // 8 rows in flight per thread: independent 1 KB-per-wave loads (latency-bound otherwise)
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = x_bf16 ? bf16_bits_to_float(X16[(size_t)(mc + r + q) * ldx + j]) : X[(size_t)(mc + r + q) * ldx + j];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int c = 0; c < NO; c += 4) {
                        const float4 d = *reinterpret_cast<const float4*>(ds + (r + q) * NO + c);   // LDS broadcast
                        acc[c] = fmaf(d.x, x[q], acc[c]); acc[c + 1] = fmaf(d.y, x[q], acc[c + 1]);
                        acc[c + 2] = fmaf(d.z, x[q], acc[c + 2]); acc[c + 3] = fmaf(d.w, x[q], acc[c + 3]);
                    }
            }
            for (; r < lim; ++r) {
                const float xv = x_bf16 ? bf16_bits_to_float(X16[(size_t)(mc + r) * ldx + j]) : X[(size_t)(mc + r) * ldx + j];
#pragma unroll
                for (int c = 0; c < NO; ++c) acc[c] = fmaf(ds[r * NO + c], xv, acc[c]);
            }
        }
        if (blockIdx.y == 0 && threadIdx.x < no && gb)
            for (int r = 0; r < lim; ++r) bsum += ds[r * NO + threadIdx.x];
    }
    if (j < ni) {
#pragma unroll
        for (int c = 0; c < NO; ++c)
            if (c < no) unsafeAtomicAdd(gW + (size_t)c * ldw + j, acc[c]);
    }
    if (blockIdx.y == 0 && threadIdx.x < no && gb) unsafeAtomicAdd(gb + threadIdx.x, bsum);
}

extern "C" int clift_wgrad_narrow(const float* dY, int ldd, int no, const float* X, int ldx, int ni, int M, float* gW, int ldw, float* gb,
                                  int x_bf16, clift_stream_t s) {
    CLIFT_REQUIRE(no >= 1 && no <= 32, "clift_wgrad_narrow: out_features must be in [1,32] (got %d)", no);
    if (M <= 0 || ni <= 0) return 0;
    if ((ni == 256 || (!x_bf16 && ni >= 32 && ni < 256 && ni % 4 == 0)) && M >= 4096 && ldd % 4 == 0 && ldd >= no && ldd <= 32 &&
        (((uintptr_t)dY) & 15) == 0 && (((uintptr_t)X) & 15) == 0 && ldx % (x_bf16 ? 8 : 4) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_wgrad_narrow_stream_launch(dY, ldd, no, X, ldx, ni, M, gW, ldw, gb, x_bf16, as_stream(s));   // matrix-core stream over X
    const int rpb = 128;        // (insensitive between 64 and 520 rows per block: the kernel is FMA-bound, not launch-shape-bound)
    const dim3 grid(cdiv(M, rpb), cdiv(ni, 256));
    hipStream_t st = as_stream(s);
    if (no <= 4) k_wgrad_narrow<4><<<grid, 256, 0, st>>>(dY, ldd, no, X, ldx, ni, M, rpb, gW, ldw, gb, x_bf16);
    else if (no <= 8) k_wgrad_narrow<8><<<grid, 256, 0, st>>>(dY, ldd, no, X, ldx, ni, M, rpb, gW, ldw, gb, x_bf16);
    else if (no <= 16) k_wgrad_narrow<16><<<grid, 256, 0, st>>>(dY, ldd, no, X, ldx, ni, M, rpb, gW, ldw, gb, x_bf16);
    else if (no <= 24) k_wgrad_narrow<24><<<grid, 256, 0, st>>>(dY, ldd, no, X, ldx, ni, M, rpb, gW, ldw, gb, x_bf16);   // 22 ScanNet classes
    else k_wgrad_narrow<32><<<grid, 256, 0, st>>>(dY, ldd, no, X, ldx, ni, M, rpb, gW, ldw, gb, x_bf16);
    return clift_check_launch("clift_wgrad_narrow");
}

// db[n] += sum_m dY[m][n]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dY, int ld, int M, int N, int rows_per_block,
                                                 float* __restrict__ db) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
    float a = 0.f;
    for (int m = mb; m < me; ++m) a += dY[(size_t)m * ld + n];
    unsafeAtomicAdd(db + n, a);
}

extern "C" int clift_colsum(const float* dY, int ld, int M, int N, float* db, clift_stream_t s) {
    if (M <= 0 || N <= 0) return 0;
    const int rpb = 256;
    k_colsum<<<dim3(cdiv(M, rpb), cdiv(N, 256)), 256, 0, as_stream(s)>>>(dY, ld, M, N, rpb, db);
    return clift_check_launch("clift_colsum");
}

// ============================================================================ row activations
// kind 1: sigmoid (appearance head, tensoRF.py:385,410); kind 2: softmax over the row (semantic head, :37,593).
// Lane group of G = 2^lg >= max(C, ld) lanes per row, lane = column: coalesced row accesses, group reductions by xor
// shuffles (one thread per row walked 22 strided floats per lane: 43-54 us per launch at 265 k rows, ~1 TB/s).
__device__ __forceinline__ float group_max(float v, int G) { for (int d = 1; d < G; d <<= 1) v = fmaxf(v, __shfl_xor(v, d)); return v; }
__device__ __forceinline__ float group_sum(float v, int G) { for (int d = 1; d < G; d <<= 1) v += __shfl_xor(v, d); return v; }

__global__ __launch_bounds__(256) void k_rows_act_fwd(const float* __restrict__ pre, int ldp, int M, int C, int kind,
                                                       float* __restrict__ out, int ldo, int lg) {
    const int G = 1 << lg;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long m = gid >> lg;
    const int c = (int)(gid & (G - 1));
    const bool row = m < limit_rows(M), on = row && c < C;        // every lane of a group takes part in the shuffles
    const float x = on ? pre[(size_t)m * ldp + c] : -INFINITY;
    float v;
    if (kind == 1) {
        v = 1.f / (1.f + expf(-x));
    } else {
        const float mx = group_max(x, G);
        const float e = on ? expf(x - mx) : 0.f;
        const float sum = group_sum(e, G);
        v = e * (1.f / sum);
    }
    if (on) out[(size_t)m * ldo + c] = v;
}

extern "C" int clift_rows_act_fwd(const float* pre, int ldp, int M, int C, int kind, float* out, int ldo, clift_stream_t s) {
    CLIFT_REQUIRE(kind == 1 || kind == 2, "clift_rows_act_fwd: kind must be 1 (sigmoid) or 2 (softmax)");
    CLIFT_REQUIRE(C >= 1 && C <= 64, "clift_rows_act_fwd: C must be in [1,64] (got %d)", C);
    if (M <= 0) return 0;
    int lg = 0;
    while ((1 << lg) < C) ++lg;
    k_rows_act_fwd<<<cdiv((long)M << lg, 256), 256, 0, as_stream(s)>>>(pre, ldp, M, C, kind, out, ldo, lg);
    return clift_check_launch("clift_rows_act_fwd");
}

__global__ __launch_bounds__(256) void k_rows_act_bwd(const float* __restrict__ out, int ldo, const float* __restrict__ dout, int lddo,
                                                       int M, int C, int kind, float* __restrict__ dpre, int ldd, int lg) {
    const int G = 1 << lg;                            // G >= max(C, ldd): lanes C..ldd-1 zero the alignment padding
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long m = gid >> lg;
    const int c = (int)(gid & (G - 1));
    const bool row = m < limit_rows(M), on = row && c < C;
    const float g = on ? dout[(size_t)m * lddo + c] : 0.f;
    const float o = (on && kind != 0) ? out[(size_t)m * ldo + c] : 0.f;
    float d;
    if (kind == 0) d = g;
    else if (kind == 1) d = g * o * (1.f - o);
    else {
        const float dot = group_sum(g * o, G);
        d = o * (g - dot);
    }
    if (row && c < ldd) dpre[(size_t)m * ldd + c] = on ? d : 0.f;   // keep the alignment padding zero (it is a GEMM K-operand)
}

extern "C" int clift_rows_act_bwd(const float* out, int ldo, const float* dout, int lddo, int M, int C, int kind, float* dpre,
                                  int ldd, clift_stream_t s) {
    CLIFT_REQUIRE(kind >= 0 && kind <= 2, "clift_rows_act_bwd: kind must be 0 (identity), 1 (sigmoid) or 2 (softmax)");
    CLIFT_REQUIRE(C >= 1 && ldd >= C && ldd <= 64, "clift_rows_act_bwd: need 1 <= C <= ldd <= 64 (got C=%d ldd=%d)", C, ldd);
    if (M <= 0) return 0;
    int lg = 0;
    while ((1 << lg) < ldd) ++lg;
    k_rows_act_bwd<<<cdiv((long)M << lg, 256), 256, 0, as_stream(s)>>>(out, ldo, dout, lddo, M, C, kind, dpre, ldd, lg);
    return clift_check_launch("clift_rows_act_bwd");
}
