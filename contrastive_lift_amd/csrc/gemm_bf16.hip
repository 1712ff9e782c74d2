// gemm_bf16.hip -- the same nn.Linear forward / dgrad / wgrad contract as gemm.hip with bf16 OPERANDS and fp32
// accumulation ("bf16 mode", BASELINE config 3): an operand stored as fp32 is rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on its way into LDS; an operand already STORED as bf16 (a_bf16 / b_bf16: the hidden activations and their
// gradients, which then cost half the HBM bytes) is copied; the output and the ReLU mask can be bf16-stored too (c_bf16 /
// mask_bf16).  The products run on
// v_mfma_f32_32x32x16_bf16 -- 16x the fp32 MFMA rate, so these launches are bound by streaming the fp32 activations
// (HBM), not by the matrix cores.  Result == fp32 GEMM of the bf16-rounded operands up to summation order.
//
// Fragment layout of the 32x32x16 bf16 MFMA: lane l supplies row (l & 31) and the 8 consecutive k = 8 (l >> 5) .. +7 of
// both operands; C/D layout identical to the fp32 32x32x2 form, hence the shared epilogue (gemm_common.h).
// LDS image of every operand tile: [row][32 k] bf16 with a 72-byte row pitch (18 dwords): an 8-k fragment is two
// conflict-free ds_read_b64 (rows r, r+1 sit 18 banks apart; 32 lanes x 2 banks cover the 64 banks once).
//   plain operand (memory [row][k]): float4 along k -> one 8-byte LDS store.
//   trans operand (memory [k][row]): a thread takes the float4s of rows 4 m4 .. +3 at k = 2 kp and 2 kp + 1 and writes four
//     packed k-pairs (4-byte stores); the lane -> (m4, kp) mapping puts the 32 lanes of a store group on 4 rows-of-four x 8
//     k-pairs = 32 distinct banks, at the price of 64-byte (not 128-byte) global segments per k row.
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BKH = 32;     // k depth of one staged tile
constexpr int LPD = 18;     // LDS row pitch in dwords (36 bf16)

static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}

template <int ROWS, int NT, bool TR>
struct StagerH {
    static constexpr int NE = TR ? ROWS * BKH / 8 : ROWS * BKH / 4;   // entries: k-pair x 4 rows | 4 k of one row
    static constexpr int NV = (NE + NT - 1) / NT;
    float4 v[TR ? 2 * NV : NV];

    static __device__ __forceinline__ bool fast_ok(int m0, int mlim, int k0, int klim) {
        return (k0 + BKH <= klim) && (m0 + ROWS <= mlim);     // full tile: no clamps, no guards
    }
    static __device__ __forceinline__ void coords(int e, int& a, int& b) {
        if (!TR) { a = e >> 3; b = (e & 7) * 4; }                                   // row, first k
        else { a = (((e >> 6) << 2) | (e & 3)) * 4; b = ((e >> 2) & 15) * 2; }       // first row, first k
    }
    __device__ __forceinline__ void load_any(const float* __restrict__ P, int ld, int m0, int mlim, int k0, int klim, int tid) {
        const bool fast = fast_ok(m0, mlim, k0, klim);
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int e = tid + p * NT;
            int a, b;
            coords(e, a, b);
            if (!TR) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (NE % NT == 0 || e < NE) {
                    if (fast) x = *reinterpret_cast<const float4*>(P + (size_t)(m0 + a) * ld + k0 + b);
                    else if (m0 + a < mlim) {
                        const int k = k0 + b;
                        const float* q = P + (size_t)(m0 + a) * ld + k;
                        if (k + 3 < klim) x = *reinterpret_cast<const float4*>(q);
                        else {
                            if (k < klim) x.x = q[0];
                            if (k + 1 < klim) x.y = q[1];
                            if (k + 2 < klim) x.z = q[2];
                        }
                    }
                }
                v[p] = x;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (NE % NT == 0 || e < NE) {
                        const int k = k0 + b + h, m = m0 + a;
                        const float* q = P + (size_t)k * ld + m;
                        if (fast) x = *reinterpret_cast<const float4*>(q);
                        else if (k < klim) {
                            if (m + 3 < mlim) x = *reinterpret_cast<const float4*>(q);
                            else {
                                if (m < mlim) x.x = q[0];
                                if (m + 1 < mlim) x.y = q[1];
                                if (m + 2 < mlim) x.z = q[2];
                            }
                        }
                    }
                    v[2 * p + h] = x;
                }
            }
        }
    }
    // ---- operand STORED as bf16 (activation storage of the bf16 mode): 8 elements per 16-byte entry.
    //   plain: entry = 8 consecutive k of one row              -> copied to LDS unchanged (two 8-byte stores)
    //   trans: entry = rows 8 r8 .. +7 at k = 2 kp and 2 kp + 1 -> eight packed k-pairs (4-byte stores); 2 row-groups x 16 k-pairs
    //          per 32 lanes = 32 distinct banks
    static constexpr int NEH = TR ? 2 * ROWS : 4 * ROWS;
    static constexpr int NVH = (NEH + NT - 1) / NT;
    static __device__ __forceinline__ void coords_h(int e, int& a, int& b) {
        if (!TR) { a = e >> 2; b = (e & 3) * 8; }                                    // row, first k
        else { a = (((e >> 5) << 1) | (e & 1)) * 8; b = ((e >> 1) & 15) * 2; }        // first row, first k
    }
    __device__ __forceinline__ void load_half(const unsigned short* __restrict__ P, int ld, int m0, int mlim, int k0, int klim, int tid) {
        const bool fast = fast_ok(m0, mlim, k0, klim) && (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(P) & 15) == 0);
        uint4* u = reinterpret_cast<uint4*>(v);
#pragma unroll
        for (int p = 0; p < NVH; ++p) {
            const int e = tid + p * NT;
            int a, b;
            coords_h(e, a, b);
#pragma unroll
            for (int h = 0; h < (TR ? 2 : 1); ++h) {
                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                if (NEH % NT == 0 || e < NEH) {
                    const unsigned short* q = TR ? P + (size_t)(k0 + b + h) * ld + m0 + a : P + (size_t)(m0 + a) * ld + k0 + b;
                    if (fast) x = *reinterpret_cast<const uint4*>(q);
                    else {                                   // ragged tile: element-wise with zero fill
                        unsigned w[4] = {0u, 0u, 0u, 0u};
                        const bool line_ok = TR ? (k0 + b + h < klim) : (m0 + a < mlim);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bool ok = line_ok && (TR ? (m0 + a + j < mlim) : (k0 + b + j < klim));
                            const unsigned val = ok ? (unsigned)q[j] : 0u;
                            w[j >> 1] |= val << (16 * (j & 1));
                        }
                        x = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
                u[TR ? 2 * p + h : p] = x;
            }
        }
    }
    __device__ __forceinline__ void store_half(unsigned* __restrict__ L, int tid) const {
        const uint4* u = reinterpret_cast<const uint4*>(v);
#pragma unroll
        for (int p = 0; p < NVH; ++p) {
            const int e = tid + p * NT;
            if (NEH % NT == 0 || e < NEH) {
                int a, b;
                coords_h(e, a, b);
                if (!TR) {
                    uint2* q = reinterpret_cast<uint2*>(L + a * LPD + (b >> 1));
                    q[0] = make_uint2(u[p].x, u[p].y);
                    q[1] = make_uint2(u[p].z, u[p].w);
                } else {
                    const uint4 lo = u[2 * p], hi = u[2 * p + 1];
                    const unsigned l4[4] = {lo.x, lo.y, lo.z, lo.w}, h4[4] = {hi.x, hi.y, hi.z, hi.w};
                    unsigned* q = L + a * LPD + (b >> 1);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned el = (l4[j >> 1] >> (16 * (j & 1))) & 0xffffu, eh = (h4[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                        q[j * LPD] = el | (eh << 16);
                    }
                }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned* __restrict__ L, int tid) const {
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int e = tid + p * NT;
            if (NE % NT == 0 || e < NE) {
                int a, b;
                coords(e, a, b);
                if (!TR) {
                    *reinterpret_cast<uint2*>(L + a * LPD + (b >> 1)) = make_uint2(pack2(v[p].x, v[p].y), pack2(v[p].z, v[p].w));
                } else {
                    const float4 lo = v[2 * p], hi = v[2 * p + 1];
                    unsigned* q = L + a * LPD + (b >> 1);
                    q[0] = pack2(lo.x, hi.x); q[LPD] = pack2(lo.y, hi.y);
                    q[2 * LPD] = pack2(lo.z, hi.z); q[3 * LPD] = pack2(lo.w, hi.w);
                }
            }
        }
    }
};

static __device__ __forceinline__ bf16x8 frag_h(const unsigned* __restrict__ L, int row, int kk16, int h) {
    const uint2* q = reinterpret_cast<const uint2*>(L + row * LPD + (kk16 >> 1) + 4 * h);
    const uint2 lo = q[0], hi = q[1];
    const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return __builtin_bit_cast(bf16x8, u);
}

// ST: the STREAMED operands are bf16-stored -- A always, B too in the wgrad form (AT && BT); weights (B of the forward / dgrad
// forms) are always fp32-stored.  A compile-time switch: carrying both load paths in one kernel spilled and cost 10 %.
template <int BM, int BN, int WM, int WN, bool AT, bool BT, bool SWAP, bool ST>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8 ? 4 : 1)) void k_gemm_bf16(GemmP g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) unsigned lds[(BM + BN) * LPD];
    unsigned* As = lds;
    unsigned* Bs = lds + BM * LPD;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

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

    StagerH<BM, NT, AT> sa;
    StagerH<BN, NT, BT> sb;
    constexpr bool a16 = ST, b16 = ST && AT && BT;
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* B16 = reinterpret_cast<const unsigned short*>(g.B);
    if (a16) sa.load_half(A16, g.lda, m0, g.M, kbeg, kend, tid); else sa.load_any(g.A, g.lda, m0, g.M, kbeg, kend, tid);
    if (b16) sb.load_half(B16, g.ldb, n0, g.N, kbeg, kend, tid); else sb.load_any(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += BKH) {
        if (a16) sa.store_half(As, tid); else sa.store(As, tid);
        if (b16) sb.store_half(Bs, tid); else sb.store(Bs, tid);
        __syncthreads();
        if (AT && do_colsum && tid < BM) {       // bias gradient from the (bf16-rounded) dY tile
#pragma unroll
            for (int d = 0; d < BKH / 2; ++d) {
                const unsigned u = As[tid * LPD + d];
                csum += __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u);
            }
        }
        if (k0 + BKH < kend) {
            if (a16) sa.load_half(A16, g.lda, m0, g.M, k0 + BKH, kend, tid); else sa.load_any(g.A, g.lda, m0, g.M, k0 + BKH, kend, tid);
            if (b16) sb.load_half(B16, g.ldb, n0, g.N, k0 + BKH, kend, tid); else sb.load_any(g.B, g.ldb, n0, g.N, k0 + BKH, kend, tid);
        }
        const int arow = wm * (BM / WM) + li, brow = wn * (BN / WN) + li;
#pragma unroll
        for (int kk = 0; kk < BKH; kk += 16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = frag_h(As, arow + t * 32, kk, lh);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = frag_h(Bs, brow + t * 32, kk, lh);
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y)
                    acc[x][y] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[y], a[x], acc[x][y], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue<BM, BN, WM, WN, SWAP, true>(g, acc, m0, n0, wm, wn, li, lh, tid, do_colsum, csum);
}

template <int BM, int BN, int WM, int WN, bool AT, bool BT>
static int launch_one_h(const GemmP& p, int splits, hipStream_t st) {
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, splits), block(WM * WN * 64);
    const bool vec = gemm_vector_epilogue_ok(p);
    if (p.a_bf16) {
        if (AT != BT) {            // (a_trans, !b_trans) / (!a_trans, b_trans with bf16 B) forms have no bf16-stored instantiation
            CLIFT_REQUIRE(!AT, "clift_gemm(bf16): a bf16-stored transposed A needs the wgrad form (b_trans = 1)");
        }
        if (!vec) k_gemm_bf16<BM, BN, WM, WN, AT, BT, false, true><<<grid, block, 0, st>>>(p);
        else k_gemm_bf16<BM, BN, WM, WN, AT, BT, true, true><<<grid, block, 0, st>>>(p);
    } else {
        if (!vec) k_gemm_bf16<BM, BN, WM, WN, AT, BT, false, false><<<grid, block, 0, st>>>(p);
        else k_gemm_bf16<BM, BN, WM, WN, AT, BT, true, false><<<grid, block, 0, st>>>(p);
    }
    return clift_check_launch("clift_gemm(bf16)");
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm_h(const GemmP& p, int a_trans, int b_trans, int splits, hipStream_t st) {
    if (!a_trans && !b_trans) return launch_one_h<BM, BN, WM, WN, false, false>(p, splits, st);
    if (!a_trans && b_trans) return launch_one_h<BM, BN, WM, WN, false, true>(p, splits, st);
    if (a_trans && b_trans) return launch_one_h<BM, BN, WM, WN, true, true>(p, splits, st);
    return launch_one_h<BM, BN, WM, WN, true, false>(p, splits, st);
}

int clift_gemm_bf16_launch(const GemmP& p, int a_trans, int b_trans, int splits, hipStream_t st) {
    // storage combinations that exist: streamed operands bf16-stored together (A; and B exactly in the wgrad form), weights fp32
    CLIFT_REQUIRE(!p.b_bf16 || (p.a_bf16 && a_trans && b_trans), "clift_gemm(bf16): a bf16-stored B needs the wgrad form with a bf16-stored A");
    CLIFT_REQUIRE(!(p.a_bf16 && a_trans && b_trans) || p.b_bf16, "clift_gemm(bf16): the wgrad form takes both streamed operands bf16-stored or neither");
    if (!a_trans && p.N == 256 && p.K == 256 && p.M >= 64 && p.a_bf16 && p.c_bf16 && splits == 1 && !p.accumulate && !p.c_trans &&
        p.lda % 8 == 0 && p.ldc % 8 == 0 && ((((uintptr_t)p.A) | ((uintptr_t)p.C)) & 15) == 0 && p.ldb % 4 == 0 && (((uintptr_t)p.B) & 15) == 0 &&
        (b_trans ? (p.mask && p.mask_bf16 && !p.bias && p.act == 0 && p.ldmask % 8 == 0 && (((uintptr_t)p.mask) & 15) == 0) : !p.mask))
        return clift_layer_bf16_launch(p, b_trans, st);             // streamed hidden layer: persistent blocks, weights in registers
    if (a_trans && b_trans && p.M == 256 && p.N == 256 && p.K >= 64 && p.a_bf16 && p.b_bf16 && p.accumulate && !p.c_trans && !p.bias && !p.mask &&
        p.lda % 8 == 0 && p.ldb % 8 == 0 && ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) == 0)
        return clift_wgrad_bf16_stream_launch(p, st);              // streamed weight gradient: persistent blocks, transposed LDS reads
    if (!a_trans && b_trans && !p.a_bf16 && !p.b_bf16 && p.c_bf16 && p.mask && p.mask_bf16 && p.N == 256 && p.K <= 32 && p.K <= p.lda && p.lda <= 32 &&
        p.lda % 4 == 0 && p.M >= 4096 && splits == 1 && !p.accumulate && !p.c_trans && !p.bias && p.act == 0 && p.ldc % 4 == 0 && p.ldmask % 4 == 0 &&
        ((((uintptr_t)p.C) | ((uintptr_t)p.mask)) & 7) == 0 && getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_dgrad_narrow_stream_launch(p, 1, st);          // output-layer dgrad with bf16-stored mask / result (fp32 products: exact)
    // the 128-wide appearance MLP with bf16-stored streams (layer_nb16.hip): forward K = 160 / 128 -> 128 (bias, ReLU), masked input gradient
    // 128 -> 128, unmasked input gradient 128 -> 160 (bf16- or fp32-stored result), weight gradients 128 x {128, 160}
    if (!a_trans && p.a_bf16 && splits == 1 && !p.accumulate && !p.c_trans && p.lda % 8 == 0 && p.lda >= p.K && (((uintptr_t)p.A) & 15) == 0 &&
        p.ldb % 4 == 0 && (((uintptr_t)p.B) & 15) == 0 && (((uintptr_t)p.C) & 15) == 0 && p.M >= 64 && getenv("CLIFT_NO_PERSISTENT") == nullptr) {
        const bool fwd = !b_trans && p.N == 128 && (p.K == 128 || p.K == 160) && p.c_bf16 && p.ldc % 8 == 0 && p.ldc >= 128 && !p.mask && p.ldb >= p.K;
        const bool dg_m = b_trans && p.N == 128 && p.K == 128 && p.c_bf16 && p.ldc % 8 == 0 && p.ldc >= 128 && p.mask && p.mask_bf16 && p.ldmask % 8 == 0 &&
                          p.ldmask >= 128 && (((uintptr_t)p.mask) & 15) == 0 && !p.bias && p.act == 0 && p.ldb >= 128;
        const bool dg_u = b_trans && p.N == 160 && p.K == 128 && !p.mask && !p.bias && p.act == 0 && p.ldb >= 160 && p.ldc >= 160 &&
                          (p.c_bf16 ? p.ldc % 8 == 0 : p.ldc % 4 == 0);
        if (fwd || dg_m || dg_u) return clift_layer_nb16_launch(p, b_trans, st);
    }
    if (a_trans && b_trans && p.M == 128 && (p.N == 128 || p.N == 160) && p.K >= 64 && p.a_bf16 && p.b_bf16 && p.accumulate && !p.c_trans && !p.bias && !p.mask &&
        p.lda % 8 == 0 && p.lda >= 128 && p.ldb % 8 == 0 && p.ldb >= p.N && p.ldc >= p.N && ((((uintptr_t)p.A) | ((uintptr_t)p.B)) & 15) == 0 &&
        getenv("CLIFT_NO_PERSISTENT") == nullptr)
        return clift_wgrad_nb16_launch(p, st);
    if (p.N > 128) return launch_gemm_h<128, 256, 2, 4>(p, a_trans, b_trans, splits, st);
    if (p.N > 32) return launch_gemm_h<128, 128, 2, 2>(p, a_trans, b_trans, splits, st);
    return launch_gemm_h<256, 32, 4, 1>(p, a_trans, b_trans, splits, st);
}
