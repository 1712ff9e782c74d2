// layer_n6.hip -- the 128-wide appearance MLP (tensoRF.py:393-397: 150 -> 128 -> 128 -> 3) in the DEFAULT arithmetic, fp32x6 (round 5).
//
// Until now the fp32x6 mode kept these layers on the exact-fp32 persistent kernels (layer_n128.hip): 6 launches, 635 us of a 5.0 ms step, each
// of them at "fp32 MFMA time + stream time" (the two add on this chip under this load: profiles/r04_x6_ablation.txt).  The split arithmetic of
// layer_x6.hip -- every fp32 operand EXACTLY three bf16 terms, the six leading cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate --
// takes 6 / 16 of the matrix time for the same fp32-faithful result, the streams stay as they are (fp32 in memory).
//
// Unlike layer_x6.hip the weights of a 128-wide layer fit ONE workgroup's registers (3 planes x (128 x 160) bf16 = 120 KB over 4 waves), so the
// kernels here are plain: compiler-scheduled, no hand-counted waits, no LDS-DMA.
//   k_layer_n6<KS, NCG, DGRAD, MASK, OUTV>   C = relu(A W^T + b) (K = 16 KS in {128, 160} -> N = 32 NCG = 128; OUTV: + the E <= 4 output layer and
//                                            sigmoid on the finished tile), or the input gradient C = [mask .] (A W) (K = 128 -> N in {128, 160})
//   k_wgrad_n6<KT>                           gW (128, 32 KT) += dY^T X, gb += column sums of dY
// Forward / dgrad: wave = one 32-column group of the output, 32-row tiles.  The rows of a tile are loaded as 16-byte pieces into registers one
// tile ahead (coalesced: lane = piece), split by whoever loaded them (cooperative: each value is split ONCE per workgroup, 22 VALU per piece)
// and written as three bf16 plane images into LDS (two stages; the image and its bank swizzle are layer_nb16.hip's: 16-byte chunk c of row r in
// slot c ^ (r & 15) for 256-byte rows, c ^ ((r >> 2) & 3) for 320-byte rows); every wave then reads whole-K fragments of the tile against its
// register-resident weight planes (reading them a whole k-step ahead of their MFMAs instead of where the compiler puts them measured the same or
// slower: profiles/r05_n6_prefetch_variant.txt).  One barrier per tile.  Two workgroups per CU (60 / 48 KB of LDS each) overlap one's split with the other's MFMAs.
// A row's bits depend on nothing but the row: fixed k order, fixed column partition -- a frame rendered in row tiles is bit-identical to the
// unsharded render.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_n6)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int N6_ROWS = 32;

static __device__ __forceinline__ unsigned n6_pk(float lo, float hi) {         // two fp32 -> packed bf16 (RNE), `lo` in the low half
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
static __device__ __forceinline__ float n6_lo(unsigned p) { return __uint_as_float(p << 16); }
static __device__ __forceinline__ float n6_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// exact three-way split of a pair (x = h + m + l, each term a bf16): the arithmetic of layer_x6.hip's x6_split_pair
static __device__ __forceinline__ void n6_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = n6_pk(a, b);
    const float ra = a - n6_lo(h), rb = b - n6_hi(h);
    m = n6_pk(ra, rb);
    l = n6_pk(ra - n6_lo(m), rb - n6_hi(m));
}
static __device__ __forceinline__ f32x16 n6_mfma(const u32x4& w, const u32x4& a, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}
template <int KCH>
static __device__ __forceinline__ int n6_sw(int r) { return KCH == 16 ? (r & 15) : ((r >> 2) & 3); }

// The fused output layer of the forward form (OUTV): the layer is the LAST hidden layer (128 -> 128) and the E <= 4 wide output layer +
// sigmoid (tensoRF.py:397,410) is applied to the finished tile in exact fp32 FMAs: a lane holds 16 columns of its row, 16 x E FMAs give its
// share, a permlane swap folds the half-waves (lower + upper), the four column-waves' shares meet in LDS in a FIXED slot each and 32 x 4 threads
// add them in wave order, add the bias, apply the sigmoid and store.
struct N6Out {
    const float* Wout;    // (E, 128), row pitch ldwo; nullptr = no fused output layer
    int ldwo;
    const float* bout;    // (E), nullable
    int E;
    float* out;           // (M, ldo): sigmoid(pre) when `sigmoid`, else pre
    int ldo;
    int sigmoid;
    int store_hidden;     // 0: the hidden activation C is not written (no backward will read it)
};

template <int KS, int NCG, bool DGRAD, bool MASK, bool OUTV>
__global__ __launch_bounds__(64 * NCG, 2) void k_layer_n6(GemmP g, int rows_per_block, N6Out op) {
    static_assert(!MASK || DGRAD, "the mask belongs to the input gradient");
    static_assert(!OUTV || (!DGRAD && NCG == 4), "the fused output layer goes with the 128-wide forward");
    constexpr int NT = 64 * NCG;                         // threads
    constexpr int KCH = 2 * KS;                          // 16-byte chunks per row of a plane image
    constexpr int PLANE = N6_ROWS * KCH;                 // uint4 per plane image
    constexpr int STAGE = 3 * PLANE;
    constexpr int C4 = 4 * KS;                           // 16-byte fp32 pieces per row
    constexpr int NPIECE = N6_ROWS * C4;
    constexpr int PER = (NPIECE + NT - 1) / NT;          // pieces per thread and tile
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * STAGE + (OUTV ? NCG * N6_ROWS + 128 : 0)];
    // register ballast (layer_x6w.hip): with two 4-wave workgroups per CU every SIMD holds two of these waves, and with 256 registers each nothing of
    // another PROCESS fits beside this bf16 MFMA stream (the 5-wave form leaves three SIMDs with one wave: no such guarantee there)
    if (NCG == 4) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (rows_limited()) {          // sync-free step: re-balance the row ranges over the true row count (see layer_f32.hip)
        g.M = limit_rows(g.M);
        rows_per_block = ((g.M + (int)gridDim.x - 1) / (int)gridDim.x + N6_ROWS - 1) / N6_ROWS * N6_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + N6_ROWS - 1) / N6_ROWS;

    // ---- weight planes of this wave's 32 columns, every k: w?[s] = planes of W(n = 32 wave + li, k = 16 s + 8 lh .. +7)
    u32x4 wh[KS], wm[KS], wl[KS];
    {
        const int n = 32 * wave + li;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 16 * s + 8 * lh;
            float v[8];
            if (!DGRAD) {
                const float4* q = reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + k);
                const float4 a = q[0], c = q[1];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            } else {
                const float* q = g.B + (size_t)k * g.ldb + n;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = q[(size_t)e * g.ldb];
            }
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                unsigned h, m, l;
                n6_split_pair(v[2 * pr], v[2 * pr + 1], h, m, l);
                wh[s][pr] = h; wm[s][pr] = m; wl[s][pr] = l;
            }
        }
    }
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[4 * q + e] = (!DGRAD && g.bias) ? g.bias[32 * wave + 8 * q + 4 * lh + e] : 0.f;
    // OUTV: the output weights live in LDS (rows c >= E zero), wl4[c * 32 + n / 4]: 64 registers per lane would not fit beside the weight planes
    float4* const part = reinterpret_cast<float4*>(lds + 2 * STAGE);       // OUTV: part[wave * 32 + row of the tile]
    float4* const wl4 = part + NCG * N6_ROWS;
    float bo = 0.f;
    if (OUTV) {
        for (int i = tid; i < 4 * 128; i += NT) {
            const int c = i >> 7, n = i & 127;
            reinterpret_cast<float*>(wl4)[i] = c < op.E ? op.Wout[(size_t)c * op.ldwo + n] : 0.f;
        }
        bo = (op.bout && (tid & 3) < op.E) ? op.bout[tid & 3] : 0.f;
    }

    // ---- the cooperative row pipeline: piece i of this thread = 16-byte piece (tid + NT i) of the tile, row-major
    int prow[PER], pofs[PER];                            // row of the tile; byte offset of the piece's 8 bytes inside a plane image
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid + NT * i, row = p / C4, c4 = p - row * C4;
        prow[i] = row;
        pofs[i] = row * (KCH * 16) + ((((c4 >> 1) ^ n6_sw<KCH>(row)) << 4) | ((c4 & 1) << 3));
    }
    float4 P[PER];
    // (addresses = a wave-uniform base of the block's row range + a 32-bit byte offset: two VALU per piece instead of the 64-bit multiply-add chain;
    // a block's range is far below 4 GB -- M / blocks rows of <= 640 B)
    const char* const Ablk = reinterpret_cast<const char*>(g.A + (size_t)rbeg * g.lda);
    const int nrows_blk = rend - rbeg;
    const unsigned row_bytes = (unsigned)g.lda * 4u;
    unsigned pcol[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) pcol[i] = (unsigned)(((tid + NT * i) - prow[i] * C4) * 16);
    auto fetch = [&](int t, int i) {
        if (NPIECE % NT != 0 && tid + NT * i >= NPIECE) return;
        const unsigned r = (unsigned)min(t * N6_ROWS + prow[i], nrows_blk - 1);  // rows past the range re-read its last row (never stored)
        P[i] = *reinterpret_cast<const float4*>(Ablk + (r * row_bytes + pcol[i]));
    };
    unsigned char* const lb = reinterpret_cast<unsigned char*>(lds);
    auto split_store = [&](int stage, int i) {
        if (NPIECE % NT != 0 && tid + NT * i >= NPIECE) return;
        unsigned h0, m0, l0, h1, m1, l1;
        n6_split_pair(P[i].x, P[i].y, h0, m0, l0);
        n6_split_pair(P[i].z, P[i].w, h1, m1, l1);
        unsigned char* d = lb + stage * (STAGE * 16) + pofs[i];
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + PLANE * 16) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * PLANE * 16) = u32x2{l0, l1};
    };
    // prologue: tile 0 -> stage 0, tile 1 on its way
#pragma unroll
    for (int i = 0; i < PER; ++i) fetch(0, i);
#pragma unroll
    for (int i = 0; i < PER; ++i) split_store(0, i);
#pragma unroll
    for (int i = 0; i < PER; ++i) fetch(1, i);

    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                                 // stage t & 1 is complete; everyone is done reading stage (t + 1) & 1 and the shares
        const int r0 = rbeg + t * N6_ROWS;
        const int m = r0 + li;
        float4 mk[4];
        if (MASK) {               // the ReLU mask (the layer's input activation), the 16 columns this lane finishes: issued before the matrix work
            const float* mp = g.mask + (size_t)min(m, rend - 1) * g.ldmask + 32 * wave + 4 * lh;
#pragma unroll
            for (int q = 0; q < 4; ++q) mk[q] = *reinterpret_cast<const float4*>(mp + 8 * q);
        }
        const uint4* T = lds + (t & 1) * STAGE;
        const int nxt = (t + 1) & 1;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = bias[r]; acc1[r] = 0.f; }     // (the bias rides in the first chain's start: sixteen adds fewer per tile)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ci = li * KCH + ((2 * s + lh) ^ n6_sw<KCH>(li));
            const u32x4 fh = __builtin_bit_cast(u32x4, T[ci]);
            const u32x4 fm = __builtin_bit_cast(u32x4, T[PLANE + ci]);
            const u32x4 fl = __builtin_bit_cast(u32x4, T[2 * PLANE + ci]);
            // the next tile's rows: split one piece per two steps beside the MFMAs, refill its registers with the tile after
            // (unconditional: past the last tile the pieces are clamped copies of the range's last row, split into a stage nobody reads)
            if (s % 2 == 0 && s / 2 < PER) {             // (every other step: ~2.5 VALU per MFMA instead of ~7 in the first half of the tile)
                split_store(nxt, s / 2);
                fetch(t + 2, s / 2);
            }
            acc0 = n6_mfma(wh[s], fh, acc0);
            acc1 = n6_mfma(wh[s], fm, acc1);
            acc0 = n6_mfma(wm[s], fm, acc0);
            acc1 = n6_mfma(wm[s], fh, acc1);
            acc0 = n6_mfma(wh[s], fl, acc0);
            acc1 = n6_mfma(wl[s], fh, acc1);
        }
        // epilogue: lane (li, lh) holds row li of the tile, columns 32 wave + 8 q + 4 lh + (0..3) for q = 0..3
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = acc0[r] + acc1[r];
            if (!DGRAD && g.act == 1) v[r] = fmaxf(v[r], 0.f);
        }
        if (MASK) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[4 * q + 0] = mk[q].x > 0.f ? v[4 * q + 0] : 0.f; v[4 * q + 1] = mk[q].y > 0.f ? v[4 * q + 1] : 0.f;
                v[4 * q + 2] = mk[q].z > 0.f ? v[4 * q + 2] : 0.f; v[4 * q + 3] = mk[q].w > 0.f ? v[4 * q + 3] : 0.f;
            }
        }
        if ((!OUTV || op.store_hidden) && m < rend) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + 32 * wave + 8 * q + 4 * lh) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
        if (OUTV) {
            float po[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {            // this lane's columns 32 wave + 8 q + 4 lh + (0..3), ascending: one FMA chain
                    const float4 w4 = wl4[c * 32 + 8 * wave + 2 * q + lh];
                    a = fmaf(v[4 * q + 0], w4.x, a); a = fmaf(v[4 * q + 1], w4.y, a); a = fmaf(v[4 * q + 2], w4.z, a); a = fmaf(v[4 * q + 3], w4.w, a);
                }
                const unsigned u = __float_as_uint(a);
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // lower + upper half-wave, a fixed order
                po[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            if (lh == 0) part[wave * N6_ROWS + li] = make_float4(po[0], po[1], po[2], po[3]);
            __syncthreads();
            if (tid < 4 * N6_ROWS) {
                const int row = tid >> 2, c = tid & 3;       // 128 threads = 32 rows x 4 outputs
                const float* pf = reinterpret_cast<const float*>(part);
                const float sv = ((pf[(0 * N6_ROWS + row) * 4 + c] + pf[(1 * N6_ROWS + row) * 4 + c]) + pf[(2 * N6_ROWS + row) * 4 + c]) + pf[(3 * N6_ROWS + row) * 4 + c] + bo;
                const int mo = r0 + row;
                if (c < op.E && mo < rend) op.out[(size_t)mo * op.ldo + c] = op.sigmoid ? 1.f / (1.f + expf(-sv)) : sv;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the fetches past the last tile were issued unconditionally: a wave does not end with loads in flight
}

static int n6_grid(int M, int per_cu, int& rpb) {
    const int tiles = cdiv(M, N6_ROWS);
    const int want = per_cu * clift_persistent_cus();
    const int blocks = tiles < want ? tiles : want;
    rpb = cdiv(cdiv(M, blocks), N6_ROWS) * N6_ROWS;
    return cdiv(M, rpb);
}

// Eligibility is decided by the caller (gemm.hip): fp32 row-major A with 16-byte-aligned rows; forward: [n][k] weights, bias, ReLU, no mask;
// dgrad (b_trans): [k][n] weights, fp32 mask (N = 128) or none (N = 160), no bias / activation.
int clift_layer_n6_launch(const GemmP& p, int b_trans, hipStream_t st) {
    int rpb;
    const N6Out none = {nullptr, 0, nullptr, 0, nullptr, 0, 0, 1};
    if (!b_trans && p.K == 160) { const int grid = n6_grid(p.M, 2, rpb); k_layer_n6<10, 4, false, false, false><<<grid, 256, 0, st>>>(p, rpb, none); }
    else if (!b_trans && p.K == 128) { const int grid = n6_grid(p.M, 2, rpb); k_layer_n6<8, 4, false, false, false><<<grid, 256, 0, st>>>(p, rpb, none); }
    else if (b_trans && p.N == 128 && p.mask) { const int grid = n6_grid(p.M, 2, rpb); k_layer_n6<8, 4, true, true, false><<<grid, 256, 0, st>>>(p, rpb, none); }
    else if (b_trans && p.N == 160 && !p.mask) { const int grid = n6_grid(p.M, 1, rpb); k_layer_n6<8, 5, true, false, false><<<grid, 320, 0, st>>>(p, rpb, none); }
    else { clift_set_error("clift_gemm(fp32x6 128-wide layer): no such form"); return 1; }
    return clift_check_launch("clift_gemm(fp32x6 128-wide layer)");
}

// Last hidden layer (128 -> 128, bias + ReLU) + E <= 4 output layer (+ sigmoid) of the appearance MLP, fp32x6 arithmetic for the 128 x 128 layer,
// exact fp32 FMAs for the output layer, one launch (the fp32x6 counterpart of clift_app_head_last2_fwd): hidden (M, ldh) fp32 or NULL.
extern "C" int clift_app_head_last2_x6_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                           const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, int sigmoid,
                                           clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4 && ldo >= E && ldwo >= 128, "clift_app_head_last2_x6_fwd: E in [1,4], ldo >= E, ldwo >= 128");
    CLIFT_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= 128 && ldw >= 128,
                  "clift_app_head_last2_x6_fwd: A / W must be 16-byte aligned with pitches >= 128 that are multiples of 4");
    CLIFT_REQUIRE(hidden == nullptr || ((((uintptr_t)hidden) & 15) == 0 && ldh % 4 == 0 && ldh >= 128), "clift_app_head_last2_x6_fwd: hidden must be 16-byte aligned, pitch >= 128");
    GemmP p = {};
    p.M = M; p.N = 128; p.K = 128; p.A = A; p.lda = lda; p.B = W; p.ldb = ldw; p.C = hidden; p.ldc = ldh; p.bias = b; p.act = 1;
    const N6Out op = {Wout, ldwo, bout, E, out, ldo, sigmoid, hidden ? 1 : 0};
    int rpb;
    const int grid = n6_grid(M, 2, rpb);
    k_layer_n6<8, 4, false, false, true><<<grid, 256, 0, as_stream(s)>>>(p, rpb, op);
    return clift_check_launch("clift_app_head_last2_x6_fwd");
}

// ============================================================================ weight gradient
// gW[n][k] += sum_m dY[m][n] X[m][k], gb[n] += sum_m dY[m][n]; dY (M, 128) and X (M, 32 KT) fp32, six bf16 products of the split operands.
// One persistent 8-wave block per CU owns a row range and the whole 128 x 32 KT product: wave (wn, wk) = dY columns 32 wn .. +31 x the k tiles of
// half wk (KT = 5: three and two) -- the shape of k_wgrad_nb16 (layer_nb16.hip), whose LDS images and transposing fragment reads
// (ds_read_b64_tr_b16: both MFMA operands are COLUMNS of the row-major tiles) are used here once per bf16 plane.  32-row tiles, two stages; the
// rows arrive through registers one tile ahead and are split by whoever loaded them, as in k_layer_n6; rows past the end of the range are
// zeroed in the split (a clamped copy would be counted twice).
template <int OFF>
static __device__ __forceinline__ uint2 n6_tr_read(unsigned addr) {          // (the constant part of the address rides in the instruction's offset field)
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
typedef __attribute__((address_space(3))) void* n6_lds_ptr_t;

template <int KT>
__global__ __launch_bounds__(512, 1) void k_wgrad_n6(GemmP g, int rows_per_block) {
    constexpr int XCH = 4 * KT;                          // 16-byte chunks per row of an X plane image
    constexpr int YPL = N6_ROWS * 256, XPL = N6_ROWS * XCH * 16;             // bytes of one plane image
    constexpr int YB = 3 * YPL, STAGE = YB + 3 * XPL;
    constexpr int KH = (KT + 1) / 2;                     // k tiles of the first half
    constexpr int YPER = N6_ROWS * 32 / 512;             // dY pieces per thread and tile (2)
    constexpr int XC4 = 8 * KT, XPIECE = N6_ROWS * XC4, XPER = (XPIECE + 511) / 512;      // X pieces (KT = 4: 2; KT = 5: 3, the last one half-filled)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");       // register ballast: eight waves x 256 registers fill the CU's files (see k_layer_n6)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3, wk = wave >> 2;
    if (rows_limited()) {
        g.K = limit_rows(g.K);
        rows_per_block = ((g.K + (int)gridDim.x - 1) / (int)gridDim.x + N6_ROWS - 1) / N6_ROWS * N6_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.K, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + N6_ROWS - 1) / N6_ROWS;

    // ---- the row pipeline: dY piece i = 16-byte piece (tid + 512 i) of the tile's dY rows, X piece likewise
    int yrow[YPER], yofs[YPER], xrow[XPER], xofs[XPER];
#pragma unroll
    for (int i = 0; i < YPER; ++i) {
        const int p = tid + 512 * i, row = p >> 5, c4 = p & 31;
        yrow[i] = row;
        yofs[i] = row * 256 + ((((c4 >> 1) ^ ((row & 3) << 2)) << 4) | ((c4 & 1) << 3));
    }
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        const int p = tid + 512 * i, row = p / XC4, c4 = p - row * XC4;
        xrow[i] = row;
        xofs[i] = YB + row * (XCH * 16) + ((((c4 >> 1) ^ (XCH == 16 ? ((row & 3) << 2) : 0)) << 4) | ((c4 & 1) << 3));
    }
    float4 PY[YPER], PX[XPER];
    // bias gradient: this thread's dY pieces are always the same four columns 4 (tid & 31) .. +3 (512 is a multiple of the 32 pieces of a row), rows
    // (tid >> 5) + 16 i of every tile: four running sums of the exact fp32 values, folded over the 16 row-threads once per block
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    // (block-relative 32-bit byte offsets from wave-uniform bases, as in k_layer_n6)
    const char* const Yblk = reinterpret_cast<const char*>(g.A + (size_t)rbeg * g.lda);
    const char* const Xblk = reinterpret_cast<const char*>(g.B + (size_t)rbeg * g.ldb);
    const int nrows_blk = rend - rbeg;
    const unsigned ybytes = (unsigned)g.lda * 4u, xbytes = (unsigned)g.ldb * 4u;
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < YPER; ++i) {
            const unsigned r = (unsigned)min(t * N6_ROWS + yrow[i], nrows_blk - 1);
            PY[i] = *reinterpret_cast<const float4*>(Yblk + (r * ybytes + (unsigned)(((tid + 512 * i) & 31) * 16)));
        }
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            if (XPIECE % 512 != 0 && tid + 512 * i >= XPIECE) continue;
            const unsigned r = (unsigned)min(t * N6_ROWS + xrow[i], nrows_blk - 1);
            PX[i] = *reinterpret_cast<const float4*>(Xblk + (r * xbytes + (unsigned)(((tid + 512 * i) - xrow[i] * XC4) * 16)));
        }
    };
    auto put = [&](int stage, int ofs, int pl_bytes, float4 v, bool live) {
        if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned h0, m0, l0, h1, m1, l1;
        n6_split_pair(v.x, v.y, h0, m0, l0);
        n6_split_pair(v.z, v.w, h1, m1, l1);
        unsigned char* d = lds + stage * STAGE + ofs;
        *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + pl_bytes) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(d + 2 * pl_bytes) = u32x2{l0, l1};
    };
    auto split_y = [&](int stage, int t, int i) {
        const bool live = rbeg + t * N6_ROWS + yrow[i] < rend;
        if (live) { csum.x += PY[i].x; csum.y += PY[i].y; csum.z += PY[i].z; csum.w += PY[i].w; }
        put(stage, yofs[i], YPL, PY[i], live);
    };
    auto split_x = [&](int stage, int t, int i) {
        if (XPIECE % 512 != 0 && tid + 512 * i >= XPIECE) return;
        put(stage, xofs[i], XPL, PX[i], rbeg + t * N6_ROWS + xrow[i] < rend);
    };

    f32x16 acc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // lane-constant fragment addresses (bytes), as in k_wgrad_nb16: row 8 lh + ((lane & 15) >> 2) of a 16-row step, 4-chunk group by column tile
    const int s4 = (lane >> 2) & 3;                      // row & 3
    const unsigned lds0 = (unsigned)(uintptr_t)(n6_lds_ptr_t)lds;
    const int frow = 8 * lh + ((lane & 15) >> 2);
    const int cin = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const unsigned ya = lds0 + (unsigned)(frow * 256 + (((4 * wn) ^ (4 * s4)) + cin) * 16 + (lane & 1) * 8);
    unsigned xa[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tk = min(wk * KH + j, KT - 1);
        xa[j] = lds0 + (unsigned)(YB + frow * XCH * 16 + ((XCH == 16 ? ((4 * tk) ^ (4 * s4)) : 4 * tk) + cin) * 16 + (lane & 1) * 8);
    }
    const int nk = wk == 0 ? KH : KT - KH;               // k tiles of this wave (wave-uniform)

    fetch(0);
#pragma unroll
    for (int i = 0; i < YPER; ++i) split_y(0, 0, i);
#pragma unroll
    for (int i = 0; i < XPER; ++i) split_x(0, 0, i);
    fetch(1);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                                 // stage t & 1 is complete; everyone is done reading stage (t + 1) & 1
        const unsigned so = (unsigned)((t & 1) * STAGE);
        const int nxt = (t + 1) & 1;
        const unsigned yb = ya + so, xb0 = xa[0] + so, xb1 = xa[1] + so, xb2 = xa[2] + so;
        auto tr_y = [&](int pl, int ms, int p) -> uint2 {
#define N6_Y(PL, MS, PP) if (pl == PL && ms == MS && p == PP) return n6_tr_read<PL * YPL + (16 * MS + 4 * PP) * 256>(yb);
            N6_Y(0, 0, 0) N6_Y(0, 0, 1) N6_Y(0, 1, 0) N6_Y(0, 1, 1) N6_Y(1, 0, 0) N6_Y(1, 0, 1) N6_Y(1, 1, 0) N6_Y(1, 1, 1) N6_Y(2, 0, 0) N6_Y(2, 0, 1) N6_Y(2, 1, 0) N6_Y(2, 1, 1)
#undef N6_Y
            return uint2{0u, 0u};
        };
        auto tr_x = [&](int pl, int j, int ms, int p) -> uint2 {
            const unsigned xb = j == 0 ? xb0 : j == 1 ? xb1 : xb2;
#define N6_X(PL, MS, PP) if (pl == PL && ms == MS && p == PP) return n6_tr_read<PL * XPL + (16 * MS + 4 * PP) * XCH * 16>(xb);
            N6_X(0, 0, 0) N6_X(0, 0, 1) N6_X(0, 1, 0) N6_X(0, 1, 1) N6_X(1, 0, 0) N6_X(1, 0, 1) N6_X(1, 1, 0) N6_X(1, 1, 1) N6_X(2, 0, 0) N6_X(2, 0, 1) N6_X(2, 1, 0) N6_X(2, 1, 1)
#undef N6_X
            return uint2{0u, 0u};
        };
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            uint2 yr[3][2], xr[3][3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int p = 0; p < 2; ++p) yr[pl][p] = tr_y(pl, ms, p);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int p = 0; p < 2; ++p) xr[pl][j][p] = tr_x(pl, j, ms, p);
            // the next tile's rows: split beside the matrix work (past the last tile: zeros into a stage nobody reads)
            if (ms == 0) {
#pragma unroll
                for (int i = 0; i < YPER; ++i) split_y(nxt, t + 1, i);
            } else {
#pragma unroll
                for (int i = 0; i < XPER; ++i) split_x(nxt, t + 1, i);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(yr[0][0]), "+v"(yr[0][1]), "+v"(yr[1][0]), "+v"(yr[1][1]), "+v"(yr[2][0]), "+v"(yr[2][1]),
                           "+v"(xr[0][0][0]), "+v"(xr[0][0][1]), "+v"(xr[0][1][0]), "+v"(xr[0][1][1]), "+v"(xr[0][2][0]), "+v"(xr[0][2][1]),
                           "+v"(xr[1][0][0]), "+v"(xr[1][0][1]), "+v"(xr[1][1][0]), "+v"(xr[1][1][1]), "+v"(xr[1][2][0]), "+v"(xr[1][2][1]),
                           "+v"(xr[2][0][0]), "+v"(xr[2][0][1]), "+v"(xr[2][1][0]), "+v"(xr[2][1][1]), "+v"(xr[2][2][0]), "+v"(xr[2][2][1])
                         :
                         : "memory");
            u32x4 a[3], b[3][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = u32x4{yr[pl][0].x, yr[pl][0].y, yr[pl][1].x, yr[pl][1].y};
#pragma unroll
                for (int j = 0; j < 3; ++j) b[pl][j] = u32x4{xr[pl][j][0].x, xr[pl][j][0].y, xr[pl][j][1].x, xr[pl][j][1].y};
            }
            // six products per k tile, the k tiles' accumulators in rotation (no two consecutive MFMAs on the same one); the number of k tiles is
            // wave-uniform: one straight-line body per count
            auto products = [&](const int nj) {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    const int pa = pr == 0 ? 0 : pr == 1 ? 0 : pr == 2 ? 1 : pr == 3 ? 1 : pr == 4 ? 0 : 2;      // dY plane:  h h m m h l
                    const int pb = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 0 : pr == 3 ? 1 : pr == 4 ? 2 : 0;      // X plane:   h m h m l h
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < nj) acc[j] = n6_mfma(a[pa], b[pb][j], acc[j]);
                }
            };
            if (KT - KH == KH || nk == KH) { if (KH == 3) products(3); else products(2); }
            else products(KT - KH);
        }
        fetch(t + 2);                                    // (clamped rows past the end; never used)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a wave does not end with loads in flight
    // lane (li, lh) holds gW rows n = 32 wn + 8 q + 4 lh + e, column k = 32 tk + li
    g.C = grad_target(g.C); g.colsum = grad_target(g.colsum);                // (this XCD's shard when a pass has them on)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < nk) {
            const int tk = wk * KH + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * wn + 8 * (r >> 2) + 4 * lh + (r & 3);
                unsafeAtomicAdd(g.C + (size_t)n * g.ldc + 32 * tk + li, acc[j][r]);
            }
        }
    }
    if (g.colsum) {                  // fold the 16 row-threads of each column quad through LDS (every wave is past its last fragment read after this barrier)
        __syncthreads();
        float4* const red = reinterpret_cast<float4*>(lds);
        red[tid] = csum;             // red[(tid >> 5) * 32 + quad]
        __syncthreads();
        if (tid < 128) {
            const int quad = tid >> 2, e = tid & 3;
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ssum += reinterpret_cast<const float*>(red + r * 32 + quad)[e];
            unsafeAtomicAdd(g.colsum + tid, ssum);
        }
    }
}

// Eligibility decided by the caller (gemm.hip): gW is 128 x {128, 160}, both streamed operands fp32 with 16-byte-aligned rows.
int clift_wgrad_n6_launch(const GemmP& p, hipStream_t st) {
    const int tiles = cdiv(p.K, N6_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(p.K, blocks), N6_ROWS) * N6_ROWS;
    if (p.N == 128) k_wgrad_n6<4><<<cdiv(p.K, rpb), 512, 0, st>>>(p, rpb);
    else k_wgrad_n6<5><<<cdiv(p.K, rpb), 512, 0, st>>>(p, rpb);
    return clift_check_launch("clift_gemm(fp32x6 128-wide wgrad)");
}
