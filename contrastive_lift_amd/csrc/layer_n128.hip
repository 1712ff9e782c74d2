// layer_n128.hip -- the exact-fp32 128-wide layers of the appearance MLP (tensoRF.py:393-397: 150 -> 128 -> ReLU -> 128 -> ReLU) as
// PERSISTENT kernels, same scheme as layer_f32.hip (weights in registers for the block's lifetime, activation rows streamed through
// two LDS stages by LDS-DMA one tile ahead, one barrier per tile, memory instructions spread through the MFMA loop), re-shaped for
// N = 128 output columns: the eight waves of a block form a 2 x 4 grid -- wave (wr, wc) owns rows 32 wr .. +31 of a 64-row tile and
// output columns 32 wc .. +31 -- and K = 4 KC floats per row is a template parameter (KC = 32: the 128 -> 128 layers; KC = 40: the
// first layer, whose 150 inputs are zero-padded to a 160-float pitch so that a row is a whole number of 8-chunk swizzle groups).
// The tiled kernel (gemm.hip, k_gemm<128,128>) ran these layers at ~36 % of the fp32-MFMA rate: a K = 128 launch is 8 k-tiles, so
// its prologue, barriers and un-overlapped epilogue are most of a tile's life.
//   forward: C = relu(A W^T + b), weights [n][k];  dgrad: C = mask . (A W), weights [k][n], fp32 ReLU mask.
// Also here: every persistent WEIGHT GRADIENT (k_wgrad_n128_stream: the 128-wide layers, and the 256 x 256 layers as four quadrants --
// optionally with the X operand generated from the sample positions, GENX: the second layer of an xyz head).
// LDS image: lane-linear (DMA); 16-byte chunk c of row r sits in slot c ^ (r & 7) of its row (low three bits only, so that a
// 40-chunk row keeps the permutation inside its 8-chunk groups); a ds_read_b128 pass of 16 lanes (rows r .. r+15, one chunk index)
// then touches every bank exactly twice -- the minimum for 256 bytes.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_n128)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_ROWS = 64;                  // rows per streamed tile (two 32-row halves)

template <int N>
static __device__ __forceinline__ void wait_vmn() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// OUTV (forward, KC = 32): the layer is the last hidden layer of the appearance MLP and the 3-wide output layer + sigmoid
// (tensoRF.py:395-397,410) is applied to the tile while it is in registers: every lane holds 16 values of its row, 4 x 16 FMAs give its
// share of the E <= 4 dot products (weights from LDS), a permlane swap folds the half-waves, the four column-waves' shares meet in LDS and
// 256 threads add them in a fixed order, add the bias, apply the sigmoid and store -- right after the tile's MFMA loop, between two barriers
// (round 5; until then spread through the MFMA loops of the two following tiles by hand, see below).  Replaces a launch that re-read the
// 512 B-per-row activation plus the row-activation launch.
struct OutN {
    const float* Wout;    // (E, 128), row pitch ldwo
    int ldwo;
    const float* bout;    // (E), nullable
    int E;
    float* pre;           // nullable: (M, ldp) pre-activation outputs
    int ldp;
    float* out;           // (M, ldo) = sigmoid(pre) when `sigmoid`, else pre
    int ldo;
    int sigmoid;
    int store_hidden;
};
constexpr int LN_OUTV_F4 = 128 + 8 * 32;      // float4: Wout rows padded to 4 x 128 floats + 8 waves x 32 rows of shares

template <int KC, bool DGRAD, bool OUTV>
__global__ __launch_bounds__(512, 2) void k_layer_n128(GemmP g, int rows_per_block, OutN op) {
    constexpr int NJ = KC / 2;               // contraction steps of 8 k
    constexpr int TILE = LN_ROWS * KC;       // float4 per stage
    constexpr int NDMA = KC / 8;             // LDS-DMA instructions per wave per tile (64 chunks each)
    static_assert(KC % 8 == 0 && NJ >= 16, "K must be a multiple of 32 floats, at least 128");
    static_assert(!OUTV || (KC == 32 && !DGRAD), "the fused output layer exists for the 128 -> 128 forward layer only");
    __shared__ __attribute__((aligned(16))) float4 lds[2 * TILE + (OUTV ? LN_OUTV_F4 : 0)];
    float4* const wl4 = lds + 2 * TILE;      // OUTV: wl4[c * 32 + k / 4]
    float4* const part = wl4 + 128;          // OUTV: part[wave * 32 + row of the wave's half]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wc = wave & 3, wr = wave >> 2;
    if (rows_limited()) {          // sync-free step: re-balance the row ranges over the true row count (see layer_f32.hip)
        g.M = limit_rows(g.M);
        rows_per_block = ((g.M + (int)gridDim.x - 1) / (int)gridDim.x + LN_ROWS - 1) / LN_ROWS * LN_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + LN_ROWS - 1) / LN_ROWS;

    float4 w[NJ];             // w[j] = B(k = 8 j + 4 lh + 0..3, n = 32 wc + li)
    float bw = 0.f;
    if (!DGRAD) {
        const float* wp = g.B + (size_t)(32 * wc + li) * g.ldb + 4 * lh;
#pragma unroll
        for (int j = 0; j < NJ; ++j) w[j] = *reinterpret_cast<const float4*>(wp + 8 * j);
        bw = (g.bias && lh == 0) ? g.bias[32 * wc + li] : 0.f;
    } else {
        const float* wp = g.B + (size_t)(4 * lh) * g.ldb + 32 * wc + li;
        const size_t ld = (size_t)g.ldb;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float* q = wp + (size_t)(8 * j) * ld;
            w[j] = make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
        }
    }
    wait_vmn<0>();
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const int rt = 32 * wr + li;                                             // this lane's row inside a tile
    unsigned off4[4];         // slot of chunk 2 jj + lh of this lane's row inside its 8-chunk group
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) off4[jj] = (unsigned)(((2 * jj + lh) ^ (rt & 7)) * 16);
    // DMA instruction i of tile t: this wave copies chunks 64 (NDMA wave + i) .. +63 of the tile's row-major chunk list
    auto dma_piece = [&](int t, int i) {
        const int q = 64 * (NDMA * wave + i) + lane;
        const int row = q / KC, c = q - row * KC;
        const int gr = min(rbeg + t * LN_ROWS + row, rend - 1);              // rows past the range re-read its last row (never stored)
        __builtin_amdgcn_global_load_lds(g.A + (size_t)gr * g.lda + (c ^ (row & 7)) * 4,
                                         (lds_ptr_t)(lds + (t & 1) * TILE + 64 * (NDMA * wave + i)), 16, 0, 0);
    };
    float4 prev[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int prev_m = rend;
    // ---- OUTV: the output layer of a finished tile, SYNCHRONOUSLY (round 5).  Rounds 2 - 4 pipelined it through the MFMA loops of the two following
    // tiles with hand-issued LDS traffic (weight fragments read one k-step ahead, the waves' shares parked in a double buffer and fetched a tile
    // later); the frame-render form of that scheme (hidden activation not stored) returned a few rows with ONE wave's share of the wrong tile in
    // ~1 of 15 000 launches, in several blocks at once, single process (tools/last2_soak.py; profiles/r05_determinism.txt) -- a timing-dependent
    // fault that inspection of the listing did not pin down.  The scheme below has no hand-issued LDS traffic and no cross-tile state: shares are
    // formed from the tile's registers right after its MFMA loop (the same FMA chain, so the same bits), meet in ONE buffer between two barriers,
    // and the next tile's barrier protects its reuse.  ~4 % of this (non-default) kernel's time.
    float bo_out = 0.f;
    if (OUTV) {
        float* wl = reinterpret_cast<float*>(wl4);
        for (int e = tid; e < 512; e += 512) { const int c = e >> 7, k = e & 127; wl[e] = c < op.E ? op.Wout[(size_t)c * op.ldwo + k] : 0.f; }
        bo_out = (op.bout && (tid & 3) < op.E) ? op.bout[tid & 3] : 0.f;
        __syncthreads();
    }
    auto outv_tile = [&](int tile) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        float po[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // this lane's columns 32 wc + 8 q + 4 lh + (0..3), ascending: the FMA chain of the pipelined form
                const float4 w4 = wl4[c * 32 + 8 * wc + 2 * q + lh];
                a = fmaf(prev[q].w, w4.w, fmaf(prev[q].z, w4.z, fmaf(prev[q].y, w4.y, fmaf(prev[q].x, w4.x, a))));
            }
            const unsigned u = __float_as_uint(a);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // lower + upper half-wave
            po[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        if (lh == 0) part[wave * 32 + li] = make_float4(po[0], po[1], po[2], po[3]);
        __syncthreads();
        if (tid < 4 * LN_ROWS) {                     // 256 threads = 64 rows x 4 outputs: the four column-waves of the row's half, in wave order
            const int row = tid >> 2, c = tid & 3;
            const float* pf = reinterpret_cast<const float*>(part) + ((4 * (row >> 5)) * 32 + (row & 31)) * 4 + c;
            const float v = ((pf[0] + pf[32 * 4]) + (pf[2 * 32 * 4] + pf[3 * 32 * 4])) + bo_out;
            const int mrow = rbeg + tile * LN_ROWS + row;
            if (c < op.E && mrow < rend) {
                if (op.pre) op.pre[(size_t)mrow * op.ldp + c] = v;
                op.out[(size_t)mrow * op.ldo + c] = op.sigmoid ? 1.f / (1.f + expf(-v)) : v;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece(0, i);
    for (int t = 0; t < ntiles; ++t) {
        // DMA of tile t was issued during tile t-1; younger than it: the 4 stores of tile t-2 (the dgrad drained everything for its mask)
        // (OUTV without a hidden store: the only younger instructions are the owner threads' output stores, which half of the waves never issue)
        if (t >= 2 && (!OUTV || op.store_hidden)) wait_vmn<4>(); else wait_vmn<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int m = rbeg + t * LN_ROWS + rt;
        const bool more = t + 1 < ntiles;
        f32x4 mk[4];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        if (!DGRAD) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bw, 1.0f, acc1, 0, 0, 0);
        const unsigned rowb = lds0 + (unsigned)(((t & 1) * TILE + rt * KC) * 16);
        unsigned ad[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ad[jj] = rowb + off4[jj];
        auto rd = [&](int j, f32x4& x0) {
            const unsigned a = ad[j & 3];
            if ((j >> 2) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 2) == 1) asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 2) == 2) asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 2) == 3) asm volatile("ds_read_b128 %0, %1 offset:384" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 2) == 4) asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(x0) : "v"(a) : "memory");
        };
        f32x4 fa[2];
        rd(0, fa[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4& c0 = fa[j & 1];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0) : : "memory");
            if (j + 1 < NJ) rd(j + 1, fa[(j + 1) & 1]);
            // memory instructions of the tile, one per k-step: next tile's DMA pieces first, then the previous tile's stores, then (dgrad) this
            // tile's mask loads
            if (j < NDMA) { if (more) dma_piece(t + 1, j); }
            if (j >= NDMA && j < NDMA + 4 && (!OUTV || op.store_hidden)) {
                const int q = j - NDMA;
                if (prev_m < rend) *reinterpret_cast<float4*>(g.C + (size_t)prev_m * g.ldc + 32 * wc + 8 * q + 4 * lh) = prev[q];
            }
            if (DGRAD && j >= NDMA + 4 && j < NDMA + 8) {
                const int q = j - NDMA - 4;
                const float* mp = g.mask + (size_t)min(m, rend - 1) * g.ldmask + 32 * wc + 4 * lh + 8 * q;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mk[q]) : "v"(mp) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].x, c0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].y, c0.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].z, c0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].w, c0.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DGRAD) asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]) : : "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o = make_float4(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1],
                                   acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]);
            if (DGRAD) {
                if (g.mask && g.act != 7) {           // act = 7: unmasked dgrad (the mask loads then point at A, results unused)
                    o.x = mk[q].x > 0.f ? o.x : 0.f; o.y = mk[q].y > 0.f ? o.y : 0.f;
                    o.z = mk[q].z > 0.f ? o.z : 0.f; o.w = mk[q].w > 0.f ? o.w : 0.f;
                }
            } else if (g.act == 1) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            prev[q] = o;
        }
        prev_m = m;
        if (OUTV) outv_tile(t);
    }
    if (prev_m < rend && (!OUTV || op.store_hidden)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(g.C + (size_t)prev_m * g.ldc + 32 * wc + 8 * q + 4 * lh) = prev[q];
    }
}

// Eligibility is decided by the caller (gemm.hip): N = 128, K in {128, 160} with lda >= K (pad columns of A and of the weight rows
// are zero), plain row-major A, 16-byte-aligned rows; forward: [n][k] weights; dgrad (b_trans): [k][n] weights, fp32 mask (required).
int clift_layer_n128_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, LN_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(p.M, blocks), LN_ROWS) * LN_ROWS;
    const dim3 grid(cdiv(p.M, rpb));
    const OutN no_out = {nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 1};
    if (p.K == 160 && !b_trans) k_layer_n128<40, false, false><<<grid, 512, 0, st>>>(p, rpb, no_out);
    else if (p.K == 128 && !b_trans) k_layer_n128<32, false, false><<<grid, 512, 0, st>>>(p, rpb, no_out);
    else if (p.K == 128 && b_trans) k_layer_n128<32, true, false><<<grid, 512, 0, st>>>(p, rpb, no_out);
    else { CLIFT_REQUIRE(false, "clift_gemm(fp32 128-wide layer): unsupported shape K=%d b_trans=%d", p.K, b_trans); }
    return clift_check_launch("clift_gemm(fp32 128-wide layer)");
}

// Last hidden layer of the appearance MLP + its output layer + sigmoid in one launch (tensoRF.py:395-397,410):
//   h = relu(A W^T + b) (written to `hidden` if non-null), pre = h Wout^T + bout (written if `pre` non-null), out = sigmoid(pre).
extern "C" int clift_app_head_last2_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                        const float* bout, int E, int M, float* hidden, int ldh, float* pre, int ldp, float* out, int ldo,
                                        int sigmoid, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4, "clift_app_head_last2_fwd: E must be in [1,4] (got %d)", E);
    CLIFT_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= 128 && ldw >= 128,
                  "clift_app_head_last2_fwd: A / W must be 16-byte aligned with pitches >= 128 that are multiples of 4");
    CLIFT_REQUIRE(hidden == nullptr || ((((uintptr_t)hidden) & 15) == 0 && ldh % 4 == 0 && ldh >= 128), "clift_app_head_last2_fwd: hidden must be 16-byte aligned, pitch >= 128");
    GemmP p = {};
    p.M = M; p.N = 128; p.K = 128; p.A = A; p.lda = lda; p.B = W; p.ldb = ldw; p.C = hidden; p.ldc = ldh; p.bias = b; p.act = 1;
    const int tiles = cdiv(M, LN_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), LN_ROWS) * LN_ROWS;
    const OutN op = {Wout, ldwo, bout, E, pre, ldp, out, ldo, sigmoid, hidden != nullptr ? 1 : 0};
    k_layer_n128<32, false, true><<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(p, rpb, op);
    return clift_check_launch("clift_app_head_last2_fwd");
}

// ============================================================================ weight gradient of the same layers, persistent
// gW[n][k] += sum_m dY[m][n] X[m][k]  (+ gb[n] += sum_m dY[m][n]),  fp32: n < 128 (the layer's outputs), k < 32 KXC / 8 ... i.e. X has
// KXC 16-byte chunks per row (32: the 128 -> 128 layer; 40: the first layer's 160-float pitch).  One persistent block per CU owns a
// contiguous range of sample rows and streams 64-row tiles of dY and X through two LDS stages by LDS-DMA (one tile ahead); wave
// (wn, wk) owns gW rows 32 wn .. +31 and the X column tiles {2 wk, 2 wk + 1} (KXC = 32) or {0,1,2} / {3,4} (KXC = 40), i.e. 64-96
// MFMAs per wave between barriers; per step a lane reads one dY and one X element per tile (row-contiguous ds_read_b32, conflict-free
// in a lane-linear image).  The block ends with ONE 128 x K partial added to gW (20 k atomics per block instead of the 64 k of a
// split-K tile launch).  The tiled split-K launch it replaces ran at ~50 TFLOP/s on these shapes.
// GENX (256 x 256 quadrant form only): X is the FIRST layer's activation of an xyz head, X[m][k] = relu(W0[k] . x_m + b0[k]) with K = 3
// (tensoRF.py:475,576) -- the weight gradient of the head's second layer.  Instead of streaming it (1 KB per row, written by the forward for
// this purpose alone) a wave GENERATES the four 1 KB pieces of the X tile it would have DMA'd: a lane owns four columns of its quadrant
// (12 weights + 4 biases in registers) and two rows per piece, whose positions come from a ring of four 1 KB slots filled by LDS-DMA two
// tiles ahead; 13 VALU instructions + one ds_write_b128 per piece, placed where the DMA instructions were.  Same operation order as the
// forward, so the generated values are the forward's bits.
struct GenX {
    const float* x4;      // (rows, 4) normalised sample positions
    const float* W0;      // (256, 3), row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
};

template <int KXC, bool GENX = false>
__global__ __launch_bounds__(512, 2) void k_wgrad_n128_stream(GemmP g, int rows_per_range, int quads, GenX gx) {
    static_assert(!GENX || KXC == 32, "GENX is the 256 x 256 quadrant form");
    constexpr int ROWS = 64;
    constexpr int YB = ROWS * 512, XB = ROWS * KXC * 16, STAGE = YB + XB;     // bytes
    constexpr int NT = (KXC == 40) ? 3 : 2;                                   // accumulator tiles per wave (KXC = 40: 3 + 2 over the two wk)
    constexpr int NDX = KXC / 8;                                              // X DMA instructions per wave per tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE + (GENX ? 4 * 1024 : 0)];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wn = wave & 3, wk = wave >> 2;
    const int tile0 = (KXC == 40) ? 3 * wk : 2 * wk;                          // first X column tile of this wave
    const int ntile = (KXC == 40 && wk == 1) ? 2 : NT;
    // quads == 1: the block owns a row range and the whole 128 x K result.  quads == 4 (256 x 256 results, KXC = 32): block b owns quadrant
    // (b >> 3) & 3 of the result -- dY columns 128 (quad >> 1) .., X columns 128 (quad & 1) .. -- for row range (b & 7) + 8 (b >> 5): the four
    // quadrant-blocks of a range have ids 8 apart, i.e. share an XCD, so dY and X are fetched from HBM once and re-read from that L2
    const int b = blockIdx.x;
    if (rows_limited()) {
        g.K = limit_rows(g.K);
        const int nranges = quads == 4 ? 64 : (int)gridDim.x;
        rows_per_range = ((g.K + nranges - 1) / nranges + ROWS - 1) / ROWS * ROWS;
    }
    const int quad = quads == 4 ? (b >> 3) & 3 : 0, range = quads == 4 ? (b & 7) + 8 * (b >> 5) : b;
    const int rbeg = range * rows_per_range, rend = min(g.K, rbeg + rows_per_range);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + ROWS - 1) / ROWS;
    const float* __restrict__ Y = g.A + 128 * (quad >> 1);            // dY (rows, 128 of lda)
    const float* __restrict__ X = g.B + 128 * (quad & 1);             // X (rows, 4 KXC of ldb)
    float* __restrict__ Cq = grad_target(g.C) + (size_t)(128 * (quad >> 1)) * g.ldc + 128 * (quad & 1);     // (this XCD's shard when a pass has them on)
    const int ncols = quads == 4 ? 128 : g.N;
    // LDS-DMA piece i of tile t (this wave's share): i < 4: two rows of dY (64 rows x 32 chunks per tile); i >= 4: 64 chunks of X
    auto dma_piece = [&](int t, int i) {
        const int r0 = rbeg + t * ROWS;
        unsigned char* st = lds + (t & 1) * STAGE;
        if (i < 4) {
            const int q = 64 * (4 * wave + i) + lane, row = q >> 5, c = q & 31, gr = min(r0 + row, rend - 1);
            __builtin_amdgcn_global_load_lds(Y + (size_t)gr * g.lda + c * 4, (lds_ptr_t)(st + 1024 * (4 * wave + i)), 16, 0, 0);
        } else {
            const int j = i - 4;
            const int q = 64 * (NDX * wave + j) + lane, row = q / KXC, c = q - row * KXC, gr = min(r0 + row, rend - 1);
            __builtin_amdgcn_global_load_lds(X + (size_t)gr * g.ldb + c * 4, (lds_ptr_t)(st + YB + 1024 * (NDX * wave + j)), 16, 0, 0);
        }
    };
    auto dma = [&](int t) {
#pragma unroll
        for (int i = 0; i < (GENX ? 4 : 4 + NDX); ++i) dma_piece(t, i);
    };
    // ---- GENX: this lane's first-layer coefficients (columns 128 (quad & 1) + 4 (lane & 31) .. +3) and the position ring
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float gw0[4] = {0.f, 0.f, 0.f, 0.f}, gw1[4] = {0.f, 0.f, 0.f, 0.f}, gw2[4] = {0.f, 0.f, 0.f, 0.f}, gbb[4] = {0.f, 0.f, 0.f, 0.f};
    if (GENX) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 128 * (quad & 1) + 4 * (lane & 31) + e;
            const float* wr0 = gx.W0 + (size_t)col * gx.ldw0;
            gw0[e] = wr0[0]; gw1[e] = wr0[1]; gw2[e] = wr0[2]; gbb[e] = gx.b0[col];
        }
    }
    const unsigned gpos0 = (unsigned)(uintptr_t)(lds_ptr_t)(lds + 2 * STAGE);
    auto pos_dma = [&](int t) {                       // positions of the 64 rows of tile t -> slot t & 3 (one wave instruction)
        if (wave == 0) {
            const int gr = min(rbeg + t * ROWS + lane, rend - 1);
            __builtin_amdgcn_global_load_lds(gx.x4 + (size_t)gr * 4, (lds_ptr_t)(lds + 2 * STAGE + (t & 3) * 1024), 16, 0, 0);
        }
    };
    f32x4 gpv;                                        // position of the row this lane generates next
    auto gen_read = [&](int t, int j) {               // piece j of tile t covers rows 2 (4 wave + j) + lh
        unsigned a = gpos0 + (unsigned)((t & 3) * 1024 + (2 * (4 * wave + j) + lh) * 16);
        asm volatile("ds_read_b128 %0, %1" : "=v"(gpv) : "v"(a) : "memory");
    };
    auto gen_write = [&](int t, int j) {              // (after an lgkmcnt(0) that covers gen_read)
        asm volatile("" : "+v"(gpv) : : "memory");
        const f32x4 x = gpv;
        f32x4 o;
        o.x = fmaxf(fmaf(gw2[0], x.z, fmaf(gw1[0], x.y, fmaf(gw0[0], x.x, gbb[0]))), 0.f);      // same order as k_linear_k3_fwd
        o.y = fmaxf(fmaf(gw2[1], x.z, fmaf(gw1[1], x.y, fmaf(gw0[1], x.x, gbb[1]))), 0.f);
        o.z = fmaxf(fmaf(gw2[2], x.z, fmaf(gw1[2], x.y, fmaf(gw0[2], x.x, gbb[2]))), 0.f);
        o.w = fmaxf(fmaf(gw2[3], x.z, fmaf(gw1[3], x.y, fmaf(gw0[3], x.x, gbb[3]))), 0.f);
        const unsigned a = gpos0 - (unsigned)(2 * STAGE) + (unsigned)((t & 1) * STAGE + YB + 1024 * (4 * wave + j) + lane * 16);
        asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(o) : "memory");
    };
    f32x16 acc[NT];
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 bsum2 = {0.f, 0.f};
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const unsigned yoff = (unsigned)(lh * 512 + (32 * wn + li) * 4);                           // dY element (row lh, column 32 wn + li)
    const unsigned xoff = (unsigned)(YB + lh * KXC * 16 + (32 * tile0 + li) * 4);              // X element (row lh, column 32 tile0 + li)

    if (GENX) {                                      // positions of tiles 0 and 1, then X of tile 0
        pos_dma(0);
        if (ntiles > 1) pos_dma(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gen_read(0, j);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gen_write(0, j);
        }
    }
    dma(0);
    for (int t = 0; t < ntiles; ++t) {
        if (GENX) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's generated pieces of tile t are written
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // tile t has landed (issued a whole tile ago)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = t + 1 < ntiles;
        const int valid = rend - (rbeg + t * ROWS);
        if (valid < ROWS) {                                                  // last tile of the range: rows past its end contribute nothing
            float* yt = reinterpret_cast<float*>(lds + (t & 1) * STAGE);
            for (int e = valid * 128 + tid; e < ROWS * 128; e += 512) yt[e] = 0.f;
            __syncthreads();
        }
        const unsigned sb = lds0 + (unsigned)((t & 1) * STAGE);
        // ping-pong groups of 4 steps; one ds_read2st64_b32 fetches a lane's element for TWO steps (k-steps are 2 rows = 4 (dY) / KXC / 8 (X)
        // units of 256 bytes apart): 6-9 LDS instructions per group instead of 12-16
        constexpr int XS = KXC / 8;
        f32x2 fa[2][2], fb[2][NT][2];
        auto rd = [&](int grp, int set) {
            const unsigned ya = sb + yoff + (unsigned)(grp * 4096);
            asm volatile("ds_read2st64_b32 %0, %1 offset1:4" : "=v"(fa[set][0]) : "v"(ya) : "memory");
            asm volatile("ds_read2st64_b32 %0, %1 offset0:8 offset1:12" : "=v"(fa[set][1]) : "v"(ya) : "memory");
#pragma unroll
            for (int x = 0; x < NT; ++x)
                if (x < ntile) {
                    const unsigned xa = sb + xoff + (unsigned)(grp * 8 * KXC * 16 + 128 * x);
                    asm volatile("ds_read2st64_b32 %0, %1 offset1:%2" : "=v"(fb[set][x][0]) : "v"(xa), "n"(XS) : "memory");
                    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(fb[set][x][1]) : "v"(xa), "n"(2 * XS), "n"(3 * XS) : "memory");
                }
        };
        rd(0, 0);
#pragma unroll
        for (int grp = 0; grp < ROWS / 8; ++grp) {
            const int set = grp & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grp + 1 < ROWS / 8) rd(grp + 1, set ^ 1);
            // the next tile's DMA pieces are spread over the groups of this tile (eight waves issuing 8-9 of them at the same moment right
            // after the barrier stall the vector-memory issue path with the MFMA pipe idle -- the effect measured in layer_f32.hip)
            if (more && !GENX) {
                if (grp < 4 + NDX) dma_piece(t + 1, grp);
                if (grp == 0 && 4 + NDX > ROWS / 8) dma_piece(t + 1, ROWS / 8);
            }
            if (more && GENX) {
                // dY pieces at groups 0..3; X of tile t+1 generated at groups 4..7 (its positions landed a tile ago: read at group j + 3, behind
                // this group's lgkmcnt(0) at group j + 4); the positions of tile t+2 leave at group 0 (a whole tile to land)
                if (grp < 4) dma_piece(t + 1, grp);
                if (grp >= 4) gen_write(t + 1, grp - 4);
                if (grp >= 3 && grp < 7) gen_read(t + 1, grp - 3);
                if (grp == 0 && t + 2 < ntiles) pos_dma(t + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int x = 0; x < NT; ++x)
                    if (x < ntile) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][u >> 1][u & 1], fb[set][x][u >> 1][u & 1], acc[x], 0, 0, 0);
                if (u & 1) bsum2 += fa[set][u >> 1];          // (one packed add per two steps)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane (li, lh) holds gW rows n = 32 wn + 8 q + 4 lh + e, column 32 (tile0 + x) + li of accumulator x
#pragma unroll
    for (int x = 0; x < NT; ++x) {
        if (x >= ntile) continue;
        const int col = 32 * (tile0 + x) + li;
        if (col >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * wn + 8 * (r >> 2) + 4 * lh + (r & 3);
            unsafeAtomicAdd(Cq + (size_t)n * g.ldc + col, acc[x][r]);
        }
    }
    if (g.colsum && wk == 0 && (quad & 1) == 0) {      // (one of the two quadrants that saw these dY columns adds the bias gradient)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned u = __float_as_uint(bsum2[0] + bsum2[1]);
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0) unsafeAtomicAdd(grad_target(g.colsum) + 128 * (quad >> 1) + 32 * wn + li, tot);
    }
}

// The 256 x 256 weight gradient as four 128 x 128 quadrants of the same kernel (64 row ranges x 4 quadrant-blocks = 256 blocks).
int clift_wgrad_f32_quads_launch(const GemmP& p, hipStream_t st) {
    const int rpr = cdiv(cdiv(p.K, 64), 64) * 64;
    k_wgrad_n128_stream<32><<<256, 512, 0, st>>>(p, rpr, 4, GenX{nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_gemm(fp32 wgrad quadrants)");
}

// Weight / bias gradient of the SECOND layer of an xyz head with the first layer's activation generated in-kernel (GENX above):
//   gW1[n][k] += sum_m dH2[m][n] relu(W0[k] . x4[m] + b0[k]),   gb1[n] += sum_m dH2[m][n]        (tensoRF.py:476-478, 577-579)
extern "C" int clift_xyz_head_first2_wgrad(const float* dH2, int ldd, const float* W0, int ldw0, const float* b0, const float* x4, int M,
                                           float* gW1, int ldgw1, float* gb1, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)dH2) & 15) == 0 && (((uintptr_t)x4) & 15) == 0 && ldd % 4 == 0 && ldd >= 256 && ldgw1 >= 256 && ldw0 >= 3,
                  "clift_xyz_head_first2_wgrad: dH2 / x4 must be 16-byte aligned, dH2's pitch >= 256 and a multiple of 4, gW1's pitch >= 256");
    CLIFT_REQUIRE(W0 != nullptr && b0 != nullptr && gW1 != nullptr, "clift_xyz_head_first2_wgrad: W0, b0 and gW1 are required");
    GemmP p = {};
    p.M = 256; p.N = 256; p.K = M; p.A = dH2; p.lda = ldd; p.B = nullptr; p.ldb = 256; p.C = gW1; p.ldc = ldgw1; p.accumulate = 1; p.colsum = gb1;
    const int rpr = cdiv(cdiv(M, 64), 64) * 64;
    k_wgrad_n128_stream<32, true><<<256, 512, 0, as_stream(s)>>>(p, rpr, 4, GenX{x4, W0, ldw0, b0});
    return clift_check_launch("clift_xyz_head_first2_wgrad");
}

// Eligibility decided by the caller (gemm.hip): wgrad form (a_trans, b_trans, accumulate) with a 128 x N result, N in {128, 160} (N <= ldb:
// the pad columns of X exist and are zero), K (sample rows) >= 4096, 16-byte-aligned rows.
int clift_wgrad_n128_stream_launch(const GemmP& p, hipStream_t st) {
    const int rpr = cdiv(cdiv(p.K, 256), 64) * 64;
    const dim3 grid(cdiv(p.K, rpr));
    if (p.N > 128) k_wgrad_n128_stream<40><<<grid, 512, 0, st>>>(p, rpr, 1, GenX{nullptr, nullptr, 0, nullptr});
    else k_wgrad_n128_stream<32><<<grid, 512, 0, st>>>(p, rpr, 1, GenX{nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_gemm(fp32 128-wide wgrad stream)");
}
