// layer_nb16.hip -- the 128-wide appearance MLP (tensoRF.py:393-397: 150 -> 128 -> 128 -> 3) in bf16 mode (round 5).
//
// Until round 4 the bf16 mode (BASELINE configs[2]) ran these layers on the exact-fp32 persistent kernels (layer_n128.hip): 6 launches,
// 635 us of a 3.4 ms step, bound by the fp32 matrix pipe.  With bf16 operands the matrix time is a sixteenth of that and the layers are
// HBM streams, so the streamed tensors are bf16-STORED here like the hidden activations of the xyz heads (layer_bf16.hip): the encoded
// input X (M, 160), both hidden activations (M, 128) and their gradients.  Three kernels, all persistent row-range streams with the
// weights in registers and the rows travelling by LDS-DMA a few tiles ahead (the scheme of layer_bf16.hip; waits for a tile are counted
// from a running tally of the wave's vector-memory instructions, so no prefetch is drained at a tile boundary):
//   k_layer_nb16<KS, NCG, DGRAD, MASK, CF32>   C = relu(A W^T + b)  (K = 16 KS in {128, 160} -> N = 32 NCG = 128), or the input gradient
//                                              C = [mask .] (A W)   (K = 128 -> N in {128, 160}; CF32: fp32 result for the encode backward)
//   k_wgrad_nb16<KT>                           gW (128, 32 KT) += dY^T X, gb += column sums of dY, both operands gathered with
//                                              ds_read_b64_tr_b16 (as k_wgrad_bf16_stream)
// Wave = one 32-column group of the output, both 32-row halves of a 64-row tile (4 or 5 waves per block, two blocks per CU).
// LDS image of a tile is lane-linear (a DMA constraint); the bank swizzle is applied at the source address: 256-byte rows (16 chunks of
// 16 B): chunk c of row r in slot c ^ (r & 15); 320-byte rows (20 chunks; consecutive rows already start 16 banks apart): c ^ ((r >> 2) & 3).
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_nb16)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NB_ROWS = 64;

static __device__ __forceinline__ bf16x8 nb_cvt8(const float4 a, const float4 b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}
static __device__ __forceinline__ unsigned nb_pack(float lo, float hi) {
    return (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
}
static __device__ __forceinline__ unsigned nb_keep_positive(unsigned v, unsigned mk) {      // per 16-bit half: v where the mask half is > 0
    const unsigned lo = bf16_bits_positive((unsigned short)(mk & 0xffffu)) ? 0x0000ffffu : 0u;
    const unsigned hi = bf16_bits_positive((unsigned short)(mk >> 16)) ? 0xffff0000u : 0u;
    return v & (lo | hi);
}
template <int N>
static __device__ __forceinline__ void nb_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// allow at most n (wave-uniform) of this wave's vector-memory instructions to be outstanding; rounding down only waits longer
static __device__ __forceinline__ void nb_wait_upto(int n) {
    if (n >= 40) nb_wait<40>();
    else if (n >= 32) nb_wait<32>();
    else if (n >= 28) nb_wait<28>();
    else if (n >= 24) nb_wait<24>();
    else if (n >= 20) nb_wait<20>();
    else if (n >= 18) nb_wait<18>();
    else if (n >= 16) nb_wait<16>();
    else if (n >= 14) nb_wait<14>();
    else if (n >= 12) nb_wait<12>();
    else if (n >= 10) nb_wait<10>();
    else if (n >= 9) nb_wait<9>();
    else if (n >= 8) nb_wait<8>();
    else if (n >= 7) nb_wait<7>();
    else if (n >= 6) nb_wait<6>();
    else if (n >= 5) nb_wait<5>();
    else if (n >= 4) nb_wait<4>();
    else if (n >= 3) nb_wait<3>();
    else if (n >= 2) nb_wait<2>();
    else if (n >= 1) nb_wait<1>();
    else nb_wait<0>();
}
template <int KCH>
static __device__ __forceinline__ int nb_sw(int r) { return KCH == 16 ? (r & 15) : ((r >> 2) & 3); }

// The fused output layer of the forward form (OUTV): the layer is the LAST hidden layer (128 -> 128) and the E <= 4 wide output layer +
// sigmoid (tensoRF.py:397,410) is applied to the finished tile: a lane holds 16 columns of each of its two rows, 16 x E FMAs per row give its
// share, a permlane swap folds the half-waves, the NCG column-waves' shares meet in LDS in a FIXED slot each and 64 x 4 threads add them in
// wave order (deterministic: a row's bits depend on nothing but the row), add the bias, apply the sigmoid and store.
struct NbOut {
    const float* Wout;    // (E, 128), row pitch ldwo; nullptr = no fused output layer
    int ldwo;
    const float* bout;    // (E), nullable
    int E;
    float* out;           // (M, ldo): sigmoid(pre) when `sigmoid`, else pre
    int ldo;
    int sigmoid;
    int store_hidden;     // 0: the hidden activation C is not written (no backward will read it)
};

constexpr int NB_DEPTH = 2, NB_NST = NB_DEPTH + 1;

template <int KS, int NCG, bool DGRAD, bool MASK, bool CF32, bool OUTV>
__global__ __launch_bounds__(64 * NCG, 2) void k_layer_nb16(GemmP g, int rows_per_block, NbOut op) {
    static_assert(!MASK || DGRAD, "the mask belongs to the input gradient");
    static_assert(!OUTV || (!DGRAD && NCG == 4 && !CF32), "the fused output layer goes with the 128-wide forward");
    constexpr int KCH = 2 * KS;                          // 16-byte chunks per A row
    constexpr int TILE = NB_ROWS * KCH;                  // uint4 per stage
    constexpr int PER = (KCH + NCG - 1) / NCG;           // DMA instructions per wave and tile (64 chunks each; the tile has KCH of them)
    __shared__ __attribute__((aligned(16))) uint4 lds[NB_NST * TILE + (OUTV ? NCG * NB_ROWS : 0)];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + NB_ROWS - 1) / NB_ROWS;

    bf16x8 w[KS];                 // w[s] = W(n = 32 wave + li, k = 16 s + 8 lh .. +7)
    float bias[16];
    {
        const int n = 32 * wave + li;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 16 * s + 8 * lh;
            if (!DGRAD) {
                const float4* q = reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + k);
                w[s] = nb_cvt8(q[0], q[1]);
            } else {
                const float* q = g.B + (size_t)k * g.ldb + n;
                const size_t ld = (size_t)g.ldb;
                w[s] = nb_cvt8(make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]), make_float4(q[4 * ld], q[5 * ld], q[6 * ld], q[7 * ld]));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = (!DGRAD && g.bias) ? g.bias[32 * wave + 8 * q + 4 * lh + e] : 0.f;
    }
    // OUTV: this lane's slice of the output weights: wo[c][4 q + e] = Wout[c][32 wave + 8 q + 4 lh + e] (zero for c >= E)
    float wo[OUTV ? 4 : 1][16];
    float bo = 0.f;
    if (OUTV) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) wo[c][4 * q + e] = c < op.E ? op.Wout[(size_t)c * op.ldwo + 32 * wave + 8 * q + 4 * lh + e] : 0.f;
        bo = (op.bout && (tid & 3) < op.E) ? op.bout[tid & 3] : 0.f;
    }
    nb_wait<0>();                                                           // ordinary loads are done before the first DMA is issued
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* K16 = reinterpret_cast<const unsigned short*>(g.mask);
    unsigned short* C16 = reinterpret_cast<unsigned short*>(g.C);
    int issued = 0;               // vector-memory instructions this wave has issued so far (wave-uniform)
    int mark[NB_NST];             // ... at the moment the DMA of the tile in stage i was complete in program order
    auto dma = [&](int t) {
        const int r0 = rbeg + t * NB_ROWS;
        uint4* st = lds + (t % NB_NST) * TILE;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int inst = min(wave * PER + i, KCH - 1);                   // (a wave past the end repeats the last piece: same bytes, same place)
            const int q = inst * 64 + lane, row = q / KCH, slot = q - row * KCH;
            const int c = slot ^ nb_sw<KCH>(row);
            const int gr = min(r0 + row, rend - 1);                          // rows past the range re-read its last row (never stored)
            __builtin_amdgcn_global_load_lds(A16 + (size_t)gr * g.lda + c * 8, (lds_ptr_t)(st + inst * 64), 16, 0, 0);
        }
        issued += PER;
        mark[t % NB_NST] = issued;
    };
    float4* const part = reinterpret_cast<float4*>(lds + NB_NST * TILE);     // OUTV: part[wave * 64 + row of the tile]
    for (int t = 0; t < NB_DEPTH && t < ntiles; ++t) dma(t);
    for (int t = 0; t < ntiles; ++t) {
        nb_wait_upto(issued - mark[t % NB_NST]);                             // this wave's part of tile t has landed ...
        __builtin_amdgcn_s_barrier();                                        // ... and everyone's; everyone is done with stage (t - 1) % NST and the shares
        asm volatile("" ::: "memory");
        const int r0 = rbeg + t * NB_ROWS;
        uint4 mk[2][2];
        if (MASK) {               // the ReLU mask (the layer's bf16-stored input activation): 16 bytes per (row half, column pair), issued before the refill
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp)
                    mk[x][qp] = *reinterpret_cast<const uint4*>(K16 + (size_t)min(r0 + 32 * x + li, rend - 1) * g.ldmask + 32 * wave + 8 * (2 * qp + lh));
            issued += 4;
        }
        if (t + NB_DEPTH < ntiles) dma(t + NB_DEPTH);                        // refill the stage that tile t - 1 just vacated
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        const uint4* T = lds + (t % NB_NST) * TILE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, T[li * KCH + ((2 * s + lh) ^ nb_sw<KCH>(li))]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, T[(li + 32) * KCH + ((2 * s + lh) ^ nb_sw<KCH>(li + 32))]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], a0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], a1, acc[1], 0, 0, 0);
        }
        // epilogue: lane (li, lh) holds row li of each 32-row half, columns 32 wave + 8 q + 4 lh + (0..3) for q = 0..3
        float po[2][4];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int m = r0 + 32 * x + li;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[x][r] + bias[r];
                if (!DGRAD && g.act == 1) v[r] = fmaxf(v[r], 0.f);
            }
            if (OUTV) {
                // the output layer sees the activation the unfused path would have stored: rounded to bf16
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a = fmaf(bf16_bits_to_float(float_to_bf16_bits(v[r])), wo[c][r], a);
                    po[x][c] = a;
                }
            }
            if (CF32) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (m < rend) *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + 32 * wave + 8 * q + 4 * lh) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else if (!OUTV || op.store_hidden) {
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const unsigned p00 = nb_pack(v[8 * qp + 0], v[8 * qp + 1]), p01 = nb_pack(v[8 * qp + 2], v[8 * qp + 3]);
                    const unsigned p10 = nb_pack(v[8 * qp + 4], v[8 * qp + 5]), p11 = nb_pack(v[8 * qp + 6], v[8 * qp + 7]);
                    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(p00, p10, false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(p01, p11, false, false);
                    uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);          // 8 consecutive columns from 32 wave + 8 (2 qp + lh)
                    if (MASK) {
                        const uint4 k4 = mk[x][qp];
                        o.x = nb_keep_positive(o.x, k4.x); o.y = nb_keep_positive(o.y, k4.y);
                        o.z = nb_keep_positive(o.z, k4.z); o.w = nb_keep_positive(o.w, k4.w);
                    }
                    if (m < rend) *reinterpret_cast<uint4*>(C16 + (size_t)m * g.ldc + 32 * wave + 8 * (2 * qp + lh)) = o;
                }
            }
        }
        issued += CF32 ? 8 : ((!OUTV || op.store_hidden) ? 4 : 0);
        if (OUTV) {
            // fold the two half-waves (lower + upper, a fixed order), park this wave's share of the tile's 64 rows
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned u = __float_as_uint(po[x][c]);
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    po[x][c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
            // lanes of the lower half write row li, those of the upper half row 32 + li
            part[wave * NB_ROWS + 32 * lh + li] = lh ? make_float4(po[1][0], po[1][1], po[1][2], po[1][3]) : make_float4(po[0][0], po[0][1], po[0][2], po[0][3]);
            __syncthreads();
            const int row = tid >> 2, c = tid & 3;           // 256 threads = 64 rows x 4 outputs
            const float* pf = reinterpret_cast<const float*>(part);
            const float sv = ((pf[(0 * NB_ROWS + row) * 4 + c] + pf[(1 * NB_ROWS + row) * 4 + c]) + pf[(2 * NB_ROWS + row) * 4 + c]) + pf[(3 * NB_ROWS + row) * 4 + c] + bo;
            const int m = r0 + row;
            if (c < op.E && m < rend) op.out[(size_t)m * op.ldo + c] = op.sigmoid ? 1.f / (1.f + expf(-sv)) : sv;
            issued += 1;
        }
    }
}

// Eligibility is decided by the caller (gemm_bf16.hip).
int clift_layer_nb16_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, NB_ROWS);
    const int want = 2 * clift_persistent_cus();                             // two blocks per CU
    const int blocks = tiles < want ? tiles : want;
    const int rpb = cdiv(cdiv(p.M, blocks), NB_ROWS) * NB_ROWS;
    const dim3 grid(cdiv(p.M, rpb));
    const NbOut none = {nullptr, 0, nullptr, 0, nullptr, 0, 0, 1};
    if (!b_trans && p.K == 160) k_layer_nb16<10, 4, false, false, false, false><<<grid, 256, 0, st>>>(p, rpb, none);
    else if (!b_trans && p.K == 128) k_layer_nb16<8, 4, false, false, false, false><<<grid, 256, 0, st>>>(p, rpb, none);
    else if (b_trans && p.N == 128 && p.mask) k_layer_nb16<8, 4, true, true, false, false><<<grid, 256, 0, st>>>(p, rpb, none);
    else if (b_trans && p.N == 160 && !p.c_bf16) k_layer_nb16<8, 5, true, false, true, false><<<grid, 320, 0, st>>>(p, rpb, none);
    else if (b_trans && p.N == 160 && p.c_bf16) k_layer_nb16<8, 5, true, false, false, false><<<grid, 320, 0, st>>>(p, rpb, none);
    else { clift_set_error("clift_gemm(bf16 128-wide layer): no such form"); return 1; }
    return clift_check_launch("clift_gemm(bf16 128-wide layer)");
}

// Last hidden layer (128 -> 128, bias + ReLU) + E <= 4 output layer (+ sigmoid) of the appearance MLP in bf16 mode, one launch (the bf16
// counterpart of clift_app_head_last2_fwd): A (M, lda) bf16-stored, W (128, 128) fp32 pitch ldw, hidden (M, ldh) bf16-stored or NULL.
extern "C" int clift_app_head_last2_bf16_fwd(const void* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                             const float* bout, int E, int M, void* hidden, int ldh, float* out, int ldo, int sigmoid,
                                             clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4 && ldo >= E && ldwo >= 128 && ldw >= 128 && ldw % 4 == 0, "clift_app_head_last2_bf16_fwd: E in [1,4], pitches >= 128");
    CLIFT_REQUIRE(lda % 8 == 0 && lda >= 128 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0, "clift_app_head_last2_bf16_fwd: A rows 16-byte aligned");
    CLIFT_REQUIRE(!hidden || (ldh % 8 == 0 && ldh >= 128 && (((uintptr_t)hidden) & 15) == 0), "clift_app_head_last2_bf16_fwd: hidden rows 16-byte aligned");
    GemmP p = {};
    p.M = M; p.N = 128; p.K = 128; p.A = reinterpret_cast<const float*>(A); p.lda = lda; p.B = W; p.ldb = ldw;
    p.C = reinterpret_cast<float*>(hidden); p.ldc = ldh; p.bias = b; p.act = 1;
    const NbOut op = {Wout, ldwo, bout, E, out, ldo, sigmoid, hidden ? 1 : 0};
    const int tiles = cdiv(M, NB_ROWS);
    const int want = 2 * clift_persistent_cus();
    const int blocks = tiles < want ? tiles : want;
    const int rpb = cdiv(cdiv(M, blocks), NB_ROWS) * NB_ROWS;
    k_layer_nb16<8, 4, false, false, false, true><<<cdiv(M, rpb), 256, 0, as_stream(s)>>>(p, rpb, op);
    return clift_check_launch("clift_app_head_last2_bf16_fwd");
}

// ============================================================================ weight gradient
// gW[n][k] += sum_m dY[m][n] X[m][k], gb[n] += sum_m dY[m][n]; dY (M, 128) and X (M, 32 KT) bf16-stored.  One persistent 8-wave block per CU
// owns a row range and the whole 128 x 32 KT product: wave (wn, wk) = dY columns 32 wn .. +31 x the k tiles of half wk (KT = 5: three and two).
// Row tiles of 64 through a three-stage LDS ring; both MFMA operands are COLUMNS of the row-major tiles: ds_read_b64_tr_b16 (layer_bf16.hip).
// dY image (256-byte rows): 16-byte chunk c of row r in slot c ^ ((r & 3) << 2), so that the four rows of a transposed read fall into four
// different 64-byte bank spans; the X image likewise when its rows are 256 bytes, unswizzled when they are 320 (consecutive rows then start 64
// bytes apart in the banks by themselves).
static __device__ __forceinline__ uint2 nb_tr_read(unsigned addr) {
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
static __device__ __forceinline__ float nb_pair_sum(unsigned u) { return __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u); }

template <int KT>
__global__ __launch_bounds__(512, 1) void k_wgrad_nb16(GemmP g, int rows_per_block) {
    constexpr int XCH = 4 * KT;                          // chunks per X row
    constexpr int YB = NB_ROWS * 256, XB = NB_ROWS * XCH * 16, STAGE = YB + XB, NST = 3, DEPTH = 2;       // bytes
    constexpr int PERX = (XCH + 7) / 8;                  // X DMA instructions per wave and tile
    constexpr int KH = (KT + 1) / 2;                     // k tiles of the first half
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3, wk = wave >> 2;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.K, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + NB_ROWS - 1) / NB_ROWS;
    const unsigned short* Y16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* X16 = reinterpret_cast<const unsigned short*>(g.B);
    auto dma = [&](int t) {
        const int r0 = rbeg + t * NB_ROWS;
        unsigned char* st = lds + (t % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                        // dY: 16 instructions of 4 rows
            const int inst = wave * 2 + i;
            const int row = inst * 4 + (lane >> 4), c = (lane & 15) ^ ((row & 3) << 2);
            const int gr = min(r0 + row, rend - 1);
            __builtin_amdgcn_global_load_lds(Y16 + (size_t)gr * g.lda + c * 8, (lds_ptr_t)(st + inst * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PERX; ++i) {                                     // X: XCH instructions of 64 chunks
            const int inst = min(wave * PERX + i, XCH - 1);
            const int q = inst * 64 + lane, row = q / XCH, slot = q - row * XCH;
            const int c = XCH == 16 ? (slot ^ ((row & 3) << 2)) : slot;
            const int gr = min(r0 + row, rend - 1);
            __builtin_amdgcn_global_load_lds(X16 + (size_t)gr * g.ldb + c * 8, (lds_ptr_t)(st + YB + inst * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float csum = 0.f;
    // lane-constant fragment addresses (bytes): row 8 lh + ((lane & 15) >> 2) of a 16-row step, 16-column half (lane >> 4) & 1, 4-column quad lane & 3
    const int s4 = (lane >> 2) & 3;                      // row & 3
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const int frow = 8 * lh + ((lane & 15) >> 2);
    const int cin = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);               // chunk inside the tile's 4-chunk group
    const unsigned ya = lds0 + (unsigned)(frow * 256 + (((4 * wn) ^ (4 * s4)) + cin) * 16 + (lane & 1) * 8);
    unsigned xa[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tk = min(wk * KH + j, KT - 1);
        xa[j] = lds0 + (unsigned)(YB + frow * XCH * 16 + ((XCH == 16 ? ((4 * tk) ^ (4 * s4)) : 4 * tk) + cin) * 16 + (lane & 1) * 8);
    }
    const int nk = wk == 0 ? KH : KT - KH;               // k tiles of this wave (wave-uniform)
    constexpr int PERT = 2 + PERX;
    for (int t = 0; t < DEPTH && t < ntiles; ++t) dma(t);
    for (int t = 0; t < ntiles; ++t) {
        if (min(t + DEPTH - 1, ntiles - 1) > t) nb_wait<PERT>(); else nb_wait<0>();           // one younger tile stays in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + DEPTH < ntiles) dma(t + DEPTH);
        const int valid = rend - (rbeg + t * NB_ROWS);
        if (valid < NB_ROWS) {                                               // zero the dY rows past the end (last tile of a range)
            uint4* st = reinterpret_cast<uint4*>(lds + (t % NST) * STAGE);
            for (int e = tid; e < (NB_ROWS - valid) * 16; e += 512) st[valid * 16 + e] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        const unsigned so = (unsigned)((t % NST) * STAGE);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            uint2 yr[2], xr[3][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) yr[p] = nb_tr_read(ya + so + (unsigned)((16 * ms + 4 * p) * 256));
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) xr[j][p] = nb_tr_read(xa[j] + so + (unsigned)((16 * ms + 4 * p) * XCH * 16));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(yr[0]), "+v"(yr[1]), "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[2][0]), "+v"(xr[2][1])
                         :
                         : "memory");
            const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(yr[0].x, yr[0].y, yr[1].x, yr[1].y));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nk) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(xr[j][0].x, xr[j][0].y, xr[j][1].x, xr[j][1].y));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                }
            }
            if (g.colsum && wk == 0) csum += (nb_pair_sum(yr[0].x) + nb_pair_sum(yr[0].y)) + (nb_pair_sum(yr[1].x) + nb_pair_sum(yr[1].y));
        }
    }
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
    if (g.colsum && wk == 0) {       // fold the two m-halves (lanes l, l + 32) first: one atomic per address and wave
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(csum), __float_as_uint(csum), false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0) unsafeAtomicAdd(g.colsum + 32 * wn + li, tot);
    }
}

// Eligibility decided by the caller: gW is 128 x {128, 160}, both streamed operands bf16-stored with 16-byte-aligned rows.
int clift_wgrad_nb16_launch(const GemmP& p, hipStream_t st) {
    const int tiles = cdiv(p.K, NB_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(p.K, blocks), NB_ROWS) * NB_ROWS;
    if (p.N == 128) k_wgrad_nb16<4><<<cdiv(p.K, rpb), 512, 0, st>>>(p, rpb);
    else k_wgrad_nb16<5><<<cdiv(p.K, rpb), 512, 0, st>>>(p, rpb);
    return clift_check_launch("clift_gemm(bf16 128-wide wgrad stream)");
}

// ============================================================================ backward of the E <= 4 output layer over the bf16-stored 128-wide activation
// dX[m][n] = (H[m][n] > 0) . sum_c dOut[m][c] W[c][n]  (bf16-stored),  gW[c][n] += sum_m dOut[m][c] H[m][n],  gb[c] += sum_m dOut[m][c]
// (tensoRF.py:397 backward) in ONE pass over H: a pure stream (256 B in + 256 B out per row), products in fp32 on the VALU (K = E <= 4: an MFMA
// tile would be > 90 % padding).  Thread = 8 consecutive columns (16 bytes of H) of a row, 16 threads per row, 16 rows per block iteration; the
// thread keeps its 4 x 8 weights and 4 x 8 + 4 running sums in registers for the block's whole row range; the 16 threads that own the same
// columns are folded through LDS once per block, one atomic per (c, n) and block (into this XCD's gradient shard when a pass has them on).
__global__ __launch_bounds__(256) void k_out_bwd_nb16(const float* __restrict__ dOut, int ldd, int E, const float* __restrict__ W, int ldw,
                                                       const unsigned short* __restrict__ H, int ldh, int M, unsigned short* __restrict__ dX, int ldx,
                                                       float* __restrict__ gW, int ldgw, float* __restrict__ gb, int rows_per_block) {
    __shared__ float red[16][16 * 36 + 1];
    const int tid = threadIdx.x, cg = tid & 15, rl = tid >> 4;               // column group (8 columns), row of the 16-row step
    const int rbeg = blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
    float w[4][8], aw[4][8], ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { w[c][e] = c < E ? W[(size_t)c * ldw + 8 * cg + e] : 0.f; aw[c][e] = 0.f; }
    for (int m = rbeg + rl; m < rend; m += 16) {
        const float4 d4 = *reinterpret_cast<const float4*>(dOut + (size_t)m * ldd);
        const float d[4] = {d4.x, E > 1 ? d4.y : 0.f, E > 2 ? d4.z : 0.f, E > 3 ? d4.w : 0.f};
        const uint4 h4 = *reinterpret_cast<const uint4*>(H + (size_t)m * ldh + 8 * cg);
        const unsigned hw[4] = {h4.x, h4.y, h4.z, h4.w};
        unsigned ow[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float o[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * p + q;
                const unsigned short hb = (unsigned short)(q ? (hw[p] >> 16) : (hw[p] & 0xffffu));
                const float hv = bf16_bits_to_float(hb);
                const float s = fmaf(d[3], w[3][e], fmaf(d[2], w[2][e], fmaf(d[1], w[1][e], d[0] * w[0][e])));
                o[q] = bf16_bits_positive(hb) ? s : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) aw[c][e] = fmaf(d[c], hv, aw[c][e]);
            }
            ow[p] = nb_pack(o[0], o[1]);
        }
        if (cg == 0) { ab[0] += d[0]; ab[1] += d[1]; ab[2] += d[2]; ab[3] += d[3]; }
        *reinterpret_cast<uint4*>(dX + (size_t)m * ldx + 8 * cg) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    // fold the 16 row-threads of each column group: red[rl][cg * 36 + 8 c + e] (+ 32 + c: the bias sums of column group 0)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl][cg * 36 + 8 * c + e] = aw[c][e];
        red[rl][cg * 36 + 32 + c] = ab[c];
    }
    __syncthreads();
    float* const gw = grad_target(gW);
    float* const gbt = grad_target(gb);
    for (int i = tid; i < 16 * 36; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][i];
        const int g = i / 36, j = i - g * 36;
        if (j < 32) { const int c = j >> 3, e = j & 7; if (c < E) unsafeAtomicAdd(gw + (size_t)c * ldgw + 8 * g + e, s); }
        else if (g == 0 && j - 32 < E && gbt) unsafeAtomicAdd(gbt + (j - 32), s);
    }
}

extern "C" int clift_out_layer_bwd_n128_bf16(const float* dOut, int ldd, int no, const float* W, int ldw, const void* H, int ldh, int M,
                                             void* dX, int ldx, float* gW, int ldgw, float* gb, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(no >= 1 && no <= 4 && ldd >= 4 && ldd % 4 == 0 && (((uintptr_t)dOut) & 15) == 0, "clift_out_layer_bwd_n128_bf16: 1 <= no <= 4, dOut rows of >= 4 floats, 16-byte aligned");
    CLIFT_REQUIRE(ldh % 8 == 0 && ldx % 8 == 0 && ldh >= 128 && ldx >= 128 && ldw >= 128 && ldgw >= 128 && (((uintptr_t)H) & 15) == 0 && (((uintptr_t)dX) & 15) == 0,
                  "clift_out_layer_bwd_n128_bf16: 16-byte aligned bf16 rows with pitches >= 128 required");
    const int want = 4 * clift_persistent_cus();
    const int steps = cdiv(M, 16);
    const int blocks = steps < want ? steps : want;
    const int rpb = cdiv(cdiv(M, blocks), 16) * 16;
    k_out_bwd_nb16<<<cdiv(M, rpb), 256, 0, as_stream(s)>>>(dOut, ldd, no, W, ldw, reinterpret_cast<const unsigned short*>(H), ldh, M,
                                                          reinterpret_cast<unsigned short*>(dX), ldx, gW, ldgw, gb, rpb);
    return clift_check_launch("clift_out_layer_bwd_n128_bf16");
}
