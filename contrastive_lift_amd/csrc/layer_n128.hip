// layer_n128.hip -- the exact-fp32 128-wide layers of the appearance MLP (tensoRF.py:393-397: 150 -> 128 -> ReLU -> 128 -> ReLU) as
// PERSISTENT kernels, same scheme as layer_f32.hip (weights in registers for the block's lifetime, activation rows streamed through
// two LDS stages by LDS-DMA one tile ahead, one barrier per tile, memory instructions spread through the MFMA loop), re-shaped for
// N = 128 output columns: the eight waves of a block form a 2 x 4 grid -- wave (wr, wc) owns rows 32 wr .. +31 of a 64-row tile and
// output columns 32 wc .. +31 -- and K = 4 KC floats per row is a template parameter (KC = 32: the 128 -> 128 layers; KC = 40: the
// first layer, whose 150 inputs are zero-padded to a 160-float pitch so that a row is a whole number of 8-chunk swizzle groups).
// The tiled kernel (gemm.hip, k_gemm<128,128>) ran these layers at ~36 % of the fp32-MFMA rate: a K = 128 launch is 8 k-tiles, so
// its prologue, barriers and un-overlapped epilogue are most of a tile's life.
//   forward: C = relu(A W^T + b), weights [n][k];  dgrad: C = mask . (A W), weights [k][n], fp32 ReLU mask.
// LDS image: lane-linear (DMA); 16-byte chunk c of row r sits in slot c ^ (r & 7) of its row (low three bits only, so that a
// 40-chunk row keeps the permutation inside its 8-chunk groups); a ds_read_b128 pass of 16 lanes (rows r .. r+15, one chunk index)
// then touches every bank exactly twice -- the minimum for 256 bytes.
#include "gemm_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_ROWS = 64;                  // rows per streamed tile (two 32-row halves)

template <int N>
static __device__ __forceinline__ void wait_vmn() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KC, bool DGRAD>
__global__ __launch_bounds__(512, 2) void k_layer_n128(GemmP g, int rows_per_block) {
    constexpr int NJ = KC / 2;               // contraction steps of 8 k
    constexpr int TILE = LN_ROWS * KC;       // float4 per stage
    constexpr int NDMA = KC / 8;             // LDS-DMA instructions per wave per tile (64 chunks each)
    static_assert(KC % 8 == 0 && NJ >= 16, "K must be a multiple of 32 floats, at least 128");
    __shared__ __attribute__((aligned(16))) float4 lds[2 * TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wc = wave & 3, wr = wave >> 2;
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
    float4 prev[4];
    int prev_m = rend;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece(0, i);
    for (int t = 0; t < ntiles; ++t) {
        // DMA of tile t was issued during tile t-1; younger than it: the 4 stores of tile t-2 (the dgrad drained everything for its mask)
        if (t >= 2) wait_vmn<4>(); else wait_vmn<0>();
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
            if (j >= NDMA && j < NDMA + 4) {
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
                if (g.mask) {
                    o.x = mk[q].x > 0.f ? o.x : 0.f; o.y = mk[q].y > 0.f ? o.y : 0.f;
                    o.z = mk[q].z > 0.f ? o.z : 0.f; o.w = mk[q].w > 0.f ? o.w : 0.f;
                }
            } else if (g.act == 1) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            prev[q] = o;
        }
        prev_m = m;
    }
    if (prev_m < rend) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(g.C + (size_t)prev_m * g.ldc + 32 * wc + 8 * q + 4 * lh) = prev[q];
    }
}

// Eligibility is decided by the caller (gemm.hip): N = 128, K in {128, 160} with lda >= K (pad columns of A and of the weight rows
// are zero), plain row-major A, 16-byte-aligned rows; forward: [n][k] weights; dgrad (b_trans): [k][n] weights, fp32 mask (required).
int clift_layer_n128_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, LN_ROWS);
    const int blocks = tiles < 256 ? tiles : 256;
    const int rpb = cdiv(cdiv(p.M, blocks), LN_ROWS) * LN_ROWS;
    const dim3 grid(cdiv(p.M, rpb));
    if (p.K == 160 && !b_trans) k_layer_n128<40, false><<<grid, 512, 0, st>>>(p, rpb);
    else if (p.K == 128 && !b_trans) k_layer_n128<32, false><<<grid, 512, 0, st>>>(p, rpb);
    else if (p.K == 128 && b_trans) k_layer_n128<32, true><<<grid, 512, 0, st>>>(p, rpb);
    else { CLIFT_REQUIRE(false, "clift_gemm(fp32 128-wide layer): unsupported shape K=%d b_trans=%d", p.K, b_trans); }
    return clift_check_launch("clift_gemm(fp32 128-wide layer)");
}
