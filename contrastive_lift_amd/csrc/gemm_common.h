// gemm_common.h -- launch parameters and the shared epilogue of the MFMA GEMM kernels (gemm.hip: fp32 operands,
// gemm_bf16.hip: bf16 operands).  The C/D register layout of the 32x32 MFMA family is dtype-independent on gfx950, so
// bias / activation / mask / accumulate / transposed-store / bias-gradient handling is written once.
#pragma once
#include "clift_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmP {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    int act;
    const float* mask; int ldmask;
    int accumulate;
    int k_per_split;
    int c_trans;      // write C[n*ldc + m] instead of C[m*ldc + n]
    float* colsum;    // nullable: colsum[m] += sum_k A(m,k)   (bias gradient fused into the wgrad, a_trans only)
    // bf16 mode only (gemm_bf16.hip): which of the tensors are STORED as bf16 (2 bytes per element, pitches in elements)
    int a_bf16, b_bf16, c_bf16, mask_bf16;
    // fp32x6 persistent layer kernels only (layer_x6.hip): one byte per lane and 32-row tile holding the signs of the 8 columns the lane finishes
    // -- written by a forward, read by the masked dgrad of the following layer INSTEAD of the fp32 mask (32 B per row instead of 1 KB)
    unsigned char* sign_bits;
};

static __device__ __forceinline__ float bf16_bits_to_float(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
static __device__ __forceinline__ unsigned short float_to_bf16_bits(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}
static __device__ __forceinline__ bool bf16_bits_positive(unsigned short u) { return (u & 0x7fffu) != 0 && !(u & 0x8000u); }

// swapped-operand (16-byte store) epilogue only where every row can take aligned float4 stores; accumulating, transposed
// and odd-pitch (narrow head outputs) launches keep lanes along the output row
static inline bool gemm_vector_epilogue_ok(const GemmP& p) {
    return !p.accumulate && !p.c_trans && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) && p.N >= 32 &&
           (!p.mask || ((p.ldmask % 4 == 0) && (((uintptr_t)p.mask & 15) == 0)));
}

int clift_gemm_bf16_launch(const GemmP& p, int a_trans, int b_trans, int splits, hipStream_t st);   // gemm_bf16.hip
int clift_layer_bf16_launch(const GemmP& p, int b_trans, hipStream_t st);                            // layer_bf16.hip
int clift_wgrad_bf16_stream_launch(const GemmP& p, hipStream_t st);
int clift_layer_nb16_launch(const GemmP& p, int b_trans, hipStream_t st);                            // layer_nb16.hip
int clift_wgrad_nb16_launch(const GemmP& p, hipStream_t st);
int clift_layer_f32_launch(const GemmP& p, int b_trans, hipStream_t st);                             // layer_f32.hip
int clift_layer_n128_launch(const GemmP& p, int b_trans, hipStream_t st);                            // layer_n128.hip
int clift_wgrad_n128_stream_launch(const GemmP& p, hipStream_t st);
int clift_wgrad_f32_quads_launch(const GemmP& p, hipStream_t st);
int clift_wgrad_narrow_stream_launch(const float* dY, int ldd, int no, const float* X, int ldx, int ni, int M, float* gW, int ldw, float* gb, int x_bf16,
                                     hipStream_t st);                                             // narrow_stream.hip
int clift_dgrad_narrow_stream_launch(const GemmP& p, int half, hipStream_t st);
int clift_k3_bwd_stream_launch(const float* x4, const float* dH, int ldh, int M, float* dW, int ldw, float* db, int dh_bf16, hipStream_t st);
long clift_gemm_split_workspace_bytes(int N, int K);                                                // gemm_split.hip
int clift_gemm_split_launch(const GemmP& p, int a_trans, int b_trans, void* workspace, hipStream_t st);
int clift_layer_x6_launch(const GemmP& p, int b_trans, hipStream_t st);                              // layer_x6.hip
int clift_wgrad_x6_launch(const GemmP& p, hipStream_t st);                                           // layer_x6w.hip
int clift_layer_n6_launch(const GemmP& p, int b_trans, hipStream_t st);                              // layer_n6.hip
int clift_wgrad_n6_launch(const GemmP& p, hipStream_t st);

// acc[x][y] = 32x32 accumulator tile (x, y) of this wave; (wm, wn) = wave coordinates in the block; li = lane & 31,
// lh = lane >> 5.  csum = this thread's partial bias-gradient (column sum of A) when do_colsum.
// HALF = true compiles in the bf16-stored output / mask forms (selected at run time by g.c_bf16 / g.mask_bf16).
template <int BM, int BN, int WM, int WN, bool SWAP, bool HALF = false>
__device__ __forceinline__ void gemm_epilogue(const GemmP& g, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int m0, int n0, int wm, int wn,
                                              int li, int lh, int tid, bool do_colsum, float csum) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const bool c16 = HALF && g.c_bf16, m16 = HALF && g.mask_bf16;
    auto mask_on = [&](size_t idx) {
        return m16 ? bf16_bits_positive(reinterpret_cast<const unsigned short*>(g.mask)[idx]) : (g.mask[idx] > 0.f);
    };
    auto put = [&](size_t idx, float v) {
        if (c16) reinterpret_cast<unsigned short*>(g.C)[idx] = float_to_bf16_bits(v);
        else if (g.accumulate) unsafeAtomicAdd(g.C + idx, v);
        else g.C[idx] = v;
    };
    // Epilogue.  The MFMAs were issued with swapped operands (weight fragment first), so the accumulator tile is C^T in
    // the standard C/D layout: lane l owns output ROW m = tile_m + (l & 31) and, per group q = reg >> 2, the four CONSECUTIVE
    // columns n = tile_n + 8q + 4(l >> 5) + (reg & 3).  That lets every lane move 16 bytes per store / mask load: a
    // 4-byte-per-lane epilogue is store-ISSUE-bound on this chip (~7 B/clk/CU, MI355X_MICROARCH.md), measured here as
    // ~16 us of a 97 us block period.
    // Accumulating launches (split-K wgrad) keep the un-swapped order instead (SWAP = false): there the 32 lanes of a
    // half-wave own 32 consecutive columns of one row, which is what keeps the fp32 atomics line-coalesced.
    if (!SWAP) {
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const int n = n0 + wn * (BN / WN) + y * 32 + li;
                if (n >= g.N) continue;
                const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / WM) + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= g.M) continue;
                    float v = acc[x][y][r] + bv;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    if (g.mask && !mask_on((size_t)m * g.ldmask + n)) v = 0.f;
                    put(g.c_trans ? (size_t)n * g.ldc + m : (size_t)m * g.ldc + n, v);
                }
            }
        if (do_colsum && tid < BM && m0 + tid < g.M) unsafeAtomicAdd(g.colsum + m0 + tid, csum);
        return;
    }
    const bool vec_ok = !g.accumulate && !g.c_trans && (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0) &&
                        (!g.mask || ((g.ldmask % 4 == 0) && (((uintptr_t)g.mask & 15) == 0)));     // (16-byte bases also cover the 8-byte bf16 forms)
#pragma unroll
    for (int x = 0; x < TM; ++x) {
        const int m = m0 + wm * (BM / WM) + x * 32 + li;
        if (m >= g.M) continue;
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * (BN / WN) + y * 32 + 8 * q + 4 * lh;
                if (n >= g.N) continue;
                float v[4] = {acc[x][y][4 * q + 0], acc[x][y][4 * q + 1], acc[x][y][4 * q + 2], acc[x][y][4 * q + 3]};
                if (vec_ok && n + 3 < g.N) {
                    if (g.bias) { const float4 bb = *reinterpret_cast<const float4*>(g.bias + n); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
                    if (g.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (g.mask) {
                        if (m16) {
                            const uint2 mk = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.mask) + (size_t)m * g.ldmask + n);
                            if (!bf16_bits_positive((unsigned short)(mk.x & 0xffffu))) v[0] = 0.f; if (!bf16_bits_positive((unsigned short)(mk.x >> 16))) v[1] = 0.f;
                            if (!bf16_bits_positive((unsigned short)(mk.y & 0xffffu))) v[2] = 0.f; if (!bf16_bits_positive((unsigned short)(mk.y >> 16))) v[3] = 0.f;
                        } else {
                            const float4 mk = *reinterpret_cast<const float4*>(g.mask + (size_t)m * g.ldmask + n);
                            if (!(mk.x > 0.f)) v[0] = 0.f; if (!(mk.y > 0.f)) v[1] = 0.f; if (!(mk.z > 0.f)) v[2] = 0.f; if (!(mk.w > 0.f)) v[3] = 0.f;
                        }
                    }
                    if (c16) {
                        const unsigned lo = (unsigned)float_to_bf16_bits(v[0]) | ((unsigned)float_to_bf16_bits(v[1]) << 16);
                        const unsigned hi = (unsigned)float_to_bf16_bits(v[2]) | ((unsigned)float_to_bf16_bits(v[3]) << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + (size_t)m * g.ldc + n) = make_uint2(lo, hi);
                    } else {
                        *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ne = n + e;
                        if (ne >= g.N) continue;
                        float val = v[e] + (g.bias ? g.bias[ne] : 0.f);
                        if (g.act == 1) val = fmaxf(val, 0.f);
                        if (g.mask && !mask_on((size_t)m * g.ldmask + ne)) val = 0.f;
                        put(g.c_trans ? (size_t)ne * g.ldc + m : (size_t)m * g.ldc + ne, val);
                    }
                }
            }
    }
    if (do_colsum && tid < BM && m0 + tid < g.M) unsafeAtomicAdd(g.colsum + m0 + tid, csum);
}

// Marks for the build's listing check (tools/isa_asm_load_check.py, "counted waits"): CLIFT_MARK_DMA(tag) stands in front of a hand-issued LDS-DMA
// (it has no destination register the register guard could follow), CLIFT_MARK_USE(tag, allow) in front of the first read of the LDS slot that
// DMA fills, behind the counted s_waitcnt vmcnt(N) that publishes it: replaying the listing's vector-memory stream in order, at most `allow`
// DMAs of that tag may still be in flight there.  Comments in the listing: no instruction, no operand.
#define CLIFT_MARK_DMA(tag) asm volatile("; @dma " tag)
#define CLIFT_MARK_USE(tag, allow) asm volatile("; @use " tag " " allow)
