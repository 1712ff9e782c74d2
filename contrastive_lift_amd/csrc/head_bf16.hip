// head_bf16.hip -- bf16 mode: a whole xyz head forward in ONE persistent kernel (tensoRF.py:475-481, 576-579).
//   h1 = relu(W0 x + b0)            K = 3, generated in LDS (never read from memory)
//   h2 = relu(W1 h1 + b1)           256 x 256, bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16)
//   h3 = relu(W2 h2 + b2)           256 x 256
//   out = h3 Wout^T + bout          E <= 4 columns (instance heads) -- or, E = 0, h3 itself is the result (semantic head: its fourth
//                                   hidden layer and 22-wide output stay separate launches)
// At the bf16 MFMA rate a 64 x 256 x 256 tile is ~1 us of matrix-core time, so the per-layer kernels (layer_bf16.hip) are pure HBM
// streams: 512 B in + 512 B out per row and layer, 0.455 of the HBM peak at best.  Here the activations of a 64-row tile never leave
// the CU: they ping-pong between two 32 KB LDS tiles (bf16, the same 16-byte-chunk swizzle as layer_bf16.hip), the weights of both
// hidden layers live in registers (2 x 64 VGPRs per lane), the sample positions of the block's rows are staged in LDS, and memory
// sees 16 B per row in and 4 E B out -- plus 512 B per row and layer ONLY when the caller keeps the activations for a backward pass
// (h1 / h2 / h3 pointers non-null; bf16-stored exactly like the per-layer path writes them).
// Arithmetic is that of the per-layer path (fp32 K = 3 layer rounded to bf16, weights rounded to bf16 RNE on load, fp32 accumulate in
// the same k order, bias + ReLU in fp32, activations rounded to bf16 between layers; the output layer multiplies the bf16-rounded h3
// with bf16-rounded weights in fp32): results agree with it up to the summation order of the E-wide output layer.
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct HeadP {
    const float* x4;                  // (M, 4) normalised positions
    const float* W0; int ldw0; const float* b0;          // (256, 3)
    const float* W1; int ldw1; const float* b1;          // (256, 256)
    const float* W2; int ldw2; const float* b2;          // (256, 256)
    const float* Wout; int ldwo; const float* bout; int E;   // (E, 256), E in [0, 4]
    float* out; int ldo;              // (M, ldo) fp32, column offset applied by the caller (E > 0)
    unsigned short* h1; unsigned short* h2; unsigned short* h3;   // nullable bf16-stored (M, 256) activations
    int M;
};

constexpr int HB_ROWS = 64;                  // rows per tile
constexpr int HB_TILE = HB_ROWS * 32;        // uint4 per activation tile (64 rows x 512 B)
constexpr int HB_XROWS = 2048;               // positions staged per refill

static __device__ __forceinline__ bf16x8 hb_cvt8(const float4 a, const float4 b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}
static __device__ __forceinline__ unsigned hb_pack(float lo, float hi) {
    return (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
}
static __device__ __forceinline__ float hb_round(float v) { return bf16_bits_to_float(float_to_bf16_bits(v)); }

__global__ __launch_bounds__(512, 2) void k_head_bf16_fwd(HeadP p, int rows_per_block) {
    // LDS: two activation tiles | staged positions | output-layer weights (4 x 256 fp32, bf16-rounded) | the waves' shares of the output layer
    __shared__ __attribute__((aligned(1024))) uint4 lds[2 * HB_TILE + HB_XROWS + 256 + 8 * HB_ROWS + 128 + 256];
    uint4* const buf0 = lds;
    uint4* const buf1 = lds + HB_TILE;
    float4* const xs = reinterpret_cast<float4*>(lds + 2 * HB_TILE);
    float* const wl = reinterpret_cast<float*>(lds + 2 * HB_TILE + HB_XROWS);
    float4* const part = reinterpret_cast<float4*>(lds + 2 * HB_TILE + HB_XROWS + 256);       // part[wave * 64 + row]
    float* const bl = reinterpret_cast<float*>(lds + 2 * HB_TILE + HB_XROWS + 256 + 8 * HB_ROWS);          // biases of the two hidden layers: bl[layer * 256 + n]
    float4* const gl = reinterpret_cast<float4*>(lds + 2 * HB_TILE + HB_XROWS + 256 + 8 * HB_ROWS + 128);  // first layer: gl[n] = (W0[n][0..2], b0[n])
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(p.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + HB_ROWS - 1) / HB_ROWS;

    // ---- weights of both hidden layers: wave w owns output columns 32 w .. +31 for all 256 k (sixteen bf16x8 fragments per layer)
    bf16x8 w1[16], w2[16];
    {
        const int n = 32 * wave + li;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 16 * s + 8 * lh;
            const float4* q1 = reinterpret_cast<const float4*>(p.W1 + (size_t)n * p.ldw1 + k);
            const float4* q2 = reinterpret_cast<const float4*>(p.W2 + (size_t)n * p.ldw2 + k);
            w1[s] = hb_cvt8(q1[0], q1[1]);
            w2[s] = hb_cvt8(q2[0], q2[1]);
        }
    }
    // ---- small operands in LDS (registers are taken by the weights): hidden-layer biases, first-layer coefficients
    if (tid < 256) {
        bl[tid] = p.b1[tid];
        bl[256 + tid] = p.b2[tid];
        const float* wr0 = p.W0 + (size_t)tid * p.ldw0;
        gl[tid] = make_float4(wr0[0], wr0[1], wr0[2], p.b0[tid]);
    }
    for (int e = tid; e < 1024; e += 512) { const int c = e >> 8, k = e & 255; wl[e] = c < p.E ? hb_round(p.Wout[(size_t)c * p.ldwo + k]) : 0.f; }
    auto fill_positions = [&](int first) {
        const int n = min(HB_XROWS, rend - rbeg - first);
        for (int e = tid; e < n; e += 512) xs[e] = *reinterpret_cast<const float4*>(p.x4 + (size_t)(rbeg + first + e) * 4);
    };
    fill_positions(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int r0 = rbeg + t * HB_ROWS;
        if (t > 0 && (t & (HB_XROWS / HB_ROWS - 1)) == 0) {       // this tile starts the next 2048 staged positions
            __syncthreads();
            fill_positions(t * HB_ROWS);
            __syncthreads();
        }
        // ================= phase G: h1 rows 8 wave .. +7 of the tile -> buf0 (bf16; chunk c of row r in slot c ^ (r & 15))
        const float4 g0 = gl[4 * lane], g1 = gl[4 * lane + 1], g2 = gl[4 * lane + 2], g3 = gl[4 * lane + 3];     // this lane generates columns 4 lane .. +3
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * wave + i, gr = min(r0 + row, rend - 1);
            const float4 x = xs[(gr - rbeg) & (HB_XROWS - 1)];
            float o[4];
            o[0] = fmaxf(fmaf(g0.z, x.z, fmaf(g0.y, x.y, fmaf(g0.x, x.x, g0.w))), 0.f);
            o[1] = fmaxf(fmaf(g1.z, x.z, fmaf(g1.y, x.y, fmaf(g1.x, x.x, g1.w))), 0.f);
            o[2] = fmaxf(fmaf(g2.z, x.z, fmaf(g2.y, x.y, fmaf(g2.x, x.x, g2.w))), 0.f);
            o[3] = fmaxf(fmaf(g3.z, x.z, fmaf(g3.y, x.y, fmaf(g3.x, x.x, g3.w))), 0.f);
            const uint2 v = make_uint2(hb_pack(o[0], o[1]), hb_pack(o[2], o[3]));
            reinterpret_cast<uint2*>(buf0 + row * 32 + ((lane >> 1) ^ (row & 15)))[lane & 1] = v;
            if (p.h1 && r0 + row < rend) *reinterpret_cast<uint2*>(p.h1 + (size_t)gr * 256 + 4 * lane) = v;
        }
        __syncthreads();
        // ================= two hidden layers: src tile -> MFMA -> (+bias, ReLU, bf16) -> dst tile / output layer
        float pv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            const uint4* T = layer == 0 ? buf0 : buf1;
            uint4* D = layer == 0 ? buf1 : buf0;
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            // fragment (row li / 32 + li, k = 16 s + 8 lh ..): slot (2 s + lh) ^ (li & 15) = 2 s ^ (lh ^ (li & 15)), and a row starts on a 512-byte
            // boundary, so its address is ONE xor away from a per-lane base (a table of 32 addresses would not fit beside the weights)
            const unsigned abase = (unsigned)(uintptr_t)(lds_ptr_t)T + (unsigned)(li * 512 + ((lh ^ (li & 15)) * 16));
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                unsigned b = abase;
                asm volatile("" : "+v"(b));
                const unsigned a = b ^ (unsigned)(32 * s);
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) uint4*>((uintptr_t)a));
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) uint4*>((uintptr_t)(a + 32 * 512)));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(layer == 0 ? w1[s] : w2[s], a0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(layer == 0 ? w1[s] : w2[s], a1, acc[1], 0, 0, 0);
                if (s & 1) __builtin_amdgcn_sched_barrier(0);          // (fragments at most two steps ahead: the weights own the register file)
            }
            // lane (li, lh) holds row li of each 32-row half, columns 32 wave + 8 q + 4 lh + (0..3).  Column group outermost: its bias and its
            // output-layer weights are fetched from LDS once and serve both row halves.
            unsigned short* hs = layer == 0 ? p.h2 : p.h3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bl + layer * 256 + 32 * wave + 8 * q + 4 * lh);
                float4 wq[4];
                if (layer == 1 && p.E > 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) wq[c] = *reinterpret_cast<const float4*>(wl + c * 256 + 32 * wave + 8 * q + 4 * lh);
                }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const int row = 32 * x + li, m = r0 + row;
                    float v[4];
                    v[0] = fmaxf(acc[x][4 * q + 0] + bq.x, 0.f); v[1] = fmaxf(acc[x][4 * q + 1] + bq.y, 0.f);
                    v[2] = fmaxf(acc[x][4 * q + 2] + bq.z, 0.f); v[3] = fmaxf(acc[x][4 * q + 3] + bq.w, 0.f);
                    const uint2 pk = make_uint2(hb_pack(v[0], v[1]), hb_pack(v[2], v[3]));
                    if (layer == 0) reinterpret_cast<uint2*>(D + row * 32 + ((4 * wave + q) ^ (row & 15)))[lh] = pk;      // next layer's operand tile
                    if (hs && m < rend) *reinterpret_cast<uint2*>(hs + (size_t)m * 256 + 32 * wave + 8 * q + 4 * lh) = pk;
                    if (layer == 1 && p.E > 0) {          // output layer on the bf16-rounded activation: this lane's share of the E sums of row `row`
                        const float r0v = bf16_bits_to_float((unsigned short)(pk.x & 0xffffu)), r1v = bf16_bits_to_float((unsigned short)(pk.x >> 16));
                        const float r2v = bf16_bits_to_float((unsigned short)(pk.y & 0xffffu)), r3v = bf16_bits_to_float((unsigned short)(pk.y >> 16));
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            pv[x][c] = fmaf(r3v, wq[c].w, fmaf(r2v, wq[c].z, fmaf(r1v, wq[c].y, fmaf(r0v, wq[c].x, pv[x][c]))));
                    }
                }
            }
            if (layer == 0) __syncthreads();          // buf1 complete; buf0 no longer read
        }
        if (p.E > 0) {
            // fold the two half-waves, park the wave's share of rows li and 32 + li, then 4 lanes per row add the eight shares in a fixed order
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned u = __float_as_uint(pv[x][c]);
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    o[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                }
                if (lh == 0) part[wave * HB_ROWS + 32 * x + li] = make_float4(o[0], o[1], o[2], o[3]);
            }
            __syncthreads();
            if (tid < 4 * HB_ROWS) {
                const int row = tid >> 2, c = tid & 3, m = r0 + row;
                const float* ps = reinterpret_cast<const float*>(part) + row * 4 + c;
                float v = ((ps[0] + ps[4 * HB_ROWS]) + (ps[8 * HB_ROWS] + ps[12 * HB_ROWS])) +
                          ((ps[16 * HB_ROWS] + ps[20 * HB_ROWS]) + (ps[24 * HB_ROWS] + ps[28 * HB_ROWS]));
                if (c < p.E && m < rend) p.out[(size_t)m * p.ldo + c] = v + (p.bout ? p.bout[c] : 0.f);
            }
        }
        __syncthreads();                              // buf0 / part free for the next tile
    }
}

// x4 (M,4) fp32; W0 (256,3) / W1, W2 (256,256) / Wout (E,256) fp32 with the given row pitches; h1/h2/h3: nullable bf16-stored (M,256)
// destinations of the hidden activations (kept for a backward pass); E > 0: out (M, ldo) fp32 receives the E-wide output layer;
// E = 0: no output layer (h3 must then be non-null: it is the result).
extern "C" int clift_xyz_head_bf16_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1, const float* b1,
                                       const float* W2, int ldw2, const float* b2, const float* Wout, int ldwo, const float* bout, int E, int M,
                                       void* h1, void* h2, void* h3, float* out, int ldo, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 0 && E <= 4, "clift_xyz_head_bf16_fwd: E must be in [0,4] (got %d)", E);
    CLIFT_REQUIRE(E > 0 || h3 != nullptr, "clift_xyz_head_bf16_fwd: with E = 0 the third activation is the result and needs a destination");
    CLIFT_REQUIRE((((uintptr_t)x4) & 15) == 0 && (((uintptr_t)W1) & 15) == 0 && (((uintptr_t)W2) & 15) == 0 && ldw1 % 4 == 0 && ldw2 % 4 == 0,
                  "clift_xyz_head_bf16_fwd: x4 / W1 / W2 must be 16-byte aligned with pitches that are multiples of 4");
    CLIFT_REQUIRE(((((uintptr_t)h1) | ((uintptr_t)h2) | ((uintptr_t)h3)) & 7) == 0, "clift_xyz_head_bf16_fwd: activation destinations must be 8-byte aligned");
    HeadP p = {x4, W0, ldw0, b0, W1, ldw1, b1, W2, ldw2, b2, Wout, ldwo, bout, E, out, ldo,
               reinterpret_cast<unsigned short*>(h1), reinterpret_cast<unsigned short*>(h2), reinterpret_cast<unsigned short*>(h3), M};
    const int tiles = cdiv(M, HB_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), HB_ROWS) * HB_ROWS;
    k_head_bf16_fwd<<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(p, rpb);
    return clift_check_launch("clift_xyz_head_bf16_fwd");
}
