// layer_bf16.hip -- the 256 -> 256 hidden layers of the bf16 mode (forward: C = relu(A W^T + b); dgrad: C = mask . (A W)), with the
// streamed tensors (A, C, mask) bf16-STORED.  At bf16 MFMA rate a 64 x 256 x 256 tile is ~1 us of matrix-core time, so these
// launches are HBM streams (512 B in + 512 B out per row, + 512 B of mask in the dgrad) and the generic tiled kernel
// (gemm_bf16.hip: 8 k-tiles per block, one global round trip each, 256 KB of weights re-read per 128 rows) spends its time on
// latency, not bandwidth.  Here instead:
//   * one persistent 8-wave block per CU, each owning a contiguous range of rows (balanced to 32 rows: no tail wave of tiles);
//   * the whole weight matrix lives in REGISTERS: wave w owns output columns 32 w .. +31 for all 256 k, i.e. sixteen bf16x8
//     B-fragments = 64 VGPRs, loaded (and rounded from fp32) once per block;
//   * the activation rows (and the mask rows) stream through a ring of 32 KB LDS buffers filled by LDS-DMA
//     (global_load_lds_dwordx4: no staging VGPRs), DEPTH tiles ahead of the multiply; waits are counted (s_waitcnt vmcnt(N) with
//     N = the vector-memory instructions this wave issued after the tile's DMA: later DMAs and the 16-byte output stores, which
//     complete in order on gfx9-family hardware) so neither the stores nor the younger DMAs are drained at the tile boundary.
//     The LDS image is lane-linear (a DMA constraint), so the bank swizzle is applied at the SOURCE: 16-byte chunk c of row r
//     lands in slot c ^ (r & 15), which makes the ds_read_b128 fragment reads (16 consecutive rows, same k) conflict-free;
//   * operands swapped in the MFMA (weights first): a lane owns one output row; v_permlane32_swap pairs the two half-waves'
//     4-column groups into 8 consecutive bf16 so that every store / mask read is 16 bytes per lane.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_bf16)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4k __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int LY_ROWS = 64;                 // rows per streamed tile (two 32-row MFMA tiles per wave)
constexpr int LY_TILE = LY_ROWS * 32;       // uint4 per tile: 64 rows x 512 B

static __device__ __forceinline__ bf16x8 cvt8(const float4 a, const float4 b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}
static __device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
}
static __device__ __forceinline__ unsigned keep_positive(unsigned v, unsigned mk) {      // per 16-bit half: v where the mask half is > 0
    const unsigned lo = bf16_bits_positive((unsigned short)(mk & 0xffffu)) ? 0x0000ffffu : 0u;
    const unsigned hi = bf16_bits_positive((unsigned short)(mk >> 16)) ? 0xffff0000u : 0u;
    return v & (lo | hi);
}
template <int N>
static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
static __device__ __forceinline__ void wait_vm_upto(int n) {       // wave-uniform n, a multiple of 4 (everything here issues in fours)
    if (n >= 36) wait_vm<36>();
    else if (n >= 32) wait_vm<32>();
    else if (n >= 28) wait_vm<28>();
    else if (n >= 24) wait_vm<24>();
    else if (n >= 20) wait_vm<20>();
    else if (n >= 16) wait_vm<16>();
    else if (n >= 12) wait_vm<12>();
    else if (n >= 8) wait_vm<8>();
    else if (n >= 4) wait_vm<4>();
    else wait_vm<0>();
}

// K3W (dgrad only; the bf16 counterpart of k_layer_f32<dgrad, K3W> / k_layer_x6<dgrad, K3W>): the layer is the SECOND layer of an xyz head, so its
// input gradient dH1 = (h1 > 0) . (dH2 W1) has one consumer, the K = 3 first layer's weight gradient gW0[n][0..2] += sum_m dH1[m][n] x_m, gb0[n] +=
// sum_m dH1[m][n] (tensoRF.py:475,576 backward).  A lane of this kernel owns one row of each 32-row half of a tile and, after the permlane
// swap, 2 x 8 consecutive columns of it: it keeps 16 x (x, y, z, 1) running sums over ITS rows for the whole row range (64 registers -- the
// kernel allocates the whole register file anyway, see the ballast), on the values the unfused path would have stored (rounded to bf16, masked),
// so the products are the same and only their summation order differs.  dH1 is never written; the tile's 64 positions (16 B each) travel by one more LDS-DMA
// instruction of wave 0 into a slot beside the tile and are read back in the epilogue.  (A first form loaded them into registers with inline asm
// at the top of the tile and waited for them before the epilogue: between the two statements the compiler, at 256 registers, moved the not yet
// written registers -- wrong sums in some launches of one size only.  Nothing that is still in flight may live in a compiler-visible register.)  The 32 lanes of a half-wave are folded once per block with xor shuffles; one atomic per column and
// component into this XCD's gradient shard.  Removes the k_wgrad_narrow_stream<true> launch that re-read dH1 (53 us at 249 k rows) and the 128 MB
// store in front of it.
struct K3B {
    const float* x4;      // (M, 4) normalised sample positions
    float* gW0;           // (256, 3), row pitch ldg
    int ldg;
    float* gb0;           // (256)
};

// DGRAD = false: forward, weights stored [n][k], bias + ReLU;  DGRAD = true: weights stored [k][n], bf16-stored ReLU mask.
// DEPTH = tiles in flight ahead of the multiply; the ring has DEPTH + 1 stages of 32 KB (A) [+ 32 KB (mask)].
template <bool DGRAD, int DEPTH, bool K3W = false>
__global__ __launch_bounds__(512, 2) void k_layer_bf16(GemmP g, int rows_per_block, K3B kb = K3B{}) {
    static_assert(!K3W || DGRAD, "K3W is a dgrad form");
    asm volatile("v_mov_b32 v255, 0" ::: "v255");       // register ballast (csrc/layer_x6w.hip): nothing else is scheduled onto this SIMD beside the bf16 MFMA stream
    constexpr int NST = DEPTH + 1, STAGE = (DGRAD ? 2 : 1) * LY_TILE, PER_DMA = DGRAD ? 8 : 4;
    __shared__ __attribute__((aligned(16))) uint4 lds[NST * STAGE + (K3W ? NST * LY_ROWS : 0)];        // the only LDS object of the kernel (K3W: + the tiles' positions)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + LY_ROWS - 1) / LY_ROWS;

    bf16x8 w[16];
    float bias[16];
    {
        const int n = 32 * wave + li;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 16 * s + 8 * lh;
            if (!DGRAD) {
                const float4* q = reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + k);
                w[s] = cvt8(q[0], q[1]);
            } else {
                const float* q = g.B + (size_t)k * g.ldb + n;
                const size_t ld = (size_t)g.ldb;
                w[s] = cvt8(make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]), make_float4(q[4 * ld], q[5 * ld], q[6 * ld], q[7 * ld]));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = (!DGRAD && g.bias) ? g.bias[32 * wave + 8 * q + 4 * lh + e] : 0.f;
    }
    wait_vm<0>();                                                           // ordinary loads are done before the first DMA is issued
    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* K16 = reinterpret_cast<const unsigned short*>(g.mask);
    unsigned short* C16 = reinterpret_cast<unsigned short*>(g.C);
    auto dma = [&](int t) {                                                 // tile t of this block -> stage t % NST
        const int r0 = rbeg + t * LY_ROWS;
        uint4* st = lds + (t % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = wave * 4 + i;                                  // wave-uniform: 1 KB = two rows per instruction
            const int row = inst * 2 + lh, c = li ^ (row & 15);              // LDS slot li of the row <- source chunk c
            const int gr = min(r0 + row, rend - 1);                          // rows past the range re-read its last row (never stored)
            __builtin_amdgcn_global_load_lds(A16 + (size_t)gr * g.lda + c * 8, (lds_ptr_t)(st + inst * 64), 16, 0, 0);
            if (DGRAD) __builtin_amdgcn_global_load_lds(K16 + (size_t)gr * g.ldmask + c * 8, (lds_ptr_t)(st + LY_TILE + inst * 64), 16, 0, 0);
        }
        if (K3W && wave == 0)           // the tile's 64 positions (16 B each): one more DMA instruction of wave 0, lane = row
            __builtin_amdgcn_global_load_lds(kb.x4 + (size_t)min(r0 + lane, rend - 1) * 4, (lds_ptr_t)(lds + NST * STAGE + (t % NST) * LY_ROWS), 16, 0, 0);
    };
    // vector-memory instructions issued by this wave after the DMA of tile t, at the time tile t is needed (the DMA of tile
    // t + DEPTH is issued only after that wait): DMAs of tiles t+1 .. min(t+DEPTH-1, ntiles-1) and the stores (4 per tile) of
    // tiles max(0, t-DEPTH) .. t-1.  Anything counted here that was not actually issued would let the wait pass early.
    float ks[K3W ? 2 : 1][K3W ? 8 : 1][4];      // K3W: running sums of column group qp, column c: (sum d x, sum d y, sum d z, sum d)
    if (K3W) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) { ks[a][c][0] = 0.f; ks[a][c][1] = 0.f; ks[a][c][2] = 0.f; ks[a][c][3] = 0.f; }
    }
    for (int t = 0; t < DEPTH && t < ntiles; ++t) dma(t);
    for (int t = 0; t < ntiles; ++t) {
        // (K3W: no stores; wave 0 issues a ninth DMA instruction per tile, the positions -- wait_vm_upto rounds down to a multiple of four, which only waits longer)
        const int younger = (PER_DMA + ((K3W && wave == 0) ? 1 : 0)) * (min(t + DEPTH - 1, ntiles - 1) - t) + (K3W ? 0 : 4 * (t - max(0, t - DEPTH)));
        wait_vm_upto(younger);                                              // this wave's part of tile t has landed
        __builtin_amdgcn_s_barrier();                                        // ... and everyone's; everyone is done reading stage (t-1) % NST
        asm volatile("" ::: "memory");
        if (t + DEPTH < ntiles) dma(t + DEPTH);                              // refill the stage that tile t-1 just vacated
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        const uint4* T = lds + (t % NST) * STAGE;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int slot = (2 * s + lh) ^ (li & 15);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, T[li * 32 + slot]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, T[(li + 32) * 32 + slot]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], a0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], a1, acc[1], 0, 0, 0);
        }
        // epilogue: lane (li, lh) holds row li of each 32-row tile, columns 32 wave + 8 q + 4 lh + (0..3) for q = 0..3
        const int r0 = rbeg + t * LY_ROWS;
        uint4 mk[4];
        if (DGRAD) {
            // the mask rows were DMA'd next to the A rows.  Read them with inline asm: beside an in-flight LDS-DMA the compiler
            // guards every ordinary read of this array with s_waitcnt vmcnt(0) -- which would drain the prefetched tile and the stores
            unsigned ad[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = (4 * wave + 2 * (j & 1) + lh) ^ (li & 15);
                ad[j] = (unsigned)(uintptr_t)(lds_ptr_t)(T + LY_TILE + (32 * (j >> 1) + li) * 32 + slot);
            }
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(mk[0]), "=&v"(mk[1]), "=&v"(mk[2]), "=&v"(mk[3])
                         : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                         : "memory");
        }
        f32x4k px0, px1;
        if (K3W) {                           // this lane's two rows' positions, from the tile's position slot (inline asm for the reason given at the mask reads)
            const unsigned pa = (unsigned)(uintptr_t)(lds_ptr_t)(lds + NST * STAGE + (t % NST) * LY_ROWS + li);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=&v"(px0), "=&v"(px1) : "v"(pa) : "memory");
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int m = r0 + 32 * x + li;
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned P0[2], P1[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * qp + h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[x][4 * q + e] + bias[4 * q + e];
                        if (!DGRAD && g.act == 1) v[e] = fmaxf(v[e], 0.f);
                    }
                    (h ? P1 : P0)[0] = pack_bf16(v[0], v[1]);
                    (h ? P1 : P0)[1] = pack_bf16(v[2], v[3]);
                }
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(P0[0], P1[0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(P0[1], P1[1], false, false);
                uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);          // 8 consecutive columns from 32 wave + 8 (2 qp + lh)
                if (DGRAD) {
                    const uint4 k4 = mk[2 * x + qp];
                    o.x = keep_positive(o.x, k4.x); o.y = keep_positive(o.y, k4.y);
                    o.z = keep_positive(o.z, k4.z); o.w = keep_positive(o.w, k4.w);
                }
                if (K3W) {
                    if (m < rend) {              // rows past the range are copies of its last row: they count for nothing
                        const f32x4k p = x ? px1 : px0;
                        const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned u = ow[c >> 1];
                            const float d = __uint_as_float((c & 1) ? (u & 0xffff0000u) : (u << 16));
                            ks[qp][c][0] = fmaf(d, p[0], ks[qp][c][0]); ks[qp][c][1] = fmaf(d, p[1], ks[qp][c][1]);
                            ks[qp][c][2] = fmaf(d, p[2], ks[qp][c][2]); ks[qp][c][3] += d;
                        }
                    }
                } else if (m < rend) *reinterpret_cast<uint4*>(C16 + (size_t)m * g.ldc + 32 * wave + 8 * (2 * qp + lh)) = o;
            }
        }
    }
    if (K3W) {
        // fold the 32 rows (lanes li) of each half-wave; lane li == 0 of a half then holds the block's sums of its 2 x 8 columns
        float* const gw = grad_target(kb.gW0);
        float* const gb = grad_target(kb.gb0);
#pragma unroll
        for (int qp = 0; qp < 2; ++qp)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = ks[qp][c][e];
#pragma unroll
                    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                    ks[qp][c][e] = v;
                }
                if (li == 0) {
                    const int n = 32 * wave + 8 * (2 * qp + lh) + c;
                    unsafeAtomicAdd(gw + (size_t)n * kb.ldg + 0, ks[qp][c][0]);
                    unsafeAtomicAdd(gw + (size_t)n * kb.ldg + 1, ks[qp][c][1]);
                    unsafeAtomicAdd(gw + (size_t)n * kb.ldg + 2, ks[qp][c][2]);
                    if (gb) unsafeAtomicAdd(gb + n, ks[qp][c][3]);
                }
            }
    }
}

// Backward of the first TWO layers of an xyz head in bf16 mode, the part behind the second layer's weight gradient (ABI 16): dH2 (M, ldd)
// bf16-stored, W1 (256, 256) fp32 [k = output of layer 1 ... stored (out, in) = the dgrad's [k][n]], h1 (M, ldm) bf16-stored first activation (the
// ReLU mask), x4 (M, 4) positions: gW0 (256, ldg) += ((h1 > 0) . bf16(dH2 W1))^T x4[:, :3], gb0 += its column sums.  dH1 is never written.
extern "C" int clift_xyz_head_first2_bf16_bwd(const void* dH2, int ldd, const float* W1, int ldw1, const void* h1, int ldm, const float* x4, int M,
                                              float* gW0, int ldg, float* gb0, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(ldd % 8 == 0 && ldm % 8 == 0 && ldd >= 256 && ldm >= 256 && ldw1 >= 256 && ldg >= 3, "clift_xyz_head_first2_bf16_bwd: pitches (bf16 rows: multiples of 8, >= 256)");
    CLIFT_REQUIRE(((((uintptr_t)dH2) | ((uintptr_t)h1) | ((uintptr_t)x4)) & 15) == 0, "clift_xyz_head_first2_bf16_bwd: 16-byte aligned dH2 / h1 / x4 required");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = reinterpret_cast<const float*>(dH2); p.lda = ldd; p.B = W1; p.ldb = ldw1;
    p.mask = reinterpret_cast<const float*>(h1); p.ldmask = ldm;
    K3B kb = {x4, gW0, ldg, gb0};
    const int tiles = cdiv(M, LY_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), 32) * 32;
    k_layer_bf16<true, 1, true><<<blocks, 512, 0, as_stream(s)>>>(p, rpb, kb);
    return clift_check_launch("clift_xyz_head_first2_bf16_bwd");
}

// Eligibility is decided by the caller (gemm_bf16.hip): N = K = 256, plain A, bf16-stored A / C (/ mask), 16-byte-aligned rows.
int clift_layer_bf16_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, LY_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();                    // one persistent block per CU
    const int rpb = cdiv(cdiv(p.M, blocks), 32) * 32;
    if (b_trans) k_layer_bf16<true, 1><<<blocks, 512, 0, st>>>(p, rpb);
    else k_layer_bf16<false, 3><<<blocks, 512, 0, st>>>(p, rpb);
    return clift_check_launch("clift_gemm(bf16 layer)");
}

// ============================================================================ streamed weight gradient of the same layers
// gW[n][k] += sum_m dY[m][n] X[m][k]  (+ gb[n] += sum_m dY[m][n]),  dY and X bf16-stored [m][256].  Both MFMA operands want 8
// consecutive m per lane, i.e. a COLUMN of the row-major tiles: the fragments are gathered with ds_read_b64_tr_b16 (a 16-lane
// group reads a [4 m][16 col] block, lane q receives column q), two per fragment.  One persistent block per CU owns a
// contiguous range of rows and the full 256 x 256 product (8 waves = 2 (n) x 4 (k), 128 x 64 per wave = 128 accumulator
// VGPRs); row tiles of 64 stream through a two-stage LDS ring by LDS-DMA (dY tile + X tile = 64 KB per stage).
// Swizzle (at the DMA source, the image being lane-linear): 16-byte chunk c of row r sits in slot c ^ ((r & 3) << 2), so the four
// rows of a transposed read land in four different 64-byte bank spans.  All LDS reads are inline asm (see the mask reads above).
// The partial products are added to gW with fp32 atomics at the end (the same count as the split-K launch this replaces).
static __device__ __forceinline__ uint2 tr_read(unsigned addr) {
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
static __device__ __forceinline__ float bf16_pair_sum(unsigned u) { return __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u); }

__global__ __launch_bounds__(512, 2) void k_wgrad_bf16_stream(GemmP g, int rows_per_block) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");       // register ballast (csrc/layer_x6w.hip): nothing else is scheduled onto this SIMD beside the bf16 MFMA stream
    constexpr int STAGE = 2 * LY_TILE;                                        // uint4 per stage: dY tile then X tile
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * STAGE];            // 128 KB, the only LDS object
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wn = wave >> 2, wk = wave & 3;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.K, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + LY_ROWS - 1) / LY_ROWS;
    const unsigned short* Y16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* X16 = reinterpret_cast<const unsigned short*>(g.B);
    auto dma = [&](int t) {
        const int r0 = rbeg + t * LY_ROWS;
        uint4* st = lds + (t & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = wave * 4 + i;
            const int row = inst * 2 + lh, c = li ^ ((row & 3) << 2);
            const int gr = min(r0 + row, rend - 1);                          // a ragged last tile is zero-filled after it lands
            __builtin_amdgcn_global_load_lds(Y16 + (size_t)gr * g.lda + c * 8, (lds_ptr_t)(st + inst * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(X16 + (size_t)gr * g.ldb + c * 8, (lds_ptr_t)(st + LY_TILE + inst * 64), 16, 0, 0);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float csum = 0.f;
    // lane-constant part of the fragment addresses (bytes): row 8 lh + ((lane & 15) >> 2), column block by tile, 8-byte half
    const int s = (lane >> 2) & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const unsigned lrow = (unsigned)((8 * lh + ((lane & 15) >> 2)) * 512 + ((lane >> 4) & 1) * 32 + ((lane & 3) >> 1) * 16 + (lane & 1) * 8);
    unsigned ya[4], xa[2];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) ya[tn] = lds0 + lrow + (unsigned)((16 * wn + 4 * (tn ^ s)) * 16);
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) xa[tk] = lds0 + LY_TILE * 16 + lrow + (unsigned)(((8 * wk + 4 * tk) ^ (4 * s)) * 16);

    dma(0);
    for (int t = 0; t < ntiles; ++t) {
        wait_vm<0>();                                                        // tile t (this wave's part) has landed; nothing else is in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) dma(t + 1);
        const int valid = rend - (rbeg + t * LY_ROWS);
        if (valid < LY_ROWS) {                                               // zero the rows past the end (only the last tile of the last block)
            uint4* st = lds + (t & 1) * STAGE;
            for (int e = tid; e < (LY_ROWS - valid) * 32; e += 512) {
                st[valid * 32 + e] = make_uint4(0u, 0u, 0u, 0u);
                st[LY_TILE + valid * 32 + e] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
        const unsigned so = (unsigned)((t & 1) * STAGE * 16);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            uint2 yr[4][2], xr[2][2];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int p = 0; p < 2; ++p) yr[tn][p] = tr_read(ya[tn] + so + (unsigned)((16 * ms + 4 * p) * 512));
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int p = 0; p < 2; ++p) xr[tk][p] = tr_read(xa[tk] + so + (unsigned)((16 * ms + 4 * p) * 512));
            // the wait must be a data dependency of every fragment: a bare asm wait does not stop the compiler from scheduling the
            // MFMAs (plain register consumers of the read results) above it
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(yr[0][0]), "+v"(yr[0][1]), "+v"(yr[1][0]), "+v"(yr[1][1]), "+v"(yr[2][0]), "+v"(yr[2][1]), "+v"(yr[3][0]), "+v"(yr[3][1]),
                           "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1])
                         :
                         : "memory");
            bf16x8 a[4], b[2];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) a[tn] = __builtin_bit_cast(bf16x8, make_uint4(yr[tn][0].x, yr[tn][0].y, yr[tn][1].x, yr[tn][1].y));
#pragma unroll
            for (int tk = 0; tk < 2; ++tk) b[tk] = __builtin_bit_cast(bf16x8, make_uint4(xr[tk][0].x, xr[tk][0].y, xr[tk][1].x, xr[tk][1].y));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) acc[tn][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tn], b[tk], acc[tn][tk], 0, 0, 0);
            if (g.colsum) {      // bias gradient: wave (wn, wk) sums the dY columns of its n-tile tn == wk (spread over the four k-waves)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    if (tn == wk) csum += (bf16_pair_sum(yr[tn][0].x) + bf16_pair_sum(yr[tn][0].y)) + (bf16_pair_sum(yr[tn][1].x) + bf16_pair_sum(yr[tn][1].y));
            }
        }
    }
    // lane (li, lh) holds gW rows n = 128 wn + 32 tn + 8 q + 4 lh + e, column k = 64 wk + 32 tk + li
    g.C = grad_target(g.C); g.colsum = grad_target(g.colsum);                // (this XCD's shard when a pass has them on)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 128 * wn + 32 * tn + 8 * (r >> 2) + 4 * lh + (r & 3), k = 64 * wk + 32 * tk + li;
                unsafeAtomicAdd(g.C + (size_t)n * g.ldc + k, acc[tn][tk][r]);
            }
    if (g.colsum) {          // fold the two m-halves (lanes l, l + 32) first: one atomic per address and wave
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(csum), __float_as_uint(csum), false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0) unsafeAtomicAdd(g.colsum + 128 * wn + 32 * wk + li, tot);
    }
}

// Eligibility decided by the caller: M = N = 256 (gW is 256 x 256), both streamed operands bf16-stored with 16-byte-aligned rows.
int clift_wgrad_bf16_stream2d_launch(const GemmP& p, hipStream_t st);
int clift_wgrad_bf16_stream_launch(const GemmP& p, hipStream_t st) {
    if (p.K >= 4096) return clift_wgrad_bf16_stream2d_launch(p, st);      // 64 row ranges x 4 column slices
    const int tiles = cdiv(p.K, LY_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(p.K, blocks), LY_ROWS) * LY_ROWS;
    k_wgrad_bf16_stream<<<cdiv(p.K, rpb), 512, 0, st>>>(p, rpb);
    return clift_check_launch("clift_gemm(bf16 wgrad stream)");
}

// ---------------------------------------------------------------------------- the same weight gradient cut in two dimensions
// The one-dimensional kernel above ends with every CU adding a full 256 x 256 partial to gW: 1024 atomic wave-instructions per CU,
// ~45-50 us per launch whatever the atomic scope -- more than the 33 us it takes to stream 249 k rows.  Here: 64 row ranges x 4 column
// slices of 64 (256 blocks, one per CU); a block streams its rows of dY (all 256 columns) and of its X slice and ends with a 256 x 64
// partial (a quarter of the atomics).  The four slice-blocks of a range have block ids 8 apart = the same XCD, so dY comes from HBM once
// and from that L2 three times.  Wave w owns dY columns 32 w .. +31 (gW rows) for both 32-column halves of the slice.
// X-slice image: rows of 128 B (8 chunks); chunk c of row r sits in slot c ^ (((r >> 1) & 1) << 1), which puts the four rows of a
// transposed read in four different 32-byte bank spans.
constexpr int W2_XB = LY_ROWS * 128, W2_STAGE = LY_TILE * 16 + W2_XB, W2_DEPTH = 2, W2_STAGES = W2_DEPTH + 1;     // bytes

__global__ __launch_bounds__(512, 2) void k_wgrad_bf16_stream2d(GemmP g, int rows_per_range) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");       // register ballast (csrc/layer_x6w.hip): nothing else is scheduled onto this SIMD beside the bf16 MFMA stream
    __shared__ __attribute__((aligned(16))) unsigned char lds[W2_STAGES * W2_STAGE];          // 120 KB, the only LDS object
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x, slice = (b >> 3) & 3, range = (b & 7) + 8 * (b >> 5);
    const int rbeg = range * rows_per_range, rend = min(g.K, rbeg + rows_per_range);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + LY_ROWS - 1) / LY_ROWS;
    const unsigned short* Y16 = reinterpret_cast<const unsigned short*>(g.A);
    const unsigned short* X16 = reinterpret_cast<const unsigned short*>(g.B) + 64 * slice;
    auto dma = [&](int t) {
        const int r0 = rbeg + t * LY_ROWS;
        unsigned char* st = lds + (t % W2_STAGES) * W2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = wave * 4 + i;
            const int row = inst * 2 + lh, c = li ^ ((row & 3) << 2);
            const int gr = min(r0 + row, rend - 1);
            __builtin_amdgcn_global_load_lds(Y16 + (size_t)gr * g.lda + c * 8, (lds_ptr_t)(st + inst * 1024), 16, 0, 0);
        }
        const int row = wave * 8 + (lane >> 3), c = (lane & 7) ^ (((row >> 1) & 1) << 1);
        const int gr = min(r0 + row, rend - 1);
        __builtin_amdgcn_global_load_lds(X16 + (size_t)gr * g.ldb + c * 8, (lds_ptr_t)(st + LY_TILE * 16 + wave * 1024), 16, 0, 0);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float csum = 0.f;
    // lane-constant fragment addresses (bytes): row 8 lh + ((lane & 15) >> 2) of a 16-row step, 16-column half (lane >> 4) & 1, 4-column quad lane & 3
    const int s4 = (lane >> 2) & 3;                 // row & 3
    const int s2 = (lane >> 3) & 1;                 // (row >> 1) & 1
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const int frow = 8 * lh + ((lane & 15) >> 2);
    const unsigned ya = lds0 + (unsigned)(frow * 512 + (((4 * wave) ^ (4 * s4)) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 16 + (lane & 1) * 8);
    unsigned xa[2];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
        xa[tk] = lds0 + (unsigned)(LY_TILE * 16 + frow * 128 + (((4 * tk + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) ^ (2 * s2)) * 16) + (lane & 1) * 8);

    for (int t = 0; t < W2_DEPTH && t < ntiles; ++t) dma(t);
    for (int t = 0; t < ntiles; ++t) {
        if (min(t + W2_DEPTH - 1, ntiles - 1) > t) wait_vm<5>(); else wait_vm<0>();           // one younger tile (5 DMAs) stays in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + W2_DEPTH < ntiles) dma(t + W2_DEPTH);
        const int valid = rend - (rbeg + t * LY_ROWS);
        if (valid < LY_ROWS) {                                               // zero the dY rows past the end (last tile of a range)
            uint4* st = reinterpret_cast<uint4*>(lds + (t % W2_STAGES) * W2_STAGE);
            for (int e = tid; e < (LY_ROWS - valid) * 32; e += 512) st[valid * 32 + e] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
        const unsigned so = (unsigned)((t % W2_STAGES) * W2_STAGE);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            uint2 yr[2], xr[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) yr[p] = tr_read(ya + so + (unsigned)((16 * ms + 4 * p) * 512));
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int p = 0; p < 2; ++p) xr[tk][p] = tr_read(xa[tk] + so + (unsigned)((16 * ms + 4 * p) * 128));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yr[0]), "+v"(yr[1]), "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]) : : "memory");
            const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(yr[0].x, yr[0].y, yr[1].x, yr[1].y));
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, make_uint4(xr[0][0].x, xr[0][0].y, xr[0][1].x, xr[0][1].y));
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, make_uint4(xr[1][0].x, xr[1][0].y, xr[1][1].x, xr[1][1].y));
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
            if (g.colsum && slice == 0) csum += (bf16_pair_sum(yr[0].x) + bf16_pair_sum(yr[0].y)) + (bf16_pair_sum(yr[1].x) + bf16_pair_sum(yr[1].y));
        }
    }
    g.C = grad_target(g.C); g.colsum = grad_target(g.colsum);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 32 * wave + 8 * (r >> 2) + 4 * lh + (r & 3);
        float* dst = g.C + (size_t)n * g.ldc + 64 * slice + li;
        unsafeAtomicAdd(dst, acc0[r]);
        unsafeAtomicAdd(dst + 32, acc1[r]);
    }
    if (g.colsum && slice == 0) {
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(csum), __float_as_uint(csum), false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0) unsafeAtomicAdd(g.colsum + 32 * wave + li, tot);
    }
}

int clift_wgrad_bf16_stream2d_launch(const GemmP& p, hipStream_t st) {
    const int rpr = cdiv(cdiv(p.K, 64), LY_ROWS) * LY_ROWS;
    k_wgrad_bf16_stream2d<<<256, 512, 0, st>>>(p, rpr);
    return clift_check_launch("clift_gemm(bf16 wgrad stream 2-D)");
}
