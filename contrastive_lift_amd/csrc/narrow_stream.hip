// narrow_stream.hip -- weight gradient of the narrow OUTPUT layers (out_features <= 32: 22 semantic classes, 3 instance
// dimensions) over a 256-wide hidden activation:  gW[c][j] += sum_m dY[m][c] X[m][j],  gb[c] += sum_m dY[m][c].
// One read of X (M x 256, fp32 or bf16-stored) -- an HBM stream.  The VALU kernel (k_wgrad_narrow, gemm.hip) is FMA-bound at
// ~1/3 of that rate (22 x 256 FMAs per row); here the products run on the fp32 matrix cores instead (v_mfma_f32_32x32x2_f32: the
// 32 x 32 tile is padded in the class dimension only, which costs MFMA time the stream has to spare): persistent blocks, row tiles
// by LDS-DMA two tiles ahead, wave w owns columns 32 w .. +31 of X for ALL classes (no cross-wave reduction), each lane reads one
// dY and one X element per MFMA step (row-contiguous, conflict-free without any swizzle).  bf16-stored X is widened in the
// register (exact), so both storage modes give fp32 products.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(narrow_stream)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int NS_ROWS = 32;              // rows per tile = 16 MFMA steps of 2 rows
constexpr int NS_DEPTH = 2;              // tiles in flight ahead of the multiply
constexpr int NS_STAGES = NS_DEPTH + 1;

static __device__ __forceinline__ void ns_wait_vm(int n) {          // wave-uniform n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
}

// XB = X is bf16-stored.  Stage layout (bytes): X tile (32 rows x 256 elements) then the dY tile (32 rows x ldd floats, ldd <= 32).
template <bool XB>
__global__ __launch_bounds__(512, 2) void k_wgrad_narrow_stream(const float* __restrict__ dY, int ldd, int no, const float* __restrict__ X, int ldx,
                                                                int M, int rows_per_block, float* __restrict__ gW, long sc, long sn, float* __restrict__ gb,
                                                                int ones_class, float* __restrict__ ones_row, int nx) {
    // nx = columns of X (<= 256, a multiple of 4; 256 when bf16-stored): waves whose 32 columns start past nx only copy and wait.
    // gW[c * sc + j * sn] += sum_m dY[m][c] X[m][j].  ones_class >= 0: that class reads as 1.0 whatever dY holds and its row goes to
    // ones_row[j] (+= sum_m X[m][j]) instead of gW -- the K = 3 first-layer backward: dY = the sample positions (x, y, z, pad),
    // X = the hidden gradient, ones_row = the bias gradient.
    constexpr int XBYTES = NS_ROWS * 256 * (XB ? 2 : 4), STAGE = XBYTES + NS_ROWS * 32 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS_STAGES * STAGE];          // the only LDS object
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    if (rows_limited()) {          // sync-free step: re-balance the row ranges over the true row count (see layer_f32.hip)
        M = limit_rows(M);
        rows_per_block = ((M + (int)gridDim.x - 1) / (int)gridDim.x + NS_ROWS - 1) / NS_ROWS * NS_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + NS_ROWS - 1) / NS_ROWS;
    const int dchunks = NS_ROWS * ldd / 4;                                   // 16-byte chunks of a dY tile (<= 256)
    const bool has_d = wave * 64 < dchunks;                                   // this wave issues one dY DMA instruction per tile
    const int per_tile = (XB ? 2 : 4) + (has_d ? 1 : 0);                      // vector-memory instructions of this wave per tile
    auto dma = [&](int t) {
        const int r0 = rbeg + t * NS_ROWS;
        unsigned char* st = lds + (t % NS_STAGES) * STAGE;
        if (!XB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 4 + i, gr = min(r0 + row, rend - 1);
                __builtin_amdgcn_global_load_lds(X + (size_t)gr * ldx + min(lane * 4, nx - 4), (lds_ptr_t)(st + row * 1024), 16, 0, 0);
            }
        } else {
            const unsigned short* X16 = reinterpret_cast<const unsigned short*>(X);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wave * 4 + 2 * i + lh, gr = min(r0 + row, rend - 1);
                __builtin_amdgcn_global_load_lds(X16 + (size_t)gr * ldx + li * 8, (lds_ptr_t)(st + (wave * 4 + 2 * i) * 512), 16, 0, 0);
            }
        }
        if (has_d) {
            const int id = wave * 64 + lane;
            if (id < dchunks) {
                const int e = id * 4, r = e / ldd, c = e - r * ldd;
                const int gr = min(r0 + r, rend - 1);
                __builtin_amdgcn_global_load_lds(dY + (size_t)gr * ldd + c, (lds_ptr_t)(st + XBYTES + wave * 1024), 16, 0, 0);
            }
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float bsum = 0.f;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    const unsigned xoff = (unsigned)(lh * 256 + 32 * wave + li) * (XB ? 2u : 4u);      // element (row lh, column 32 wave + li) of the X tile
    const unsigned doff = (unsigned)(XBYTES + (lh * ldd + min(li, ldd - 1)) * 4);       // element (row lh, class li) of the dY tile
    const bool cls = li < ldd;                                                           // lanes past the padded class count feed zeros

    for (int t = 0; t < NS_DEPTH && t < ntiles; ++t) dma(t);
    for (int t = 0; t < ntiles; ++t) {
        ns_wait_vm(per_tile * (min(t + NS_DEPTH - 1, ntiles - 1) - t));                  // tile t landed; younger tiles stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NS_DEPTH < ntiles) dma(t + NS_DEPTH);
        const unsigned sb = lds0 + (unsigned)((t % NS_STAGES) * STAGE);
        const int valid = rend - (rbeg + t * NS_ROWS);
        if (valid < NS_ROWS) {                                                           // last tile of the range: rows past the end contribute nothing
            float* dt = reinterpret_cast<float*>(lds + (t % NS_STAGES) * STAGE + XBYTES);
            for (int e = valid * ldd + tid; e < NS_ROWS * ldd; e += 512) dt[e] = 0.f;
            __syncthreads();
        }
        if (32 * wave >= nx) continue;                                                   // (wave-uniform; the barrier above is the tile's only one)
        float a[16], b[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {                                                   // rows 2 s + lh
            asm volatile("ds_read_b32 %0, %1" : "=v"(a[s]) : "v"(sb + doff + (unsigned)(2 * s * ldd * 4)) : "memory");
            if (!XB) asm volatile("ds_read_b32 %0, %1" : "=v"(b[s]) : "v"(sb + xoff + (unsigned)(2 * s * 1024)) : "memory");
            else asm volatile("ds_read_u16 %0, %1" : "=v"(b[s]) : "v"(sb + xoff + (unsigned)(2 * s * 512)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                     "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]) : : "memory");
        asm volatile("" : "+v"(a[15]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                     "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]) : : "memory");
        asm volatile("" : "+v"(b[14]), "+v"(b[15]) : : "memory");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float av = li == ones_class ? ((2 * s + lh < valid) ? 1.f : 0.f) : (cls ? a[s] : 0.f);    // rows past the end count for nothing
            const float bv = XB ? __uint_as_float(__float_as_uint(b[s]) << 16) : b[s];
            if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
            bsum += av;
        }
    }
    // lane (li, lh) holds gW rows c = 8 q + 4 lh + e, column 32 wave + li
    gW = grad_target(gW); gb = grad_target(gb); ones_row = grad_target(ones_row);       // (this XCD's shard when a pass has them on)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = 8 * (r >> 2) + 4 * lh + (r & 3);
        const int j = 32 * wave + li;
        if (j >= nx) continue;
        if (c == ones_class) unsafeAtomicAdd(ones_row + j, acc0[r] + acc1[r]);
        else if (c < no) unsafeAtomicAdd(gW + (size_t)c * sc + (size_t)j * sn, acc0[r] + acc1[r]);
    }
    if (gb && wave == 0) {       // every wave read the same dY; wave 0 folds the two row parities and adds the bias gradient
        const unsigned u = __float_as_uint(bsum);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0 && li < no) unsafeAtomicAdd(gb + li, tot);
    }
}

// Eligibility decided by the caller (gemm.hip): ni = 256 (fp32 X: any multiple of 4 up to 256), no <= ldd <= 32, ldd % 4 == 0, M >= 4096,
// 16-byte-aligned rows.
int clift_wgrad_narrow_stream_launch(const float* dY, int ldd, int no, const float* X, int ldx, int ni, int M, float* gW, int ldw, float* gb, int x_bf16,
                                     hipStream_t st) {
    const int tiles = cdiv(M, NS_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), NS_ROWS) * NS_ROWS;
    if (x_bf16) k_wgrad_narrow_stream<true><<<cdiv(M, rpb), 512, 0, st>>>(dY, ldd, no, X, ldx, M, rpb, gW, (long)ldw, 1L, gb, -1, nullptr, 256);
    else k_wgrad_narrow_stream<false><<<cdiv(M, rpb), 512, 0, st>>>(dY, ldd, no, X, ldx, M, rpb, gW, (long)ldw, 1L, gb, -1, nullptr, ni);
    return clift_check_launch("clift_wgrad_narrow(stream)");
}

// K = 3 first-layer backward through the same kernel: dW[n][0..2] += sum_m dH[m][n] x[m][0..2], db[n] += sum_m dH[m][n]
// (x4 = (M, 4) positions, dH = (M, 256) hidden gradient, fp32 or bf16-stored).  Eligibility decided by the caller.
int clift_k3_bwd_stream_launch(const float* x4, const float* dH, int ldh, int M, float* dW, int ldw, float* db, int dh_bf16, hipStream_t st) {
    const int tiles = cdiv(M, NS_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), NS_ROWS) * NS_ROWS;
    if (dh_bf16) k_wgrad_narrow_stream<true><<<cdiv(M, rpb), 512, 0, st>>>(x4, 4, 3, dH, ldh, M, rpb, dW, 1L, (long)ldw, nullptr, 3, db, 256);
    else k_wgrad_narrow_stream<false><<<cdiv(M, rpb), 512, 0, st>>>(x4, 4, 3, dH, ldh, M, rpb, dW, 1L, (long)ldw, nullptr, 3, db, 256);
    return clift_check_launch("clift_linear_k3_bwd(stream)");
}

// ============================================================================ dgrad of the narrow output layers
// C[m][n] = mask[m][n] > 0 ? sum_k dOut[m][k] W[k][n] : 0   for K <= 32 (22 classes / 3 instance dims), N = 256: the first step of
// the hidden-layer backward.  Writes M x 256 and reads the M x 256 ReLU mask: an HBM stream with a few MFMAs per tile.  Same
// skeleton as k_layer_f32<true> (layer_f32.hip): persistent blocks, the weight slice of a wave in registers (KJ float4 per lane),
// 32-row tiles of dOut by LDS-DMA (a linear copy: the tile is contiguous in memory), the mask of a tile prefetched into registers
// at tile start, swapped MFMA operands so a lane owns one output row and stores 16 bytes (8 when bf16-stored) at a time.
// HB = the mask and the output are bf16-stored (bf16 mode); the products are fp32 either way.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

// MASK = false (fp32 only): plain dX = dOut W with N = g.N <= 256 output columns (a multiple of 4): the dgrad of the appearance basis
// (27 -> 144, tensoRF.py:65,127-134), which the tiled kernel ran at 1.4 TB/s.  Waves whose 32 columns lie past N only help with the DMA.
// WG (fp32, masked, N = 256 or a smaller multiple of 32 -- the 128-wide appearance head; waves past N only help with the DMA): the output layer's WEIGHT gradient in the same pass -- gW[c][n] += sum_m dOut[m][c] h[m][n], gb[c] += sum_m dOut[m][c] --
// where h IS the mask tensor (the post-ReLU activation): the separate k_wgrad_narrow_stream launch streams the M x 256 activation from
// memory a second time.  A lane holds its row's 16 mask values; the wave turns its 32 x 32 block through a private padded LDS buffer
// (4 ds_write_b128, 16 ds_read_b32 per lane) so that rows become the MFMA reduction index, reads dOut column-wise from the tile that is
// already in LDS, and adds 16 MFMAs per tile into one accumulator kept for the whole row range (flushed once, like k_wgrad_narrow_stream).
constexpr int DN_TPITCH = 36;            // floats per row of the transposition buffer (16-byte rows, conflict-free both ways)
template <int KJ, bool HB, bool MASK, bool WG = false, bool PF = WG>
__global__ __launch_bounds__(512, 2) void k_dgrad_narrow_stream(GemmP g, int rows_per_block, float* __restrict__ gW = nullptr, int ldgw = 0,
                                                                float* __restrict__ gb = nullptr, int no = 0) {
    static_assert(!WG || MASK, "the fused weight gradient exists for the masked forms");
    constexpr int ROWS = 32, TILEB = ROWS * 32 * 4;                          // stage: up to 32 rows x 32 floats
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILEB + (WG ? 8 * ROWS * DN_TPITCH * 4 : 0)];   // the only LDS object
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    if (rows_limited()) {
        g.M = limit_rows(g.M);
        rows_per_block = ((g.M + (int)gridDim.x - 1) / (int)gridDim.x + ROWS - 1) / ROWS * ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + ROWS - 1) / ROWS;
    const int lda = g.lda, K = g.K;

    float4 w[KJ];             // w[j] = W[k = 8 j + 4 lh + 0..3][n = 32 wave + li], zero past K
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * j + 4 * lh + i;
            v[i] = (k < K && 32 * wave + li < g.N) ? g.B[(size_t)k * g.ldb + 32 * wave + li] : 0.f;
        }
        w[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int dchunks = ROWS * lda / 4;                                      // 16-byte chunks of a tile (<= 256)
    auto dma = [&](int t) {
        const int id = wave * 64 + lane;
        if (wave * 64 < dchunks && id < dchunks) {
            const int e = id * 4, r = e / lda, c = e - r * lda;
            const int gr = min(rbeg + t * ROWS + r, rend - 1);
            __builtin_amdgcn_global_load_lds(g.A + (size_t)gr * lda + c, (lds_ptr_t)(lds + (t & 1) * TILEB + wave * 1024), 16, 0, 0);
        }
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    unsigned foff[KJ];        // fragment (row li, k = 8 j + 4 lh .. +3); a fragment past the row (its weights are zero) re-reads k = 0..3
#pragma unroll
    for (int j = 0; j < KJ; ++j) foff[j] = (unsigned)((li * lda + ((8 * j + 4 * lh + 3 < lda) ? 8 * j + 4 * lh : 0)) * 4);
    const bool full_wave = 32 * wave + 32 <= g.N;                            // all four column groups of this wave are stored
    f32x16 accw;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.f;
    float bsum = 0.f;
    const unsigned tb0 = lds0 + (unsigned)(2 * TILEB + wave * ROWS * DN_TPITCH * 4);      // this wave's transposition buffer
    // WG form: the mask of a tile is fetched ONE TILE AHEAD (into `nk` / `nh`, copied to `mk` / `mh` at the top of its tile): fetched at the start of
    // its own tile and awaited before the epilogue, the kernel kept only 32 KB of reads in flight per CU and a tile lasted one HBM round trip
    // (140.6 -> 110.3 us at 249 k rows x 22 classes).  Always issued -- for the tile past the last one on clamped rows -- so that no run-time
    // branch separates the asm loads from the asm wait that publishes their registers.  The plain masked form keeps the same-tile fetch: its
    // tiles are a handful of MFMAs long, one tile of distance hides nothing and moves the wait in front of the barrier (102 -> 116 us).
    f32x4 nk[4];
    u32x2v nh[4];
    // (masked form: N is a multiple of 32; a wave past N only helps with the DMA.  Its mask loads stay in the instruction stream --
    // pointed at columns that exist -- for the same reason)
    const bool wave_on = 32 * wave < g.N;
    const int mcol = wave_on ? 32 * wave : 0;
    auto mask_fetch = [&](int t) {
        const int mrow = min(rbeg + t * ROWS + li, rend - 1);
        if (!MASK) {
        } else if (!HB) {
            const float* mp = g.mask + (size_t)mrow * g.ldmask + mcol + 4 * lh;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                         "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                         : "=&v"(nk[0]), "=&v"(nk[1]), "=&v"(nk[2]), "=&v"(nk[3]) : "v"(mp) : "memory");
        } else {
            const unsigned short* mp = reinterpret_cast<const unsigned short*>(g.mask) + (size_t)mrow * g.ldmask + mcol + 4 * lh;
            asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:16\n\t"
                         "global_load_dwordx2 %2, %4, off offset:32\n\tglobal_load_dwordx2 %3, %4, off offset:48"
                         : "=&v"(nh[0]), "=&v"(nh[1]), "=&v"(nh[2]), "=&v"(nh[3]) : "v"(mp) : "memory");
        }
    };
    dma(0);
    if (PF) mask_fetch(0);
    for (int t = 0; t < ntiles; ++t) {
        // Issue order of a wave per tile: DMA(t+1), mask(t+1) [4], ..., stores(t) [4].  At the top of tile t everything older than the four
        // stores of tile t-1 must have landed -- the DMA of this tile and (MASK) this tile's mask, both issued before them; a wave that does not
        // store all four column groups drains
        if (t > 0 && ((MASK && 32 * wave < g.N) || (!MASK && full_wave))) {
            if (!MASK) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (!HB) asm volatile("s_waitcnt vmcnt(4)" : "+v"(nk[0]), "+v"(nk[1]), "+v"(nk[2]), "+v"(nk[3]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(4)" : "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(nh[3]) : : "memory");
        } else {
            if (!MASK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (!HB) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nk[0]), "+v"(nk[1]), "+v"(nk[2]), "+v"(nk[3]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(nh[3]) : : "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < ntiles) dma(t + 1);
        f32x4 mk[4];
        u32x2v mh[4];
        if (PF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { mk[q] = nk[q]; mh[q] = nh[q]; }
            mask_fetch(t + 1);
        } else {
            mask_fetch(t);
        }
        if (WG) {
            const int valid = rend - (rbeg + t * ROWS);
            if (valid < ROWS) {                                              // last tile of the range: rows past its end add nothing to gW
                float* dt = reinterpret_cast<float*>(lds + (t & 1) * TILEB);
                for (int e = valid * lda + tid; e < ROWS * lda; e += 512) dt[e] = 0.f;
                __syncthreads();
            }
        }
        const int m = rbeg + t * ROWS + li;
        f32x4 fa[KJ];
        const unsigned sb = lds0 + (unsigned)((t & 1) * TILEB);
#pragma unroll
        for (int j = 0; j < KJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[j]) : "v"(sb + foff[j]) : "memory");
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[j]) : : "memory");
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].x, fa[j].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].y, fa[j].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].z, fa[j].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].w, fa[j].w, acc1, 0, 0, 0);
        }
        if (!PF && MASK) {                   // same-tile fetch: the mask is needed from here on
            if (!HB) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nk[0]), "+v"(nk[1]), "+v"(nk[2]), "+v"(nk[3]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(nh[3]) : : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) { mk[q] = nk[q]; mh[q] = nh[q]; }
        }
        if (WG && wave_on) {
            // the wave's 32 x 32 block of h: row-per-lane registers -> LDS [row][36] -> column-per-lane registers (rows 2 s + lh)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 hv = mk[q];
                if (HB) {                    // bf16-stored h: widened in the register (exact), so the products are fp32 as in k_wgrad_narrow_stream<true>
                    const unsigned m0 = mh[q][0], m1 = mh[q][1];
                    hv[0] = __uint_as_float(m0 << 16); hv[1] = __uint_as_float(m0 & 0xffff0000u);
                    hv[2] = __uint_as_float(m1 << 16); hv[3] = __uint_as_float(m1 & 0xffff0000u);
                }
                asm volatile("ds_write_b128 %0, %1" : : "v"(tb0 + (unsigned)((li * DN_TPITCH + 8 * q + 4 * lh) * 4)), "v"(hv) : "memory");
            }
            float hb[16], da[16];
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                asm volatile("ds_read_b32 %0, %1" : "=v"(hb[s_]) : "v"(tb0 + (unsigned)(((2 * s_ + lh) * DN_TPITCH + li) * 4)) : "memory");
                asm volatile("ds_read_b32 %0, %1" : "=v"(da[s_]) : "v"(sb + (unsigned)(((2 * s_ + lh) * lda + min(li, lda - 1)) * 4)) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hb[0]), "+v"(hb[1]), "+v"(hb[2]), "+v"(hb[3]), "+v"(hb[4]), "+v"(hb[5]), "+v"(hb[6]), "+v"(hb[7]),
                         "+v"(hb[8]), "+v"(hb[9]), "+v"(hb[10]), "+v"(hb[11]), "+v"(hb[12]), "+v"(hb[13]), "+v"(hb[14]) : : "memory");
            asm volatile("" : "+v"(hb[15]), "+v"(da[0]), "+v"(da[1]), "+v"(da[2]), "+v"(da[3]), "+v"(da[4]), "+v"(da[5]), "+v"(da[6]), "+v"(da[7]),
                         "+v"(da[8]), "+v"(da[9]), "+v"(da[10]), "+v"(da[11]), "+v"(da[12]), "+v"(da[13]) : : "memory");
            asm volatile("" : "+v"(da[14]), "+v"(da[15]) : : "memory");
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                const float av = li < lda ? da[s_] : 0.f;                    // lanes past the padded class count feed zeros
                accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hb[s_], accw, 0, 0, 0);
                bsum += av;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[4] = {acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1], acc0[4 * q + 2] + acc1[4 * q + 2],
                          acc0[4 * q + 3] + acc1[4 * q + 3]};
            if (!MASK) {
                if (m < rend && 32 * wave + 8 * q + 4 * lh + 3 < g.N)
                    *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + 32 * wave + 8 * q + 4 * lh) = make_float4(o[0], o[1], o[2], o[3]);
            } else if (!wave_on) {
            } else if (!HB) {
                o[0] = mk[q].x > 0.f ? o[0] : 0.f; o[1] = mk[q].y > 0.f ? o[1] : 0.f;
                o[2] = mk[q].z > 0.f ? o[2] : 0.f; o[3] = mk[q].w > 0.f ? o[3] : 0.f;
                if (m < rend) *reinterpret_cast<float4*>(g.C + (size_t)m * g.ldc + 32 * wave + 8 * q + 4 * lh) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                const unsigned m0 = mh[q][0], m1 = mh[q][1];
                if (!bf16_bits_positive((unsigned short)(m0 & 0xffffu))) o[0] = 0.f;
                if (!bf16_bits_positive((unsigned short)(m0 >> 16))) o[1] = 0.f;
                if (!bf16_bits_positive((unsigned short)(m1 & 0xffffu))) o[2] = 0.f;
                if (!bf16_bits_positive((unsigned short)(m1 >> 16))) o[3] = 0.f;
                const unsigned lo = (unsigned)float_to_bf16_bits(o[0]) | ((unsigned)float_to_bf16_bits(o[1]) << 16);
                const unsigned hi = (unsigned)float_to_bf16_bits(o[2]) | ((unsigned)float_to_bf16_bits(o[3]) << 16);
                if (m < rend) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + (size_t)m * g.ldc + 32 * wave + 8 * q + 4 * lh) = make_uint2(lo, hi);
            }
        }
    }
    if (PF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the mask prefetch of the tile past the end: a wave must not end with loads in flight
    if (WG) {
        // lane (li, lh) holds gW rows c = 8 q + 4 lh + e, column 32 wave + li
        gW = grad_target(gW); gb = grad_target(gb);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 8 * (r >> 2) + 4 * lh + (r & 3);
            if (c < no && wave_on) unsafeAtomicAdd(gW + (size_t)c * ldgw + 32 * wave + li, accw[r]);
        }
        if (gb && wave == 0) {       // every wave read the same dOut; wave 0 folds the two row parities and adds the bias gradient
            const unsigned u = __float_as_uint(bsum);
            const u32x2v sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (lh == 0 && li < no) unsafeAtomicAdd(gb + li, tot);
        }
    }
}

// Backward of a narrow output layer over a 256-wide ReLU hidden layer in ONE pass over the hidden activation H (tensoRF.py:480-481,
// 593-594 backward): dX = (H > 0) . (dOut W), gW += dOut^T H, gb += column sums of dOut.  dOut (M, ldd) fp32 with zero pad columns,
// W (no, 256) pitch ldw, H (M, 256) pitch ldh, dX (M, 256) pitch ldx; no <= ldd <= 32, ldd % 4 == 0, M >= 1, 16-byte-aligned rows.
extern "C" int clift_out_layer_bwd(const float* dOut, int ldd, int no, const float* W, int ldw, const float* H, int ldh, int M,
                                   float* dX, int ldx, float* gW, int ldgw, float* gb, clift_stream_t s) {
    return clift_out_layer_bwd_nh(dOut, ldd, no, W, ldw, H, ldh, 256, M, dX, ldx, gW, ldgw, gb, 0, s);
}

// The same over a hidden layer of nh <= 256 units, nh % 32 == 0 (ABI 16: the 128-wide appearance head, tensoRF.py:393-397 backward), and --
// h_bf16 -- over a bf16-STORED hidden activation with a bf16-stored input gradient (bf16 mode; pitches in elements; nh = 256).
extern "C" int clift_out_layer_bwd_nh(const float* dOut, int ldd, int no, const float* W, int ldw, const float* H, int ldh, int nh, int M,
                                      float* dX, int ldx, float* gW, int ldgw, float* gb, int h_bf16, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(no >= 1 && no <= ldd && ldd <= 32 && ldd % 4 == 0, "clift_out_layer_bwd: need no <= ldd <= 32, ldd %% 4 == 0 (got no=%d ldd=%d)", no, ldd);
    CLIFT_REQUIRE(nh >= 32 && nh <= 256 && nh % 32 == 0, "clift_out_layer_bwd: hidden width must be a multiple of 32 in [32, 256] (got %d)", nh);
    CLIFT_REQUIRE(!h_bf16 || nh == 256, "clift_out_layer_bwd: the bf16-stored form takes a 256-wide hidden layer (got %d)", nh);
    CLIFT_REQUIRE(ldh % 4 == 0 && ldx % 4 == 0 && ldh >= nh && ldx >= nh && ldw >= nh && ldgw >= nh && (((uintptr_t)dOut) & 15) == 0 &&
                  (((uintptr_t)H) & 15) == 0 && (((uintptr_t)dX) & 15) == 0, "clift_out_layer_bwd: 16-byte aligned rows with pitches >= the hidden width required");
    GemmP p = {};
    p.M = M; p.N = nh; p.K = no; p.A = dOut; p.lda = ldd; p.B = W; p.ldb = ldw; p.C = dX; p.ldc = ldx; p.mask = H; p.ldmask = ldh;
    const int tiles = cdiv(M, 32);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), 32) * 32;
    const dim3 grid(cdiv(M, rpb));
    hipStream_t st = as_stream(s);
    const int kj = cdiv(no, 8);
    if (h_bf16) {
        if (kj <= 1) k_dgrad_narrow_stream<1, true, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
        else if (kj == 2) k_dgrad_narrow_stream<2, true, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
        else if (kj == 3) k_dgrad_narrow_stream<3, true, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
        else k_dgrad_narrow_stream<4, true, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
    } else if (kj <= 1) k_dgrad_narrow_stream<1, false, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
    else if (kj == 2) k_dgrad_narrow_stream<2, false, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
    else if (kj == 3) k_dgrad_narrow_stream<3, false, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
    else k_dgrad_narrow_stream<4, false, true, true><<<grid, 512, 0, st>>>(p, rpb, gW, ldgw, gb, no);
    return clift_check_launch("clift_out_layer_bwd");
}

// Eligibility decided by the callers (gemm.hip / gemm_bf16.hip): plain fp32 A with lda in {4, 8, .., 32} = its row pitch, K <= lda, b_trans
// weights, no bias / activation, M >= 4096, and either N = 256 with a mask (half = bf16-stored mask and output) or no mask, fp32, N <= 256,
// N % 4 == 0.
int clift_dgrad_narrow_stream_launch(const GemmP& p, int half, hipStream_t st) {
    const int tiles = cdiv(p.M, 32);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(p.M, blocks), 32) * 32;
    const dim3 grid(cdiv(p.M, rpb));
    const int kj = cdiv(p.K, 8);
#define CLIFT_DN(KJ)                                                                                     \
    do {                                                                                                 \
        if (!p.mask) k_dgrad_narrow_stream<KJ, false, false><<<grid, 512, 0, st>>>(p, rpb);              \
        else if (half) k_dgrad_narrow_stream<KJ, true, true><<<grid, 512, 0, st>>>(p, rpb);              \
        else k_dgrad_narrow_stream<KJ, false, true><<<grid, 512, 0, st>>>(p, rpb);                       \
    } while (0)
    if (kj <= 1) CLIFT_DN(1);
    else if (kj == 2) CLIFT_DN(2);
    else if (kj == 3) CLIFT_DN(3);
    else CLIFT_DN(4);
#undef CLIFT_DN
    return clift_check_launch("clift_gemm(narrow dgrad stream)");
}

// ============================================================================ forward of a narrow output layer (+ row softmax)
// out[m][0..no) = act( H[m][0..256) W^T + b ),  no <= 32 (22 semantic classes), act 0 = none, 2 = softmax over the row (tensoRF.py:591-594,37):
// ONE read of the 256-wide hidden activation -- an HBM stream -- instead of the tiled 256 x 32 GEMM (which re-streams it at < half the
// stream rate) followed by a row-activation launch over the logits.  Persistent blocks, 32-row tiles by LDS-DMA three tiles ahead (source-side
// bank swizzle as in layer_f32.hip: 16-byte chunk c of row r sits in slot c ^ (r & 15)).
// Round 6: PRODUCER waves 0 - 3 and CONSUMER waves 4 - 7 (waves w and w + 4 share a SIMD).  Producer w issues the DMA of 8 rows per tile and
// contracts k = 64 w .. +63 for all 32 rows x 32 (padded) classes: weights first, so a lane owns one row -- 8 ds_read_b128 + 32
// v_mfma_f32_32x32x2_f32 per tile, its 32 weights in registers -- and parks its slice (4 KB) in one of TWO areas.  The consumers reduce the tile
// before: 8 threads per row add the four slices in slice order + bias, fold max / sum of the softmax over their 4 classes and with xor shuffles
// over the 8 lanes, and store 4 classes each.  One barrier per tile.  Before (two barriers, every wave contracting a 32-wide slice and then
// reducing, every wave in the same phase): 7100 cycles per tile for a 32 KB stream that needs ~4400; by ablation the MFMAs, the slice reads +
// softmax, and the wait for the stores' acknowledgement (vmcnt is one in-order counter) each cost their full time, 20 - 28 % of the kernel
// apiece (docs/history/round6.md).  Now the matrix pipe of a SIMD belongs to one wave (2048 cycles per tile), the reduce runs beside it on the
// vector pipe, stores are the consumers' only vector-memory operations and nothing waits for them, and four stages keep 96 KB in flight.
constexpr int OF_ROWS = 32, OF_STAGE = OF_ROWS * 1024, OF_STAGES = 4, OF_PART = OF_STAGES * OF_STAGE, OF_AREA = 4 * 4096;      // bytes

__global__ __launch_bounds__(512, 2) void k_out_narrow_fwd(const float* __restrict__ H, int ldh, const float* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, int no, int M, int rows_per_block, float* __restrict__ out,
                                                           int ldo, int act) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[OF_PART + 2 * OF_AREA];         // 160 KB, the only LDS object
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    if (rows_limited()) {
        M = limit_rows(M);
        rows_per_block = ((M + (int)gridDim.x - 1) / (int)gridDim.x + OF_ROWS - 1) / OF_ROWS * OF_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + OF_ROWS - 1) / OF_ROWS;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    if (wave < 4) {
        // ---------------------------------------------------------------- producer
        auto dma = [&](int t) {                         // rows 8 wave .. +7 of tile t, one 1 KB row per instruction
            unsigned char* st = lds + (t % OF_STAGES) * OF_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wave * 8 + i, gr = min(rbeg + t * OF_ROWS + row, rend - 1);
                __builtin_amdgcn_global_load_lds(H + (size_t)gr * ldh + ((lane ^ (row & 15)) << 2), (lds_ptr_t)(st + row * 1024), 16, 0, 0);
            }
        };
        // weights of class li, k = 64 wave + 8 j + 4 lh + i (classes past `no` read as zero)
        f32x4 wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            wv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (li < no) wv[j] = *reinterpret_cast<const f32x4*>(W + (size_t)li * ldw + 64 * wave + 8 * j + 4 * lh);
        }
        const unsigned fbase = lds0 + (unsigned)(li * 1024);
        // the slice is parked with row li of class quad g = 2 q + lh in slot (li + 2 g) & 31 of the quad's 512 bytes: a consumer wave (8 rows x 8
        // quads) then reads 16 different 16-byte slots in every 16 lanes
        unsigned pw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pw[q] = lds0 + (unsigned)(OF_PART + wave * 4096 + q * 1024 + lh * 512 + ((li + 2 * (2 * q + lh)) & 31) * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the weight loads: from here on this wave's only vector-memory operations are the DMAs
        for (int t = 0; t < OF_STAGES - 1 && t < ntiles; ++t) dma(t);
        for (int t = 0; t < ntiles; ++t) {
            // tile t has landed when at most the younger tiles' DMAs (8 per tile, at most two tiles) are outstanding
            const int younger = min(t + OF_STAGES - 2, ntiles - 1) - t;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // tile t visible to every producer; tile t - 1's stage is free; the consumers are done with area t & 1
            asm volatile("" ::: "memory");
            if (t + OF_STAGES - 1 < ntiles) dma(t + OF_STAGES - 1);      // into tile t - 1's stage
            const unsigned sb = fbase + (unsigned)((t % OF_STAGES) * OF_STAGE);
            f32x4 xb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned a = sb + (unsigned)((((2 * (8 * wave + j) + lh) ^ (li & 15))) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(xb[j]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xb[4]), "+v"(xb[5]), "+v"(xb[6]), "+v"(xb[7]) : : "memory");
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][i], xb[j][i], acc, 0, 0, 0);
            // lane (li = row, lh) holds classes 8 q + 4 lh + e in acc[4 q + e]: park the slice.  The writes are inline asm, which the compiler's hazard
            // recogniser does not look into: an LDS instruction that reads the destination of a 16-pass MFMA needs 18 wait states the hardware does
            // NOT interlock (without them the slice leaves before the last MFMAs have landed -- seen as a few per cent error in every output)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc) : : "memory");
            const unsigned ao = (unsigned)((t & 1) * OF_AREA);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                asm volatile("ds_write_b128 %0, %1" : : "v"(pw[q] + ao), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                   // the last tile's slices are parked
        return;
    }

    // -------------------------------------------------------------------- consumer: thread (row rm, class quad cg = classes 4 cg .. +3)
    const int ct = tid - 256, rm = ct >> 3, cg = ct & 7, c0 = 4 * cg;
    float bq[4];
    bool on[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { on[e] = c0 + e < no; bq[e] = on[e] ? bias[c0 + e] : 0.f; }
    const unsigned pread = lds0 + (unsigned)(OF_PART + (cg >> 1) * 1024 + (cg & 1) * 512 + ((rm + 2 * cg) & 31) * 16);
    // The outputs leave as BUFFER stores through a descriptor cut to this block's rows: a lane without an output (class >= no, row past the end)
    // gets an offset past the range and the hardware drops it -- no branches around the stores
    const int nrows = __builtin_amdgcn_readfirstlane(rend - rbeg);
    const unsigned long long oa = (unsigned long long)(uintptr_t)(out + (size_t)rbeg * ldo);
    const unsigned oa_lo = __builtin_amdgcn_readfirstlane((unsigned)oa), oa_hi = __builtin_amdgcn_readfirstlane((unsigned)(oa >> 32));
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>((uintptr_t)(((unsigned long long)oa_hi << 32) | oa_lo)), 0,
        (int)__builtin_amdgcn_readfirstlane((unsigned)((((size_t)(nrows - 1)) * ldo + no) * 4)), 0x00020000);
    auto reduce = [&](int t) {
        const unsigned pr = pread + (unsigned)((t & 1) * OF_AREA);
        f32x4 s[4];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) asm volatile("ds_read_b128 %0, %1" : "=v"(s[w4]) : "v"(pr + (unsigned)(w4 * 4096)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : : "memory");
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ((s[0][e] + s[1][e]) + s[2][e]) + s[3][e] + bq[e];
        if (act == 2) {
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, on[e] ? x[e] : -INFINITY);
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = on[e] ? expf(x[e] - mx) : 0.f; sum += x[e]; }
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) sum += __shfl_xor(sum, d);
            const float inv = 1.f / sum;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] *= inv;
        }
        const int lr = t * OF_ROWS + rm;                // row within the block's range
        const int off = (lr * ldo + c0) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[e]), rs_o, (lr < nrows && on[e]) ? off + 4 * e : -1, 0, 0);
    };
    for (int t = 0; t < ntiles; ++t) {
        __builtin_amdgcn_s_barrier();                   // tile t - 1's slices are parked (area (t - 1) & 1)
        asm volatile("" ::: "memory");
        if (t > 0) reduce(t - 1);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    reduce(ntiles - 1);
}

extern "C" int clift_out_layer_fwd(const float* H, int ldh, const float* W, int ldw, const float* b, int no, int M, float* out, int ldo, int act,
                                   clift_stream_t s) {
    CLIFT_REQUIRE(no >= 1 && no <= 32, "clift_out_layer_fwd: out_features must be in [1,32] (got %d)", no);
    CLIFT_REQUIRE(act == 0 || act == 2, "clift_out_layer_fwd: act must be 0 (none) or 2 (softmax over the row)");
    CLIFT_REQUIRE((((uintptr_t)H) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && ldh % 4 == 0 && ldh >= 256 && ldw % 4 == 0 && ldw >= 256 && ldo >= no && b != nullptr,
                  "clift_out_layer_fwd: H / W must be 16-byte aligned with pitches >= 256 that are multiples of 4, ldo >= out_features, bias required");
    if (M <= 0) return 0;
    const int tiles = cdiv(M, OF_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), OF_ROWS) * OF_ROWS;
    k_out_narrow_fwd<<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(H, ldh, W, ldw, b, no, M, rpb, out, ldo, act);
    return clift_check_launch("clift_out_layer_fwd");
}
