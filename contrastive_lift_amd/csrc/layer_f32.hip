// layer_f32.hip -- the exact-fp32 256 -> 256 hidden layers (forward: C = relu(A W^T + b); dgrad: C = mask . (A W)) as PERSISTENT
// kernels.  The tiled kernel (gemm.hip) spends ~40 % of a K = 256 launch outside its k-loop: per 128-row tile a prologue (first
// global round trip), 16 barriers and an output burst that cannot overlap the next tile because the staging registers and the
// accumulators fill the 128-VGPR budget of a 4-waves-per-SIMD kernel; and a launch is a whole number of 512-tile chip rounds.
// Here one 8-wave block per CU owns a contiguous range of rows (balanced to 32 rows: no tail round) and
//   * keeps the whole weight matrix in REGISTERS: wave w owns output columns 32 w .. +31 for all 256 k = 128 VGPRs (the block
//     runs at 2 waves per SIMD, 256-VGPR budget), loaded once per block;
//   * streams the activation rows through two 32 KB LDS stages by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
//     ds_write pass), one 32-row tile ahead; ONE barrier per tile (8 k MFMA cycles) instead of one per 16-k step;
//   * leaves the 16-byte output stores of tile t in flight under the MFMAs of tile t+1 (counted s_waitcnt vmcnt).
// MFMA: v_mfma_f32_32x32x2_f32 with swapped operands (weights first), so a lane owns one output row and stores 4 consecutive
// columns per instruction.  k order inside an 8-k group: lane half lh uses k = 8 j + 4 lh + i at step (j, i), which makes the
// A fragment of four consecutive steps one ds_read_b128; the LDS image is lane-linear (DMA), bank swizzle at the source:
// 16-byte chunk c of row r sits in slot c ^ (r & 15).  The contraction runs as two accumulator chains (steps i = 0, 2 and 1, 3)
// summed in the epilogue, so that the MFMA sequence of a wave is not one serial dependency chain.
// Forward: the bias enters as one more contraction step (first operand bias[n] at k-half 0, second operand 1.0).
// Dgrad: the ReLU mask of a tile (the layer's fp32 input activation, as large as the tile) does not fit in LDS next to the rows, so
// each lane fetches the 4 x 16 bytes it needs straight into registers at the START of the tile and waits for them at the epilogue,
// ~8 k MFMA cycles later.
// Three fused forms share the skeleton (template flags): GEN (forward: the K = 3 input layer generated in-kernel), OUTV (forward: a narrow
// output layer applied to the tile in registers) and K3W (dgrad: the result consumed in-kernel by the K = 3 layer's weight gradient, mask
// re-derived from the positions) -- described at their parameter structs below.
// All LDS reads and the mask loads are inline asm: beside an in-flight LDS-DMA the compiler guards ordinary reads of the array /
// ordinary global loads with s_waitcnt vmcnt(0), which would drain the prefetched tile and the output stores; each asm wait is
// tied to the registers it guards as a data dependency, and sched_barriers pin the read-ahead order.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_f32)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LF_ROWS = 32;                  // rows per streamed tile
constexpr int LF_TILE = LF_ROWS * 64;        // float4 per tile: 32 rows x 1 KB

template <int N>
static __device__ __forceinline__ void wait_vmf() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// GEN (forward only): the layer's input is not read from memory but GENERATED -- it is the first layer of an xyz head,
// A[m][c] = relu(W0[c] . x_m + b0[c]) with K = 3 (tensoRF.py:475,576), i.e. 13 VALU instructions per 16 bytes where the streamed
// form costs a 255 MB write (k_linear_k3_fwd) plus its re-read.  A wave writes the four rows it would have DMA'd with ds_write_b128
// into the same swizzled slots; a lane owns columns 4 lane .. +3 of every row, so its 12 weights + 4 biases stay in registers; the
// sample positions of the block's rows are staged in LDS (2048 rows = 32 KB at a time).  h1 (nullable) receives the generated
// activation when the head has a backward (the next layer's weight gradient and the ReLU mask need it).
struct GenP {
    const float* x4;      // (M, 4) normalised sample positions
    const float* W0;      // (256, 3), row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
    float* h1;            // nullable: (M, ldh1) generated first-layer activation
    int ldh1;
};
constexpr int LF_XROWS = 2048;               // positions staged per refill (64 tiles)

// OUTV (forward only): the layer is the LAST hidden layer of a head with a narrow output (E <= 4 columns: the instance heads,
// tensoRF.py:480-481), and the output layer out[m][c] = sum_k h[m][k] Wout[c][k] + bout[c] is applied to the tile while it is still in
// registers instead of by a separate launch that re-reads the 1 KB-per-row activation: every lane holds 16 values of its row, so
// 4 x 16 FMAs give its share of the E dot products (weights from LDS), a permlane swap folds the two half-waves, the eight waves'
// shares meet in LDS and 128 threads add them up (fixed order) and store -- right after the tile's MFMA loop, between two barriers (round 5: rounds
// 2 - 4 spread it through the MFMA loops of the two following tiles by hand; see the OUTV state below).  store_hidden = 0: the hidden activation
// itself is not written (no backward through the head).
struct OutP {
    const float* Wout;    // (E, 256), row pitch ldwo
    int ldwo;
    const float* bout;    // (E), nullable
    int E;
    float* out;           // (M, ldo), column offset already applied
    int ldo;
    int store_hidden;
};
constexpr int LF_OUTV_F4 = 256 + 2 * 8 * 32;  // float4: Wout rows padded to 4 x 256 floats + two buffers of 8 waves x 32 rows of partial sums

// K3W (dgrad only): the layer is the SECOND layer of an xyz head, so its input gradient dH1 = mask . (dH2 W1) has one consumer, the K = 3
// first layer's weight / bias gradient dW0[n][c] = sum_m dH1[m][n] x[m][c], db0[n] = sum_m dH1[m][n] (tensoRF.py:475,576).  Instead of
// writing dH1 (1 KB per row) for a second launch to re-read, a finished tile goes into LDS (two 32 KB images, same swizzle as the row
// stages) and is summed DOWN ITS COLUMNS during the next tile's MFMA loop: thread (n = column, half) reads 16 elements of its column
// (consecutive lanes = consecutive words: conflict-free) and the 16 positions (broadcast reads; DMA'd per tile into a ring of four 1 KB
// slots) -- 4 FMAs per element on the VALU, which idles under the MFMAs.  dH1 is never stored.
// The ReLU mask is not read either: the tile goes into LDS UNmasked, and the thread that sums column n re-derives "h1[m][n] > 0" from the
// position it is holding anyway -- W0[n] . x_m + b0[n] in the forward's operation order (3 FMAs, the same bits as the activation the
// forward generated) -- so this form of the kernel has no mask loads, no registers for them, and reads 1 KB per row instead of 2.
struct K3P {
    const float* x4;      // (M, 4) normalised sample positions
    const float* W0;      // (256, 3) first-layer weights, row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
    float* gW0;           // (256, 3) gradient, row pitch ldgw0 (accumulated into)
    int ldgw0;
    float* gb0;           // (256) (accumulated into)
};

// DGRAD = false: weights stored [n][k], bias + optional ReLU.  DGRAD = true: weights stored [k][n], fp32 ReLU mask.
template <bool DGRAD, bool GEN, bool OUTV, bool K3W = false>
__global__ __launch_bounds__(512, 2) void k_layer_f32(GemmP g, int rows_per_block, GenP gp, OutP op, K3P kp) {
    static_assert(!K3W || (DGRAD && !GEN && !OUTV), "K3W is a dgrad form");
    __shared__ __attribute__((aligned(1024))) float4 lds[2 * LF_TILE + (GEN ? LF_XROWS : 0) + (OUTV ? LF_OUTV_F4 : 0) + (K3W ? 2 * LF_TILE + 4 * 64 : 0)];   // the only LDS object
    float4* const xs = lds + 2 * LF_TILE;
    float4* const kd = lds + 2 * LF_TILE;                                    // K3W: two dH1 tile images ...
    float4* const kx = kd + 2 * LF_TILE;                                     // ... and four position slots of 64 float4
    float4* const wl4 = lds + 2 * LF_TILE + (GEN ? LF_XROWS : 0);            // OUTV: wl4[c * 64 + k / 4]
    float4* const part = wl4 + 256;                                          // OUTV: part[wave * 32 + row]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    if (rows_limited()) {          // sync-free step: the launch was sized by a capacity; re-balance the row ranges over the true row count
        g.M = limit_rows(g.M);
        rows_per_block = ((g.M + (int)gridDim.x - 1) / (int)gridDim.x + LF_ROWS - 1) / LF_ROWS * LF_ROWS;
    }
    const int rbeg = blockIdx.x * rows_per_block, rend = min(g.M, rbeg + rows_per_block);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + LF_ROWS - 1) / LF_ROWS;

    float4 w[32];             // w[j] = B(k = 8 j + 4 lh + 0..3, n = 32 wave + li)
    float bw = 0.f;
    if (!DGRAD) {
        const float* wr = g.B + (size_t)(32 * wave + li) * g.ldb + 4 * lh;
#pragma unroll
        for (int j = 0; j < 32; ++j) w[j] = *reinterpret_cast<const float4*>(wr + 8 * j);
        bw = (g.bias && lh == 0) ? g.bias[32 * wave + li] : 0.f;
    } else {
        const float* wc = g.B + (size_t)(4 * lh) * g.ldb + 32 * wave + li;
        const size_t ld = (size_t)g.ldb;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float* q = wc + (size_t)(8 * j) * ld;
            w[j] = make_float4(q[0], q[ld], q[2 * ld], q[3 * ld]);
        }
    }
    wait_vmf<0>();
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    // GEN: this lane's first-layer coefficients (columns 4 lane .. +3)
    float gw0[4], gw1[4], gw2[4], gbb[4];
    if (GEN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wr0 = gp.W0 + (size_t)(4 * lane + e) * gp.ldw0;
            gw0[e] = wr0[0]; gw1[e] = wr0[1]; gw2[e] = wr0[2]; gbb[e] = gp.b0[4 * lane + e];
        }
    }
    // positions of rows [rbeg + first, rbeg + first + LF_XROWS) -> xs (all threads; callers bracket it with barriers)
    auto fill_positions = [&](int first) {
        const int n = min(LF_XROWS, rend - rbeg - first);
        for (int e = tid; e < n; e += 512) xs[e] = *reinterpret_cast<const float4*>(gp.x4 + (size_t)(rbeg + first + e) * 4);
    };
    // One row of tile t (this wave provides rows 4 wave .. +3 of every tile): DMA'd from A, or generated.
    auto dma_row = [&](int t, int i) {
        const int row = wave * 4 + i;                                        // one 1 KB row per wave instruction
        const int gr = min(rbeg + t * LF_ROWS + row, rend - 1);              // rows past the range re-read its last row (never stored)
        const int c = lane ^ (row & 15);
        if (!GEN) {
            __builtin_amdgcn_global_load_lds(g.A + (size_t)gr * g.lda + c * 4, (lds_ptr_t)(lds + (t & 1) * LF_TILE + row * 64), 16, 0, 0);
        } else {
            const float4 x = xs[(gr - rbeg) & (LF_XROWS - 1)];                // broadcast read
            float4 o;
            o.x = fmaxf(fmaf(gw2[0], x.z, fmaf(gw1[0], x.y, fmaf(gw0[0], x.x, gbb[0]))), 0.f);      // same order as k_linear_k3_fwd
            o.y = fmaxf(fmaf(gw2[1], x.z, fmaf(gw1[1], x.y, fmaf(gw0[1], x.x, gbb[1]))), 0.f);
            o.z = fmaxf(fmaf(gw2[2], x.z, fmaf(gw1[2], x.y, fmaf(gw0[2], x.x, gbb[2]))), 0.f);
            o.w = fmaxf(fmaf(gw2[3], x.z, fmaf(gw1[3], x.y, fmaf(gw0[3], x.x, gbb[3]))), 0.f);
            lds[(t & 1) * LF_TILE + row * 64 + c] = o;
            if (gp.h1 && rbeg + t * LF_ROWS + row < rend) *reinterpret_cast<float4*>(gp.h1 + (size_t)gr * gp.ldh1 + 4 * lane) = o;
        }
    };
    // The memory instructions of a tile are SPREAD through its MFMA loop instead of bunched at the tile boundary: eight waves issuing
    // 4 stores + 4 DMA rows each at the same moment cost 1.3 us of a 8.6 us tile (the MFMA pipe idles while the waves sit in the vector-memory
    // issue queue); one instruction every other k-step disappears under the MFMAs (probe: 124 -> 145 TFLOP/s for the same work).  So the rows
    // of tile t+1 are copied at steps 0, 2, 4, 6 of tile t, and the results of tile t-1 -- kept in 16 registers -- are stored at steps 8, 10,
    // 12, 14 of tile t (dgrad: the mask loads of tile t go at steps 16 .. 22).
    float4 prev[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int prev_m = rend;                                                       // row of `prev`; rend = nothing to store yet
    if (GEN) { fill_positions(0); __syncthreads(); }
    // ---- OUTV: the narrow output layer of a finished tile, SYNCHRONOUSLY (round 5; see the same change in layer_n128.hip).  The pipelined form of
    // rounds 2 - 4 (weight fragments read one k-step ahead by hand, the waves' shares parked in a double buffer and fetched one tile later) is the
    // scheme whose frame-render form was caught returning rows with one wave's share of the wrong tile (tools/last2_soak.py on its 128-wide twin;
    // profiles/r05_determinism.txt).  Here: shares from the tile's registers right after its MFMA loop (the same FMA chain: the same bits), ONE share
    // buffer between two barriers, the next tile's barrier protects its reuse; no hand-issued LDS traffic, no cross-tile state.
    float bo_out = 0.f;
    if (OUTV) {
        float* wl = reinterpret_cast<float*>(wl4);
        for (int e = tid; e < 1024; e += 512) { const int c = e >> 8, k = e & 255; wl[e] = c < op.E ? op.Wout[(size_t)c * op.ldwo + k] : 0.f; }
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
            for (int q = 0; q < 4; ++q) {            // this lane's columns 32 wave + 8 q + 4 lh + (0..3), ascending: the FMA chain of the pipelined form
                const float4 w4 = wl4[c * 64 + 8 * wave + 2 * q + lh];
                a = fmaf(prev[q].w, w4.w, fmaf(prev[q].z, w4.z, fmaf(prev[q].y, w4.y, fmaf(prev[q].x, w4.x, a))));
            }
            const unsigned u = __float_as_uint(a);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // lower + upper half-wave
            po[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        if (lh == 0) part[wave * 32 + li] = make_float4(po[0], po[1], po[2], po[3]);
        __syncthreads();
        if (tid < 4 * LF_ROWS) {                     // 128 threads = 32 rows x 4 outputs: the eight column-waves' shares in the pipelined form's association
            const int row = tid >> 2, c = tid & 3;
            const float* pf = reinterpret_cast<const float*>(part) + row * 4 + c;
            const float v = ((pf[0] + pf[128]) + (pf[2 * 128] + pf[3 * 128])) + ((pf[4 * 128] + pf[5 * 128]) + (pf[6 * 128] + pf[7 * 128])) + bo_out;
            const int mrow = rbeg + tile * LF_ROWS + row;
            if (c < op.E && mrow < rend) op.out[(size_t)mrow * op.ldo + c] = v;
        }
    };
    // ---- K3W state: thread (kn, khalf) sums column kn of a finished dH1 tile over rows 16 khalf .. +15
    const int kn = (wave & 3) * 64 + lane, khalf = wave >> 2;
    const unsigned kd0 = (unsigned)(uintptr_t)(lds_ptr_t)kd, kx0 = (unsigned)(uintptr_t)(lds_ptr_t)kx;
    const unsigned kcol = kd0 + (unsigned)(khalf * 16 * 1024 + ((kn >> 2) << 4) + (kn & 3) * 4);    // row 16 khalf, UNswizzled slot of column kn
    const unsigned kpos = kx0 + (unsigned)(khalf * 16 * 16);
    float k3w0 = 0.f, k3w1 = 0.f, k3w2 = 0.f, k3b = 0.f;
    float kc0 = 0.f, kc1 = 0.f, kc2 = 0.f, kcb = 0.f;                        // first-layer coefficients of column kn
    if (K3W) { const float* wr0 = kp.W0 + (size_t)kn * kp.ldw0; kc0 = wr0[0]; kc1 = wr0[1]; kc2 = wr0[2]; kcb = kp.b0[kn]; }
    float kv;             // (one set: the FMAs of element r are issued before the reads of element r + 1 overwrite it)
    f32x4 kq;
    auto k3_pos_dma = [&](int t) {                    // positions of tile t -> slot t & 3 (lanes 32..63 repeat rows 0..31 into the slot's second half)
        if (wave == 0) {
            const int gr = min(rbeg + t * LF_ROWS + li, rend - 1);
            __builtin_amdgcn_global_load_lds(kp.x4 + (size_t)gr * 4, (lds_ptr_t)(kx + (t & 3) * 64), 16, 0, 0);
        }
    };
    auto k3_issue = [&](int tile, int r) {            // element (row 16 khalf + r, column kn) of tile `tile` and that row's position
        unsigned cb = kcol + (unsigned)((tile & 1) * LF_TILE * 16), pb = kpos + (unsigned)((tile & 3) * 1024);
        asm volatile("" : "+v"(cb), "+v"(pb));       // (opaque: keeps the compiler from hoisting sixteen precomputed address pairs into registers)
        const unsigned a = (cb ^ (unsigned)(r << 4)) + (unsigned)(r * 1024);
        asm volatile("ds_read_b32 %0, %1" : "=v"(kv) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(kq) : "v"(pb + (unsigned)(r * 16)) : "memory");
    };
    auto k3_fma = [&](int r) {
        asm volatile("" : "+v"(kv), "+v"(kq) : : "memory");         // (placed after the step's lgkmcnt(0): the values are there)
        const f32x4 x = kq;
        const float pre = fmaf(kc2, x.z, fmaf(kc1, x.y, fmaf(kc0, x.x, kcb)));      // same order as the forward (k_linear_k3_fwd / GEN): the sign test sees the forward's bits
        const float v = pre > 0.f ? kv : 0.f;
        k3w0 = fmaf(v, x.x, k3w0); k3w1 = fmaf(v, x.y, k3w1); k3w2 = fmaf(v, x.z, k3w2); k3b += v;
        asm volatile("" : "+v"(k3w0), "+v"(k3w1), "+v"(k3w2), "+v"(k3b));          // (pins the four operations HERE: left alone, the compiler sinks all
                                                                                // sixteen elements' arithmetic to the end of the tile and keeps 80 registers of reads alive)
    };
    if (K3W) {
        // tile 0's loop sums "tile -1": image 1 and slot 3 start as zeros
        for (int e = tid; e < LF_TILE; e += 512) kd[LF_TILE + e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 256) kx[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        k3_pos_dma(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_row(0, i);
    for (int t = 0; t < ntiles; ++t) {
        if (K3W) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's dH1 image writes of tile t-1
        if (!GEN) {
            // DMA of tile t: issued during tile t-1; younger than it: the 4 stores of tile t-2 (forward; the dgrad drained everything at the
            // end of tile t-1 for its mask)
            // (K3W: no stores, no mask loads -- the tile's DMAs are all there is; OUTV without a hidden store: the only younger instructions are the owner
            // threads' output stores, which most waves never issue)
            if (t >= 2 && !K3W && (!OUTV || op.store_hidden)) wait_vmf<4>(); else wait_vmf<0>();
            __builtin_amdgcn_s_barrier();                                    // everyone's rows have landed; everyone is done with the other stage
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (no vmcnt wait: the output stores stay in flight)
            __builtin_amdgcn_s_barrier();                                    // the generated rows of tile t are written; the other stage is free
            asm volatile("" ::: "memory");
            if (((t + 1) & (LF_XROWS / LF_ROWS - 1)) == 0 && t + 1 < ntiles) {   // tile t+1 (generated during this tile) starts the next 2048 rows
                fill_positions((t + 1) * LF_ROWS);
                __syncthreads();
            }
        }
        const int m = rbeg + t * LF_ROWS + li;
        const bool more = t + 1 < ntiles;
        f32x4 mk[4];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        if (!DGRAD) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bw, 1.0f, acc1, 0, 0, 0);
        const unsigned rowb = lds0 + (unsigned)((t & 1) * LF_TILE * 16 + li * 1024);
        // slot of chunk 2 jj + lh of this lane's row, low four bits swizzled: (2 jj + lh) ^ (li & 15) = 2 jj ^ (lh ^ (li & 15)), and the row base
        // is a multiple of 1 KB, so the address is one XOR away from a per-tile base (no table of eight addresses in registers)
        const unsigned adk = rowb + (unsigned)((lh ^ (li & 15)) * 16);
        auto rd = [&](int j, f32x4& x0) {
            const unsigned a = adk ^ (unsigned)(32 * (j & 7));
            if ((j >> 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 3) == 1) asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 3) == 2) asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(x0) : "v"(a) : "memory");
            if ((j >> 3) == 3) asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(x0) : "v"(a) : "memory");
        };
        f32x4 fa[2];              // ping-pong: fragment j+1 is read while the MFMAs of step j run
        rd(0, fa[0]);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            f32x4& c0 = fa[j & 1];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0) : : "memory");
            if (j + 1 < 32) rd(j + 1, fa[(j + 1) & 1]);
            if (j < 8 && (j & 1) == 0) { if (more) dma_row(t + 1, j >> 1); }
            if (K3W) {
                // column sums of tile t-1 (image (t-1) & 1, slot (t-1) & 3; zeros for t = 0): element r is read at k-step r, used at r + 1 -- done
                // before the mask loads of this tile take their registers
                if (j == 7 && more) k3_pos_dma(t + 1);
                if (j >= 1 && j < 17) k3_fma(j - 1);
                if (j < 16) k3_issue(t + 3, j);
            }
            if (!K3W && j >= 8 && j < 16 && (j & 1) == 0 && (!OUTV || op.store_hidden)) {
                const int q = (j - 8) >> 1;
                if (prev_m < rend) *reinterpret_cast<float4*>(g.C + (size_t)prev_m * g.ldc + 32 * wave + 8 * q + 4 * lh) = prev[q];
            }
            if (DGRAD && !K3W && j >= 16 && j < 24 && (j & 1) == 0) {
                const int q = (j - 16) >> 1;
                const float* mp = g.mask + (size_t)min(m, rend - 1) * g.ldmask + 32 * wave + 4 * lh + 8 * q;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mk[q]) : "v"(mp) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the read (and this step's memory instruction) ahead of this step's MFMAs
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].x, c0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].y, c0.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].z, c0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j].w, c0.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // ... and this step's MFMAs ahead of the next step's wait
        }
        if (DGRAD && !K3W) asm volatile("s_waitcnt vmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]) : : "memory");
        // lane (li, lh) holds row li of the tile, columns 32 wave + 8 q + 4 lh + (0..3) for q = 0..3: kept for the next tile's loop
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o = make_float4(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1],
                                   acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]);
            if (DGRAD && !K3W) {
                o.x = mk[q].x > 0.f ? o.x : 0.f; o.y = mk[q].y > 0.f ? o.y : 0.f;
                o.z = mk[q].z > 0.f ? o.z : 0.f; o.w = mk[q].w > 0.f ? o.w : 0.f;
            } else if (g.act == 1) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (K3W) {                               // into the tile's LDS image instead of memory; rows past the range count for nothing
                if (m >= rend) o = make_float4(0.f, 0.f, 0.f, 0.f);
                const f32x4 ov = {o.x, o.y, o.z, o.w};
                asm volatile("ds_write_b128 %0, %1" : : "v"(kd0 + (unsigned)((t & 1) * LF_TILE * 16 + li * 1024 + (((8 * wave + 2 * q + lh) ^ (li & 15)) << 4))), "v"(ov) : "memory");
            } else {
                prev[q] = o;
            }
        }
        prev_m = m;
        if (OUTV) outv_tile(t);
    }
    if (K3W) {
        // the last tile's column sums, the two row halves of a column folded through LDS, one atomic per gradient entry and block
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            k3_issue(ntiles - 1, r);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            k3_fma(r);
        }
        if (khalf == 1) lds[kn] = make_float4(k3w0, k3w1, k3w2, k3b);        // (the row stages are idle: every wave is past its last MFMA loop)
        __syncthreads();
        if (khalf == 0) {
            const float4 o = lds[kn];
            float* gw = grad_target(kp.gW0) + (size_t)kn * kp.ldgw0;
            unsafeAtomicAdd(gw + 0, k3w0 + o.x);
            unsafeAtomicAdd(gw + 1, k3w1 + o.y);
            unsafeAtomicAdd(gw + 2, k3w2 + o.z);
            unsafeAtomicAdd(grad_target(kp.gb0) + kn, k3b + o.w);
        }
        return;
    }
    if (prev_m < rend && (!OUTV || op.store_hidden)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(g.C + (size_t)prev_m * g.ldc + 32 * wave + 8 * q + 4 * lh) = prev[q];
    }
}

// Eligibility is decided by the caller (gemm.hip): N = K = 256, plain row-major A, 16-byte-aligned rows; forward: [n][k] weights,
// no mask; dgrad (b_trans): [k][n] weights, fp32 mask, no bias / activation.
int clift_layer_f32_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, LF_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();                    // one persistent block per CU
    const int rpb = cdiv(cdiv(p.M, blocks), LF_ROWS) * LF_ROWS;
    const GenP none = {nullptr, nullptr, 0, nullptr, nullptr, 0};
    const OutP no_out = {nullptr, 0, nullptr, 0, nullptr, 0, 1};
    if (b_trans) k_layer_f32<true, false, false><<<cdiv(p.M, rpb), 512, 0, st>>>(p, rpb, none, no_out, K3P{nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr});
    else k_layer_f32<false, false, false><<<cdiv(p.M, rpb), 512, 0, st>>>(p, rpb, none, no_out, K3P{nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_gemm(fp32 layer)");
}

// LAST hidden layer of an xyz head together with its narrow output layer (E <= 4: the instance heads, tensoRF.py:478-481):
//   h = relu(W A^T + b) (written to `hidden` only if it is non-null), out[:, 0:E] = h Wout^T + bout.
extern "C" int clift_xyz_head_last2_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                        const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4, "clift_xyz_head_last2_fwd: E must be in [1,4] (got %d)", E);
    CLIFT_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= 256 && ldw >= 256,
                  "clift_xyz_head_last2_fwd: A / W must be 16-byte aligned with pitches >= 256 that are multiples of 4");
    CLIFT_REQUIRE(hidden == nullptr || ((((uintptr_t)hidden) & 15) == 0 && ldh % 4 == 0 && ldh >= 256), "clift_xyz_head_last2_fwd: hidden must be 16-byte aligned, pitch >= 256");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = A; p.lda = lda; p.B = W; p.ldb = ldw; p.C = hidden; p.ldc = ldh; p.bias = b; p.act = 1;
    const int tiles = cdiv(M, LF_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), LF_ROWS) * LF_ROWS;
    const GenP none = {nullptr, nullptr, 0, nullptr, nullptr, 0};
    const OutP op = {Wout, ldwo, bout, E, out, ldo, hidden != nullptr ? 1 : 0};
    k_layer_f32<false, false, true><<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(p, rpb, none, op, K3P{nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_xyz_head_last2_fwd");
}

// First TWO layers of an xyz head in one launch: h2 = relu(W1 relu(W0 x + b0) + b1)   (tensoRF.py:475-478, 576-579), fp32.
// x4 (M,4); W0 (256,3) pitch ldw0, b0 (256); W1 (256,256) pitch ldw1, b1 (256); h2 (M, ldh2); h1 (nullable, (M, ldh1)) = the first
// layer's activation, written only when the caller needs it for a backward pass.
extern "C" int clift_xyz_head_first2_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1, const float* b1,
                                         int M, float* h1, int ldh1, float* h2, int ldh2, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)x4) & 15) == 0 && (((uintptr_t)W1) & 15) == 0 && (((uintptr_t)h2) & 15) == 0 && ldw1 % 4 == 0 && ldh2 % 4 == 0,
                  "clift_xyz_head_first2_fwd: x4 / W1 / h2 must be 16-byte aligned with pitches that are multiples of 4");
    CLIFT_REQUIRE(h1 == nullptr || ((((uintptr_t)h1) & 15) == 0 && ldh1 % 4 == 0 && ldh1 >= 256), "clift_xyz_head_first2_fwd: h1 must be 16-byte aligned, pitch >= 256");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = nullptr; p.lda = 256; p.B = W1; p.ldb = ldw1; p.C = h2; p.ldc = ldh2; p.bias = b1; p.act = 1;
    const int tiles = cdiv(M, LF_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), LF_ROWS) * LF_ROWS;
    const GenP gp = {x4, W0, ldw0, b0, h1, ldh1};
    const OutP no_out = {nullptr, 0, nullptr, 0, nullptr, 0, 1};
    k_layer_f32<false, true, false><<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(p, rpb, gp, no_out, K3P{nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_xyz_head_first2_fwd");
}

// Backward of the first TWO layers of an xyz head in one launch (the forward is clift_xyz_head_first2_fwd): with dH2 (M, ldd) the gradient
// at the second layer's output (already masked by its ReLU), dH1 = (W0 x + b0 > 0) . (dH2 W1) is formed tile by tile and consumed on the spot:
//   gW0[n][0..2] += sum_m dH1[m][n] x4[m][0..2],   gb0[n] += sum_m dH1[m][n]          (tensoRF.py:475-476, 576-577)
// dH1 is never written and the first layer's activation is not read (its sign is re-derived from the positions).  (The second layer's own
// weight gradient, dH2^T h1, stays a clift_gemm call.)
extern "C" int clift_xyz_head_first2_bwd(const float* dH2, int ldd, const float* W1, int ldw1, const float* W0, int ldw0, const float* b0, const float* x4,
                                         int M, float* gW0, int ldgw0, float* gb0, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)dH2) & 15) == 0 && (((uintptr_t)x4) & 15) == 0 && ldd % 4 == 0 && ldd >= 256 && ldw1 >= 256,
                  "clift_xyz_head_first2_bwd: dH2 / x4 must be 16-byte aligned, dH2's pitch >= 256 and a multiple of 4, W1's pitch >= 256");
    CLIFT_REQUIRE(W0 != nullptr && b0 != nullptr && ldw0 >= 3 && gW0 != nullptr && gb0 != nullptr && ldgw0 >= 3,
                  "clift_xyz_head_first2_bwd: W0 / gW0 (256 x 3, pitch >= 3) and b0 / gb0 (256) are required");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = dH2; p.lda = ldd; p.B = W1; p.ldb = ldw1; p.C = nullptr; p.ldc = 256;
    const int tiles = cdiv(M, LF_ROWS);
    const int blocks = tiles < clift_persistent_cus() ? tiles : clift_persistent_cus();
    const int rpb = cdiv(cdiv(M, blocks), LF_ROWS) * LF_ROWS;
    const GenP none = {nullptr, nullptr, 0, nullptr, nullptr, 0};
    const OutP no_out = {nullptr, 0, nullptr, 0, nullptr, 0, 1};
    k_layer_f32<true, false, false, true><<<cdiv(M, rpb), 512, 0, as_stream(s)>>>(p, rpb, none, no_out, K3P{x4, W0, ldw0, b0, gW0, ldgw0, gb0});
    return clift_check_launch("clift_xyz_head_first2_bwd");
}
