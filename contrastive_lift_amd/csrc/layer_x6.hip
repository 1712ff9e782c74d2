// layer_x6.hip -- the 256 -> 256 hidden layers of the xyz heads as PERSISTENT fp32-FAITHFUL kernels on the bf16 matrix cores
// ("fp32x6", clift_gemm precision = 2; forward C = relu(A W^T + b) and masked dgrad C = mask . (A W)).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TFLOP/s) and the exact kernel (layer_f32.hip) sits at 0.77 of it, limited by
// the clock the chip holds under dense fp32 MFMA load.  v_mfma_f32_32x32x16_bf16 is 16x faster.  Every fp32 value splits EXACTLY into
// three bf16 terms x = x1 + x2 + x3 (round-to-nearest each: 8 significant bits apiece, 24 together), so
//     x w = x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1) + O(2^-26 |x w|)
// is six bf16 products, each exact in the MFMA's fp32 accumulator; the dropped terms are below one fp32 rounding of the product.
// Six bf16 MFMAs cost 6/16 of the fp32-MFMA time.  (gemm_split.hip is the same arithmetic as a tiled kernel -- load-latency bound.)
//
// Shape of the kernel.  The three bf16 planes of a 256 x 256 weight matrix are 384 KB: they only fit in the register files of TWO CUs.
// So a row range is shared by a PAIR of blocks (ids 8 apart = the same XCD, so the second read of an activation row comes from that
// L2), each owning 128 output columns; a block is 4 waves (one per SIMD, 512-register budget), wave w owns 32 columns for all 256 k =
// 3 x 64 registers of weight fragments, split from the fp32 weights once per block.  Activation rows (fp32 in memory) are split
// COOPERATIVELY: per 32-row tile each lane loads eight 16-byte pieces into registers, splits them (22 VALU per piece) and writes three
// 8-byte pieces into the tile's bf16 plane images in LDS (2 stages x 3 planes x 16 KB); every wave then reads the whole tile as MFMA
// fragments (3 x ds_read_b128 per 16-k step, 16-byte chunk c of row r in slot c ^ (r & 15): conflict-free for reads and writes).
// The split of tile t+1 and the loads of tile t+2 are spread through the MFMA loop of tile t: one piece per two k-steps, a few VALU /
// one LDS write / one memory instruction per MFMA gap (pinned by sched_barriers), ONE s_barrier per tile.
// MFMA operands are swapped (weights first), so a lane owns one output row and stores four consecutive columns per instruction.
// The six products of a k-step alternate between two accumulator chains.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_x6)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int X6_ROWS = 32;                         // rows per tile
constexpr int X6_PLANE = X6_ROWS * 512;             // bytes of one bf16 plane image of a tile (32 rows x 256 k x 2 B)
constexpr int X6_STAGE = 3 * X6_PLANE;              // 48 KB

static __device__ __forceinline__ unsigned x6_pk(float lo, float hi) {        // two fp32 -> packed bf16 (RNE), `lo` in the low half
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
static __device__ __forceinline__ float x6_lo(unsigned p) { return __uint_as_float(p << 16); }
static __device__ __forceinline__ float x6_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// exact three-way split of a pair: planes[0..2] receive the packed (a, b) terms
static __device__ __forceinline__ void x6_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = x6_pk(a, b);
    const float ra = a - x6_lo(h), rb = b - x6_hi(h);
    m = x6_pk(ra, rb);
    const float sa = ra - x6_lo(m), sb = rb - x6_hi(m);
    l = x6_pk(sa, sb);
}

static __device__ __forceinline__ f32x16 x6_mfma(const u32x4& w, const u32x4& a, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// (non-temporal output stores, -DX6_NT_STORES: measured SLOWER -- forward 190 -> 217 us, dgrad 200 -> 258 us at 249 k rows; off)
#ifdef X6_NT_STORES
#define X6_STORE(ptr, val) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&(val)), reinterpret_cast<f32x4*>(ptr))
#else
#define X6_STORE(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#endif
constexpr int X6_RAW = 2 * X6_STAGE;                // fp32 staging: wave w owns rows w + 4 i as 1 KB slots (w * 8 + i), 32 KB

// vmcnt before the read-back of staged row i (tools/x6_vmcnt_model.py replays the instruction stream and prints these)
constexpr int X6_VM_FWD[8] = {11, 10, 10, 10, 10, 9, 9, 9};
constexpr int X6_VM_DGRAD[8] = {13, 12, 12, 13, 14, 13, 13, 13};

template <int N>
static __device__ __forceinline__ void x6_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool DGRAD>
static __device__ __forceinline__ void x6_wait_piece(int i) {       // (i is a constant after unrolling: one s_waitcnt survives)
    if (i == 0) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[0]>();
    if (i == 1) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[1]>();
    if (i == 2) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[2]>();
    if (i == 3) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[3]>();
    if (i == 4) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[4]>();
    if (i == 5) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[5]>();
    if (i == 6) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[6]>();
    if (i == 7) x6_wait_vm<(DGRAD ? X6_VM_DGRAD : X6_VM_FWD)[7]>();
}

// DGRAD = false: weights stored [n][k], bias + optional ReLU.  DGRAD = true: weights stored [k][n], fp32 ReLU mask.
//
// Memory-instruction bookkeeping.  All vector-memory instructions of the loop are issued by hand so that the waits can be COUNTED (the
// compiler's own accounting stops at the loop's back edge and falls back to vmcnt(0), which drained the newest prefetch every tile).
// Per tile and wave, in program order (m = mask load, dgrad only; D = LDS-DMA of one 1 KB row of tile t + 2; S = 16-byte store of the
// previous tile's results):   forward  D0 D1 D2 D3 D4 S0 D5 S1 D6 S2 D7 S3,   dgrad  m0 m1 D0 m2 D1 m3 D2 D3 D4 S0 D5 S1 D6 S2 D7 S3.
// The row D_i of tile t is read back (for the split) at step 2 i of tile t + 1; the number of younger instructions issued by then is a
// compile-time constant per i (X6_VM_*), the same for every tile because NOTHING in the loop is conditional: rows past the end of the
// range are CLAMPED to its last row on the way in (DMA, mask), so their results are copies of that row's and the store simply writes
// them to that row again; the first tile "stores" zeros to its own rows, which its real results overwrite later (same wave, same
// addresses, program order).
// OUTV (forward only): the layer is the LAST hidden layer of a head with a narrow output (E <= 4: the instance heads, tensoRF.py:480-481) and
// the output layer out[m][c] = sum_k h[m][k] Wout[c][k] + bout[c] is applied to the tile while it is in registers: 4 x 16 FMAs per lane
// against register-resident output weights (spread over the MFMA gaps of the following tile), half-waves folded with a permlane swap, the four
// waves' shares meet in LDS and are summed in a fixed order after the next barrier.  A block owns only 128 of the 256 hidden columns, so
// its sum is HALF a dot product: the two blocks of a pair add theirs to `out` with one float atomic each -- two addends on a zero-filled
// target commute, so the result does not depend on their order (the caller zero-fills `out`; bias comes with column half 0).
// OUTV = 2: the hidden activation itself is not written (no backward through the head) -- the kernel then has no output stream at all,
// which is worth more than the fused layer: the 255 MB of HBM writes of a plain launch cost ~55 us of its 190 (profiles/r03_x6_notes.txt).
struct X6Out {
    const float* Wout;    // (E, 256), row pitch ldwo
    int ldwo;
    const float* bout;    // (E), nullable
    int E;
    float* out;           // (M, ldo), column offset already applied, zero-filled by the caller
    int ldo;
};
constexpr int X6_PART = X6_RAW + X6_ROWS * 1024;    // OUTV: [tile parity][wave][64] float4 partial sums, 2 x 4 KB
// vmcnt before the read-back of staged row i with the output layer's atomic (X) at the top of every tile: X D0 D1 D2 D3 D4 S0 D5 S1 D6 S2 D7 S3
// and, without the hidden stores, X D0 .. D7
constexpr int X6_VM_OUTV1[8] = {12, 11, 11, 11, 11, 10, 10, 10};
constexpr int X6_VM_OUTV2[8] = {8, 7, 7, 7, 7, 7, 7, 7};

template <bool DGRAD, int OUTV>
static __device__ __forceinline__ void x6_wait_piece_v(int i) {
    if (OUTV == 0) { x6_wait_piece<DGRAD>(i); return; }
    if (i == 0) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[0]>();
    if (i == 1) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[1]>();
    if (i == 2) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[2]>();
    if (i == 3) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[3]>();
    if (i == 4) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[4]>();
    if (i == 5) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[5]>();
    if (i == 6) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[6]>();
    if (i == 7) x6_wait_vm<(OUTV == 1 ? X6_VM_OUTV1 : X6_VM_OUTV2)[7]>();
}

template <bool DGRAD, int OUTV = 0>
__global__ __launch_bounds__(256, 1) void k_layer_x6(GemmP g, int rows_per_range, int nranges, X6Out op) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * X6_STAGE + X6_ROWS * 1024 + (OUTV ? 8192 : 0)];     // 128 (136) KB, the only LDS object
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: row numbers, DMA bases and the clamps derived from it stay on the SALU
    const int b = blockIdx.x, half = (b >> 3) & 1, range = (b & 7) + 8 * (b >> 4);
    if (rows_limited()) {          // sync-free step: the launch was sized by a capacity; re-balance the row ranges over the true row count
        g.M = limit_rows(g.M);
        rows_per_range = ((g.M + nranges - 1) / nranges + X6_ROWS - 1) / X6_ROWS * X6_ROWS;
    }
    if (range >= nranges) return;
    const int rbeg = range * rows_per_range, rend = min(g.M, rbeg + rows_per_range);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + X6_ROWS - 1) / X6_ROWS;
    const int ncol = 128 * half + 32 * wave;                 // first output column of this wave
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    // ---- cooperative split: this wave provides rows wave + 4 i (i = 0..7) of every tile, lane = 16-byte piece (k = 4 lane .. +3)
    auto dma_piece = [&](int t, int i) {
        const int gr = min(rbeg + t * X6_ROWS + wave + 4 * i, rend - 1);         // rows past the range re-read its last row
        __builtin_amdgcn_global_load_lds(g.A + (size_t)gr * g.lda + 4 * lane, (lds_ptr_t)(lds + X6_RAW + (wave * 8 + i) * 1024), 16, 0, 0);
    };
    const unsigned rawa = lds0 + (unsigned)(X6_RAW + wave * 8192 + lane * 16);   // this lane's piece of staging slot (wave, 0)
    // LDS byte offset (inside a plane image) of this lane's 8-byte piece of row wave + 4 i; rows i and i + 4 differ by 8 KB exactly
    unsigned wofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave + 4 * i;
        wofs[i] = lds0 + (unsigned)(r * 512 + (((lane >> 1) ^ (r & 15)) << 4) + (lane & 1) * 8);
    }
    // the split of one piece, in parts that fit MFMA gaps; sp_* carry the intermediate terms
    f32x4 sp_x;
    unsigned sp_h0, sp_h1, sp_m0, sp_m1;
    float sp_r0, sp_r1, sp_r2, sp_r3, sp_s0, sp_s1, sp_s2, sp_s3;
    auto raw_read = [&](int i) {
        if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 3) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 4) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 5) asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 6) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 7) asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(sp_x) : "v"(rawa) : "memory");
    };
    // (the packed words are made opaque: otherwise the compiler re-derives "low half << 16" with a second conversion of the same value)
    auto split_a = [&]() {
        sp_h0 = x6_pk(sp_x[0], sp_x[1]); sp_h1 = x6_pk(sp_x[2], sp_x[3]);
        asm volatile("" : "+v"(sp_h0), "+v"(sp_h1));
    };
    auto split_b0 = [&]() { sp_r0 = sp_x[0] - x6_lo(sp_h0); sp_r1 = sp_x[1] - x6_hi(sp_h0); };
    auto split_b1 = [&]() { sp_r2 = sp_x[2] - x6_lo(sp_h1); sp_r3 = sp_x[3] - x6_hi(sp_h1); };
    auto wr = [&](unsigned addr, int plane, int i, unsigned a, unsigned c) {
        const u32x2 d = {a, c};
        const int off = plane * X6_PLANE + (i >> 2) * 8192;                      // < 65536: fits the instruction's offset field
        if (off == 0) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(d) : "memory");
        if (off == 8192) asm volatile("ds_write_b64 %0, %1 offset:8192" : : "v"(addr), "v"(d) : "memory");
        if (off == 16384) asm volatile("ds_write_b64 %0, %1 offset:16384" : : "v"(addr), "v"(d) : "memory");
        if (off == 24576) asm volatile("ds_write_b64 %0, %1 offset:24576" : : "v"(addr), "v"(d) : "memory");
        if (off == 32768) asm volatile("ds_write_b64 %0, %1 offset:32768" : : "v"(addr), "v"(d) : "memory");
        if (off == 40960) asm volatile("ds_write_b64 %0, %1 offset:40960" : : "v"(addr), "v"(d) : "memory");
    };
    auto split_c = [&](unsigned stage, int i) {
        sp_m0 = x6_pk(sp_r0, sp_r1); sp_m1 = x6_pk(sp_r2, sp_r3);
        asm volatile("" : "+v"(sp_m0), "+v"(sp_m1));
        wr(wofs[i & 3] + stage, 0, i, sp_h0, sp_h1);
    };
    auto split_d0 = [&]() { sp_s0 = sp_r0 - x6_lo(sp_m0); sp_s1 = sp_r1 - x6_hi(sp_m0); };
    auto split_d1 = [&]() { sp_s2 = sp_r2 - x6_lo(sp_m1); sp_s3 = sp_r3 - x6_hi(sp_m1); };
    auto split_e = [&](unsigned stage, int i) {
        wr(wofs[i & 3] + stage, 1, i, sp_m0, sp_m1);
        wr(wofs[i & 3] + stage, 2, i, x6_pk(sp_s0, sp_s1), x6_pk(sp_s2, sp_s3));
    };

    // ---- prologue: tile 0's rows -> staging -> split into stage 0; tile 1's rows on their way while the weights are prepared
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(0, i);
    x6_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        raw_read(i);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sp_x) : : "memory");
        split_a(); split_b0(); split_b1(); split_c(0u, i); split_d0(); split_d1(); split_e(0u, i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave is done with its staging slots
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(1, i);

    // ---- weight fragments: w?[j] = planes of W(n = ncol + li, k = 16 j + 8 lh .. +7), split once per block
    u32x4 wh[16], wm[16], wl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v[8];
        if (!DGRAD) {
            const float* q = g.B + (size_t)(ncol + li) * g.ldb + 16 * j + 8 * lh;
            const float4 a = *reinterpret_cast<const float4*>(q), c = *reinterpret_cast<const float4*>(q + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        } else {
            const float* q = g.B + (size_t)(16 * j + 8 * lh) * g.ldb + ncol + li;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = q[(size_t)e * g.ldb];
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            unsigned h, m, l;
            x6_split_pair(v[2 * pr], v[2 * pr + 1], h, m, l);
            wh[j][pr] = h; wm[j][pr] = m; wl[j][pr] = l;
        }
    }
    // bias of this lane's 16 output columns (accumulator register r <-> column ncol + 8 (r >> 2) + 4 lh + (r & 3))
    f32x16 bv;
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = (!DGRAD && g.bias) ? g.bias[ncol + 8 * (r >> 2) + 4 * lh + (r & 3)] : 0.f;
    x6_wait_vm<0>();                     // tile 1's rows have landed: from here on every wait is counted

    // fragment address of this lane: row li, chunk 2 j + lh -> slot (2 j + lh) ^ (li & 15) = (2 j) ^ (lh ^ (li & 15)): one XOR per step
    const unsigned adk = lds0 + (unsigned)(li * 512 + ((lh ^ (li & 15)) << 4));
    float4 prev[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int prev_m = min(rbeg + li, rend - 1);                                       // (tile 0 "stores" zeros to its own rows first)

    // The accumulators of a tile are turned into results at the TOP of the next iteration, between the issue of that tile's first fragment
    // reads (behind the barrier) and their use: the ~32 VALU instructions cover the LDS latency that nothing else can (the reads cannot be
    // issued before the barrier), and the last MFMAs of the tile drain meanwhile.  Iteration 0 "finishes" zero accumulators (prev = 0).
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x4 mk[4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    auto finish = [&]() {
        // lane (li, lh) holds row li of the tile, columns ncol + 8 q + 4 lh + (0..3) for q = 0..3: kept for the next tile's loop
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o = make_float4(acc0[4 * q + 0] + acc1[4 * q + 0], acc0[4 * q + 1] + acc1[4 * q + 1],
                                   acc0[4 * q + 2] + acc1[4 * q + 2], acc0[4 * q + 3] + acc1[4 * q + 3]);
            if (DGRAD) {
                o.x = mk[q][0] > 0.f ? o.x : 0.f; o.y = mk[q][1] > 0.f ? o.y : 0.f;
                o.z = mk[q][2] > 0.f ? o.z : 0.f; o.w = mk[q][3] > 0.f ? o.w : 0.f;
            } else if (g.act == 1) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            prev[q] = o;
        }
    };
    int m_done = prev_m;                                                         // row of the accumulators waiting to be finished
    // ---- OUTV state: output weights of this lane's 16 columns, its running share of the E dot products, row bookkeeping two tiles deep
    float wo[4][16];
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    const unsigned part0 = lds0 + (unsigned)X6_PART;
    int t_fin = -1;                          // tile whose results are in `prev` (their shares are being summed during the current tile)
    float bo_c = 0.f;                        // bias of output lane & 3, added by column half 0 only
    if (OUTV) {
        if (half == 0 && op.bout && (lane & 3) < op.E) bo_c = op.bout[lane & 3];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) wo[c][r] = c < op.E ? op.Wout[(size_t)c * op.ldwo + ncol + 8 * (r >> 2) + 4 * lh + (r & 3)] : 0.f;
    }
    // two registers of `prev` per call (part p = 0..7 <-> prev[p >> 1] components 2 (p & 1), +1) against the four output rows: 8 FMAs
    auto outv_fma = [&](int p_) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 2 * p_ + u;
            const float h = prev[r >> 2][r & 3];
#pragma unroll
            for (int c = 0; c < 4; ++c) pv[c] = fmaf(h, wo[c][r], pv[c]);
        }
    };
    // fold the two half-waves and park the wave's share in LDS (both halves write: no branch beside the hand-issued memory instructions)
    auto outv_park = [&](int par) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned u = __float_as_uint(pv[c]);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            pv[c] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        asm volatile("ds_write_b128 %0, %1" : : "v"(part0 + (unsigned)(par * 4096 + wave * 1024 + lane * 16)), "v"(pv) : "memory");
        pv = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // lanes (r = (lane >> 2) & 7, c = lane & 3) of wave w: the four waves' shares of row 8 w + r, output c, of tile `tile` (parked with parity par)
    f32x4 sh;
    auto outv_fetch = [&](int par) {
        const unsigned a = part0 + (unsigned)(par * 4096 + (8 * wave + ((lane >> 2) & 7)) * 16 + (lane & 3) * 4);
        asm volatile("ds_read_b32 %0, %1" : "=v"(sh[0]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(sh[1]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:2048" : "=v"(sh[2]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:3072" : "=v"(sh[3]) : "v"(a) : "memory");
    };
    auto outv_add = [&](int tile) {          // fixed order; ONE atomic per (row, output) and block; lanes 32..63 and invalid rows add 0 to a valid address
        const int c = lane & 3, row = rbeg + tile * X6_ROWS + 8 * wave + ((lane >> 2) & 7);
        const bool live = tile >= 0 && lane < 32 && c < op.E && row < rend;
        float v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        v += bo_c;
        v = live ? v : 0.f;
        const int rr = min(max(row, rbeg), rend - 1), cc = min(c, op.E - 1);
        unsafeAtomicAdd(op.out + (size_t)rr * op.ldo + cc, v);
    };

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // this wave's plane writes of tile t are done ...
        __builtin_amdgcn_s_barrier();                                            // ... and everyone's; everyone is done reading the other stage
        asm volatile("" ::: "memory");
        const unsigned cur = (unsigned)((t & 1) * X6_STAGE), nxt = (unsigned)(((t + 1) & 1) * X6_STAGE);
        const int m = min(rbeg + t * X6_ROWS + li, rend - 1);                    // clamped like the loads: see above
        u32x4 fa[2][3];                  // ping-pong fragments (hi, mid, lo) of the activation tile
        auto rd = [&](int j, u32x4 (&f)[3]) {
            const unsigned a = (adk + cur) ^ (unsigned)(32 * j);
            asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(f[1]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(f[2]) : "v"(a) : "memory");
        };
        if (OUTV) outv_fetch((t + 1) & 1);                                      // parked during tile t - 1: the shares of tile t - 2's rows
        rd(0, fa[0]);
        if (OUTV) {
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(sh) : : "memory");        // (the three fragment reads stay in flight)
            outv_add(t - 2);
        }
        finish();                        // the previous tile's results (prev), while the reads are in flight
        prev_m = m_done;
        m_done = m;
        t_fin = t - 1;
        acc0 = bv;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32x4 (&f)[3] = fa[j & 1];
            const int i = j >> 1;        // piece (row wave + 4 i) of tile t + 1 handled during steps 2 i, 2 i + 1 (its plane writes 1 and 2: step 2 i + 2)
            const bool even = (j & 1) == 0;
            // this step's fragments were issued in gap 0 of the previous step; LDS instructions issued since (tools/x6_vmcnt_model.py):
            // j = 0: none (they were issued just now, behind the barrier); other even steps: one plane write; odd steps: the staging read and two
            // plane writes (step 1: the staging read only)
            if (j == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            else if (even || j == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 0: next step's fragments.  Even: the staged fp32 piece i (DMA'd during tile t - 1) starts its way into registers.
            //      Odd: that piece has had six MFMAs to arrive: wait for it (younger: two plane writes + the three reads just issued), first split
            if (j + 1 < 16) rd(j + 1, fa[(j + 1) & 1]);
            if (even) {
                x6_wait_piece_v<DGRAD, OUTV>(i);
                raw_read(i);
            } else {
                if (j == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(sp_x) : : "memory");
                else if (j == 15) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(sp_x) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(sp_x) : : "memory");
                split_a();
            }
            acc0 = x6_mfma(wh[j], f[0], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 1
            if (even) { if (i > 0) split_e(nxt, i - 1); }                        // planes 1 and 2 of the previous piece
            else split_b0();
            acc1 = x6_mfma(wh[j], f[1], acc1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 2
            if (even) {
                if (DGRAD && i < 4) {                                            // the ReLU mask of this tile's rows: four loads, steps 0, 2, 4, 6
                    const float* mp = g.mask + (size_t)m * g.ldmask + ncol + 8 * i + 4 * lh;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mk[i]) : "v"(mp) : "memory");
                }
            } else split_b1();
            acc0 = x6_mfma(wm[j], f[1], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 3
            if (even) { if (i > 0) dma_piece(t + 2, i - 1); }                    // the slot consumed one step ago is refilled with tile t + 2's row
            else split_c(nxt, i);
            acc1 = x6_mfma(wm[j], f[0], acc1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 4
            if (even) {
                if (i >= 5 && OUTV != 2) {                                                    // the previous tile's results leave: steps 10, 12, 14 (and 15)
                    const int q = i - 5;
                    X6_STORE(g.C + (size_t)prev_m * g.ldc + ncol + 8 * q + 4 * lh, prev[q]);
                }
            } else { split_d0(); if (j == 15) split_d1(); }
            acc0 = x6_mfma(wh[j], f[2], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 5
            if (even) { if (OUTV) outv_fma(i); }
            else {
                if (j < 15) split_d1();
                else {                                                           // last piece of the tile: everything of it has to be out before the barrier
                    split_e(nxt, 7);
                    dma_piece(t + 2, 7);
                    if (OUTV != 2) X6_STORE(g.C + (size_t)prev_m * g.ldc + ncol + 8 * 3 + 4 * lh, prev[3]);
                    if (OUTV) outv_park(t & 1);                                  // the shares of the rows finished at the top of this tile
                }
            }
            acc1 = x6_mfma(wl[j], f[0], acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DGRAD) asm volatile("s_waitcnt vmcnt(10)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]) : : "memory");
    }
    if (!OUTV) {
        finish();
#pragma unroll
        for (int q = 0; q < 4; ++q) X6_STORE(g.C + (size_t)m_done * g.ldc + ncol + 8 * q + 4 * lh, prev[q]);
        return;
    }
    // OUTV drain: the shares of tile ntiles - 2 are parked (parity (ntiles - 1) & 1); the last tile still sits in the accumulators
    x6_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    outv_fetch((ntiles + 1) & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sh) : : "memory");
    outv_add(ntiles - 2);
    finish();
    if (OUTV != 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) X6_STORE(g.C + (size_t)m_done * g.ldc + ncol + 8 * q + 4 * lh, prev[q]);
    }
#pragma unroll
    for (int p_ = 0; p_ < 8; ++p_) outv_fma(p_);
    outv_park(ntiles & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    outv_fetch(ntiles & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sh) : : "memory");
    outv_add(ntiles - 1);
    (void)t_fin;
}

// Eligibility is decided by the caller (gemm.hip): N = K = 256, plain row-major fp32 A, 16-byte-aligned rows; forward: [n][k] weights,
// no mask; dgrad (b_trans): [k][n] weights, fp32 mask, no bias / activation.
int clift_layer_x6_launch(const GemmP& p, int b_trans, hipStream_t st) {
    const int tiles = cdiv(p.M, X6_ROWS);
    const int pairs = clift_persistent_cus() / 2;                    // two blocks (CUs) per row range
    const int nranges = tiles < pairs ? tiles : pairs;
    const int rpr = cdiv(cdiv(p.M, nranges), X6_ROWS) * X6_ROWS;
    const int nr = cdiv(p.M, rpr);
    const int grid = 16 * cdiv(nr, 8);                               // block b: half (b >> 3) & 1 of range (b & 7) + 8 (b >> 4)
    const X6Out none = {nullptr, 0, nullptr, 0, nullptr, 0};
    if (b_trans) k_layer_x6<true, 0><<<grid, 256, 0, st>>>(p, rpr, nr, none);
    else k_layer_x6<false, 0><<<grid, 256, 0, st>>>(p, rpr, nr, none);
    return clift_check_launch("clift_gemm(fp32x6 layer)");
}

// LAST hidden layer of an xyz head together with its narrow output layer, fp32x6 form of clift_xyz_head_last2_fwd (tensoRF.py:478-481):
//   h = relu(A W^T + b) (written to `hidden` only if it is non-null), out[:, 0:E] = h Wout^T + bout.
// `out` columns 0..E-1 of rows 0..M-1 are zero-filled here (stream-ordered) and then receive one atomic add from each of the two blocks that
// share a row (two addends: order-independent).
extern "C" int clift_xyz_head_last2_x6_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                           const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4, "clift_xyz_head_last2_x6_fwd: E must be in [1,4] (got %d)", E);
    CLIFT_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= 256 && ldw >= 256,
                  "clift_xyz_head_last2_x6_fwd: A / W must be 16-byte aligned with pitches >= 256 that are multiples of 4");
    CLIFT_REQUIRE(hidden == nullptr || ((((uintptr_t)hidden) & 15) == 0 && ldh % 4 == 0 && ldh >= 256), "clift_xyz_head_last2_x6_fwd: hidden must be 16-byte aligned, pitch >= 256");
    CLIFT_REQUIRE(ldo >= E, "clift_xyz_head_last2_x6_fwd: ldo < E");
    hipStream_t st = as_stream(s);
    if (hipMemset2DAsync(out, (size_t)ldo * sizeof(float), 0, (size_t)E * sizeof(float), (size_t)M, st) != hipSuccess) {
        clift_set_error("clift_xyz_head_last2_x6_fwd: zero-fill of the output failed");
        return 2;
    }
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = A; p.lda = lda; p.B = W; p.ldb = ldw; p.C = hidden; p.ldc = ldh; p.bias = b; p.act = 1;
    const int tiles = cdiv(M, X6_ROWS);
    const int pairs = clift_persistent_cus() / 2;
    const int nranges = tiles < pairs ? tiles : pairs;
    const int rpr = cdiv(cdiv(M, nranges), X6_ROWS) * X6_ROWS;
    const int nr = cdiv(M, rpr);
    const int grid = 16 * cdiv(nr, 8);
    const X6Out op = {Wout, ldwo, bout, E, out, ldo};
    if (hidden) k_layer_x6<false, 1><<<grid, 256, 0, st>>>(p, rpr, nr, op);
    else k_layer_x6<false, 2><<<grid, 256, 0, st>>>(p, rpr, nr, op);
    return clift_check_launch("clift_xyz_head_last2_x6_fwd");
}
