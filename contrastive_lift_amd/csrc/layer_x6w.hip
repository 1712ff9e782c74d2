// layer_x6w.hip -- weight gradient of the 256 x 256 hidden layers, fp32-FAITHFUL on the bf16 matrix cores ("fp32x6"):
//     gW[n][k] += sum_m dY[m][n] X[m][k],   gb[n] += sum_m dY[m][n]            (dY, X: M x 256 fp32, row-major)
// as six v_mfma_f32_32x32x16_bf16 products of exactly three-way-split operands (see layer_x6.hip for the arithmetic).  Unlike the forward /
// dgrad kernels this one has NO output stream -- the 128 x 128 partial of a workgroup is flushed once, with atomics -- so the format's
// advantage is not eaten by HBM writes (profiles/r03_x6_notes.txt).
//
// Shape: 64 row ranges x 4 quadrant workgroups (128 n x 128 k each; the four of a range have block ids 8 apart = one XCD, so dY / X come
// from HBM once), EIGHT waves = 4 n-tiles x 2 pairs of k-tiles, two per SIMD.  BOTH operands are streamed, and the MFMA contraction index is
// the ROW index m, so a fragment is 8 consecutive rows of one column.  The split does the transposition: rows travel by LDS-DMA into fp32
// staging (wave v owns rows 8 (v >> 1) .. + 7 of a 32-row tile x columns 64 (v & 1) .. + 63 of the quadrant, both operands, 4 KB), each lane
// reads COLUMN runs of 4 rows (4 x ds_read_b32, lanes along the columns: conflict-free), splits the 4 values (22 VALU) and writes three 8-byte
// pieces into TRANSPOSED bf16 plane images [operand][plane][column][32 m], column pitch 80 bytes.  Fragment reads (ds_read_b128: banks mod 64,
// 16-lane groups whose columns cover every residue mod 16; 80 = 5 x 16) are conflict-free at that pitch.  The WRITES (ds_write_b64: banks
// mod 32, groups of 16 CONTIGUOUS lanes) are not when every lane of a wave writes the same rows: columns 8 apart are 640 bytes = 5 x 128 apart,
// the same banks -- 2-way, 24 % of the kernel's LDS cycles in round 5 (profiles/r05_pmc_tables.txt) and no 16-byte-multiple pitch removes it.
// So lanes l and l + 8 write DIFFERENT rows: in run u a lane takes the row quad ((lane >> 3) & 1) ^ u of its wave's eight rows, i.e. 8 bytes
// further into its column than its neighbour eight lanes away -- the 16 lanes of a group then cover the 32 banks exactly once (round 6).
// A wave multiplies the dY fragments of its 32 n against the X fragments of its two k-tiles: 24 MFMAs per tile in two groups of 12 that
// alternate between two accumulators.  Rows past the end of a range are zeroed in the split (a clamped copy would be counted twice).
// Memory instructions per tile and wave: X rows x 2 (group 0), dY rows x 2 (group 1) -- every wait is vmcnt(2).
//
// GPU sharing.  A first build of this kernel (139 registers per wave) made ANOTHER process's kernels return wrong values whenever the two
// shared the device: bisected with a standalone copy (profiles/r03_x6_notes.txt) to its bf16 MFMAs -- not its LDS-DMA, not its atomics, not
// its end -- and to CO-RESIDENCY: with 2 x 176 of a SIMD's 512 registers taken, waves of the other process' small kernels were scheduled onto the
// same SIMD beside the MFMA stream and came back with wrong lanes 48..63 (59 % of the neighbour's launches; one process alone: never).  The
// remedy is a register ballast: every wave claims 256 registers (one asm clobber of v255), two of them fill the SIMD's file and nothing else
// fits beside them -- 0 wrong results in 43 000 launches of the same neighbour, same speed.  k_layer_x6 carries the same ballast.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_x6w)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int XW_ROWS = 32;
constexpr int XW_PITCH = 80;                        // bytes per column of a transposed plane image: 32 m x 2 B + 16
constexpr int XW_PLANE = 128 * XW_PITCH;            // 10240
constexpr int XW_STAGE = 6 * XW_PLANE;              // [operand 0 = dY, 1 = X][plane]: 61440
constexpr int XW_RAW = 2 * XW_STAGE;                // fp32 staging: wave w: [operand][8 rows][512 B] = 8 KB

static __device__ __forceinline__ unsigned xw_pk(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
static __device__ __forceinline__ float xw_lo(unsigned p) { return __uint_as_float(p << 16); }
static __device__ __forceinline__ float xw_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
static __device__ __forceinline__ f32x16 xw_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// GENX (clift_xyz_head_first2_x6_wgrad): the layer is the SECOND layer of an xyz head and its input X[m][k] = relu(W0[k] . x_m + b0[k]) (K = 3,
// tensoRF.py:475,576) is GENERATED in the split instead of streamed -- the forward never wrote it (clift_xyz_head_first2_x6_fwd).  The X rows'
// two DMA slots per tile carry the POSITIONS of the wave's four rows (2 x 32 bytes), a column run reads them back as four broadcast
// ds_read_b128 where it read four column words, and the lane evaluates its column (4 coefficients per run in registers; the forward's FMA
// order, so the operand has the forward's bits).  Same number of vector-memory and LDS instructions in the same places: every wait is unchanged.
struct XwGen {
    const float* x4;      // (M, 4) normalised sample positions
    const float* W0;      // (256, 3), row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
};

template <bool GENX>
__global__ __launch_bounds__(512, 2) void k_wgrad_x6(GemmP g, int rows_per_range, int nranges, XwGen gx) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[XW_RAW + 8 * 4096];                 // 152 KB, the only LDS object
    asm volatile("v_mov_b32 v255, 0" ::: "v255");       // register ballast (see the header): the wave allocates the whole 256-register budget
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3, kp = wave >> 2;       // n-tile, pair of k-tiles (scalars)
    const int b = blockIdx.x, quad = (b >> 3) & 3, qn = quad >> 1, qk = quad & 1, range = (b & 7) + 8 * (b >> 5);
    (void)gx;
    if (rows_limited()) {
        g.K = limit_rows(g.K);
        rows_per_range = ((g.K + nranges - 1) / nranges + XW_ROWS - 1) / XW_ROWS * XW_ROWS;
    }
    if (range >= nranges) return;
    const int rbeg = range * rows_per_range, rend = min(g.K, rbeg + rows_per_range);
    if (rbeg >= rend) return;
    const int ntiles = (rend - rbeg + XW_ROWS - 1) / XW_ROWS;
    const float* __restrict__ Y = g.A + 128 * qn;       // dY columns of this quadrant's n-half
    const float* __restrict__ X = g.B + 128 * qk;       // X columns of its k-half
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    // ---- rows by LDS-DMA.  The wave owns rows 8 ro .. 8 ro + 7 of the tile x columns 64 ch .. 64 ch + 63 of the quadrant; piece j (0, 1) of operand o =
    // rows 4 j .. 4 j + 3 of those eight, 256 B each: lane = (row lane >> 4, 16-byte chunk lane & 15), staged as [8 rows][256 B] per operand
    const int ro = wave >> 1, ch = wave & 1, lq = (lane >> 3) & 1;
    // Buffer loads (raw, stride 0) with the descriptor cut to THIS range's rows: a row at or past the end of the range is out of bounds and
    // arrives as zeros -- no clamping, no masking in the split (a zero dY row contributes nothing to gW or gb, whatever the X row holds; GENX:
    // the generated X of a zero position is finite) -- and the per-lane part of the address is one 32-bit offset computed once: a DMA costs
    // no vector instruction besides itself (round 6: the loop was bound by its vector issue slots, 189 per wave and tile for 24 MFMAs).
    const int nrows = __builtin_amdgcn_readfirstlane(rend - rbeg);
    auto uniform_rsrc = [](const float* base, unsigned bytes) {        // descriptor inputs made PROVABLY wave-uniform (no waterfall loop around the loads)
        const unsigned long long a = (unsigned long long)(uintptr_t)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), 0,
                                                 (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_y = uniform_rsrc(Y + (size_t)rbeg * g.lda, (unsigned)((((size_t)(nrows - 1)) * g.lda + 128) * 4));
    const __amdgpu_buffer_rsrc_t rs_x = GENX ? uniform_rsrc(gx.x4 + (size_t)rbeg * 4, (unsigned)nrows * 16u)
                                             : uniform_rsrc(X + (size_t)rbeg * g.ldb, (unsigned)((((size_t)(nrows - 1)) * g.ldb + 128) * 4));
    const int vo_y = ((lane >> 4) * g.lda + 64 * ch + 4 * (lane & 15)) * 4;             // byte offsets of this lane's 16 bytes within a piece
    const int vo_x = GENX ? (lane & 3) * 16 : ((lane >> 4) * g.ldb + 64 * ch + 4 * (lane & 15)) * 4;
    const int so_y = g.lda * 16, so_x = GENX ? 64 : g.ldb * 16;                         // bytes per 4 rows
    auto dma = [&](int t, int o, int j) {
        const int piece = __builtin_amdgcn_readfirstlane(t * (XW_ROWS / 4) + 2 * ro + j);     // 4-row piece of the range (scalar)
        if (o) CLIFT_MARK_DMA("x"); else CLIFT_MARK_DMA("y");
        if (o) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(lds + XW_RAW + wave * 4096 + 2048 + j * 1024), 16, vo_x, piece * so_x, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr_t)(lds + XW_RAW + wave * 4096 + j * 1024), 16, vo_y, piece * so_y, 0, 0);
    };
    // ---- a column run: 4 rows (quad lq ^ u of the wave's eight) at column 64 ch + lane of operand o
    float v[4];
    f32x4 pz[4];                                        // GENX: the positions of the quad's four rows
    float gw[4] = {0.f, 0.f, 0.f, 0.f};                 // GENX: W0 row and bias of this lane's X column (of this quadrant's k-half)
    if (GENX) {
        const int col = 128 * qk + 64 * ch + lane;
        const float* wr0 = gx.W0 + (size_t)col * gx.ldw0;
        gw[0] = wr0[0]; gw[1] = wr0[1]; gw[2] = wr0[2]; gw[3] = gx.b0[col];
    }
    auto run_read = [&](int o, int u) {
        const int q = lq ^ u;
        if (GENX && o == 1) {                               // quad 1 reads the copy held by lanes 4..7 of its piece (+ 64 B: other banks than quad 0's)
            const unsigned a = lds0 + (unsigned)(XW_RAW + wave * 4096 + 2048 + q * (1024 + 64));
            asm volatile("ds_read_b128 %0, %1" : "=v"(pz[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(pz[1]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(pz[2]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(pz[3]) : "v"(a) : "memory");
            return;
        }
        const unsigned a = lds0 + (unsigned)(XW_RAW + wave * 4096 + o * 2048 + q * 1024 + lane * 4);
        asm volatile("ds_read_b32 %0, %1" : "=v"(v[0]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(v[1]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(v[2]) : "v"(a) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(v[3]) : "v"(a) : "memory");
    };
    auto run_wait = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory"); };
    auto gen_wait = [&](int u) {                        // GENX, operand X: wait for the positions, then this lane's column of the four rows
        (void)u;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pz[0]), "+v"(pz[1]), "+v"(pz[2]), "+v"(pz[3]) : : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e)     // same order as k_linear_k3_fwd / k_layer_x6<GEN>
            v[e] = fmaxf(fmaf(gw[2], pz[e][2], fmaf(gw[1], pz[e][1], fmaf(gw[0], pz[e][0], gw[3]))), 0.f);
    };
    float bs0 = 0.f;                                    // bias gradient of column 64 ch + lane (this wave's rows)
    unsigned sh[2], sm[2], sl[2];
    auto run_mask = [&](int valid, int o, int u) {      // (rows past the end of the range arrive as zeros: nothing to mask)
        (void)valid; (void)u;
        if (o == 0) bs0 += (v[0] + v[1]) + (v[2] + v[3]);
    };
    auto run_split = [&](int q) {                       // pair q (rows 2 q, 2 q + 1 of the run)
        const float a = v[2 * q], c = v[2 * q + 1];
        unsigned h = xw_pk(a, c);
        asm volatile("" : "+v"(h));
        const float ra = a - xw_lo(h), rc = c - xw_hi(h);
        unsigned m = xw_pk(ra, rc);
        asm volatile("" : "+v"(m));
        sh[q] = h; sm[q] = m; sl[q] = xw_pk(ra - xw_lo(m), rc - xw_hi(m));
    };
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto run_write = [&](unsigned stage, int o, int u) {
        const unsigned a = lds0 + stage + (unsigned)(o * 3 * XW_PLANE + (64 * ch + lane) * XW_PITCH + (2 * ro + (lq ^ u)) * 8);
        const u32x2 dh = {sh[0], sh[1]}, dm = {sm[0], sm[1]}, dl = {sl[0], sl[1]};
        asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(dh) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:10240" : : "v"(a), "v"(dm) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:20480" : : "v"(a), "v"(dl) : "memory");
    };
    auto valid_of = [&](int t) { return rend - (rbeg + t * XW_ROWS); };

    // ---- prologue: tile 0 -> staging -> stage 0; tile 1's rows on their way (X first, then dY: the order the loop keeps)
    dma(0, 1, 0); dma(0, 1, 1); dma(0, 0, 0); dma(0, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = r < 2 ? 1 : 0, u = r & 1;
        run_read(o, u);
        if (GENX && o == 1) gen_wait(u); else run_wait();
        run_mask(valid_of(0), o, u); run_split(0); run_split(1); run_write(0u, o, u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma(1, 1, 0); dma(1, 1, 1); dma(1, 0, 0); dma(1, 0, 1);

    f32x16 acc[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    // fragment addresses: column (32 wn | 32 (2 kp + x)) + li, rows 16 s + 8 lh
    const unsigned fa = lds0 + (unsigned)((32 * wn + li) * XW_PITCH + 16 * lh);
    const unsigned fb = lds0 + (unsigned)(3 * XW_PLANE + (64 * kp + li) * XW_PITCH + 16 * lh);

    // ONE barrier per tile, and not at the top: a wave's last plane-image write of tile t + 1 happens two MFMAs before the end of tile t, so the
    // barrier stands there and the first fragments of tile t + 1 (k-step 0: their registers are free during the second group) are read behind
    // it, under the tile's last MFMAs -- the loop top used to expose barrier skew + the latency of nine ds_read_b128 every tile
    // (SQ_WAIT_INST_LDS 18 % of the wave cycles, profiles/r06_pmc_tables.txt).  Every wave has finished its own fragment reads of tile t when
    // it arrives (it waited for them at the start of the second group), so the writes of tile t + 1 into tile t's stage are safe behind it.
    u32x4 A[2][3], B[2][2][3];                          // dY fragments of k-step s; X fragments of k-step s, k-tiles 2 kp, 2 kp + 1
    auto rd = [&](int s, unsigned cur) {
        const unsigned a = fa + cur + (unsigned)(32 * s);
        asm volatile("ds_read_b128 %0, %1" : "=v"(A[s][0]) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(A[s][1]) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(A[s][2]) : "v"(a) : "memory");
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const unsigned c = fb + cur + (unsigned)(x * 32 * XW_PITCH + 32 * s);
            asm volatile("ds_read_b128 %0, %1" : "=v"(B[s][x][0]) : "v"(c) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(B[s][x][1]) : "v"(c) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(B[s][x][2]) : "v"(c) : "memory");
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd(0, 0u);
    for (int t = 0; t < ntiles; ++t) {
        const unsigned cur = (unsigned)((t & 1) * XW_STAGE), nxt = (unsigned)(((t + 1) & 1) * XW_STAGE);
        const int valid = valid_of(t + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int o = s == 0 ? 1 : 0;                                        // the operand whose two runs are split during this group: X, then dY
            u32x4 (&a)[3] = A[s];
            u32x4 (&b0)[3] = B[s][0];
            u32x4 (&b1)[3] = B[s][1];
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) rd(1, cur);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                     // this operand's rows (DMA'd during the previous tile) have landed
            if (o) CLIFT_MARK_USE("x", "0"); else CLIFT_MARK_USE("y", "0");
            run_read(o, 0);
            acc[0] = xw_mfma(a[0], b0[0], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = xw_mfma(a[0], b1[0], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (GENX && o == 1) gen_wait(0); else run_wait();
            run_mask(valid, o, 0);
            acc[0] = xw_mfma(a[0], b0[1], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            run_split(0);
            acc[1] = xw_mfma(a[0], b1[1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            run_split(1);
            acc[0] = xw_mfma(a[1], b0[0], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            run_write(nxt, o, 0);
            run_read(o, 1);
            acc[1] = xw_mfma(a[1], b1[0], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = xw_mfma(a[1], b0[1], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (GENX && o == 1) gen_wait(1); else run_wait();
            run_mask(valid, o, 1);
            acc[1] = xw_mfma(a[1], b1[1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            run_split(0);
            acc[0] = xw_mfma(a[0], b0[2], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            run_split(1);
            acc[1] = xw_mfma(a[0], b1[2], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            run_write(nxt, o, 1);
            if (s == 1) {                                                         // the next stage is complete on this wave: meet the others, then its first fragments
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                rd(0, nxt);
            }
            acc[0] = xw_mfma(a[2], b0[0], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            dma(t + 2, o, 0); dma(t + 2, o, 1);                                  // both runs of this operand are read: refill its rows with tile t + 2's
            acc[1] = xw_mfma(a[2], b1[0], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the fragments read ahead for a tile that does not exist)
    // lane (li, lh) holds gW rows n = 128 qn + 32 wn + 8 q + 4 lh + e, column k = 128 qk + 64 kp + 32 x + li
    // The loop issues its LDS-DMA unconditionally (clamped rows past the end), so the newest ones are still in flight here.  A wave must NOT end with
    // vector-memory loads outstanding: the hardware frees its registers at s_endpgm and the late data beats land in whatever wave owns them next
    // (seen as wrong lanes 48..63 in the next kernel to start on the CU -- with two processes on the device that is immediately; r03_x6_notes.txt).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.C = grad_target(g.C); g.colsum = grad_target(g.colsum);            // (this XCD's shard when a pass has them on)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 128 * qn + 32 * wn + 8 * (r >> 2) + 4 * lh + (r & 3), k = 128 * qk + 64 * kp + 32 * x + li;
            unsafeAtomicAdd(g.C + (size_t)n * g.ldc + k, acc[x][r]);
        }
    if (g.colsum && qk == 0)                            // both k-quadrants saw the same dY: one of them adds the bias gradient
        unsafeAtomicAdd(g.colsum + 128 * qn + 64 * ch + lane, bs0);
}

// row ranges = a quarter of the CUs the persistent launches may use (clift_set_cu_reserve leaves CUs to a running all-reduce); the four
// quadrant blocks of a range keep block ids 8 apart (one XCD)
static int xw_ranges() { const int r = clift_persistent_cus() / 4; return r < 1 ? 1 : r; }

int clift_wgrad_x6_launch(const GemmP& p, hipStream_t st) {
    const int nr = xw_ranges();
    const int rpr = cdiv(cdiv(p.K, nr), XW_ROWS) * XW_ROWS;
    k_wgrad_x6<false><<<32 * cdiv(nr, 8), 512, 0, st>>>(p, rpr, nr, XwGen{nullptr, nullptr, 0, nullptr});
    return clift_check_launch("clift_gemm(fp32x6 wgrad)");
}

// Weight gradient of the SECOND layer of an xyz head without the first layer's activation, fp32x6 arithmetic (the fp32x6 counterpart of
// clift_xyz_head_first2_wgrad):  gW1[n][k] += sum_m dH2[m][n] relu(W0[k] . x_m + b0[k]),  gb1[n] += sum_m dH2[m][n]   (tensoRF.py:475-478)
extern "C" int clift_xyz_head_first2_x6_wgrad(const float* dH2, int ldd, const float* W0, int ldw0, const float* b0, const float* x4, int M,
                                              float* gW1, int ldgw1, float* gb1, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)dH2) & 15) == 0 && ldd % 4 == 0 && ldd >= 256 && (((uintptr_t)x4) & 15) == 0 && ldw0 >= 3 && ldgw1 >= 256,
                  "clift_xyz_head_first2_x6_wgrad: dH2 / x4 must be 16-byte aligned, ldd a multiple of 4 and >= 256, ldgw1 >= 256");
    GemmP p = {};
    p.M = 256; p.N = 256; p.K = M; p.A = dH2; p.lda = ldd; p.B = nullptr; p.ldb = 256; p.C = gW1; p.ldc = ldgw1; p.colsum = gb1; p.accumulate = 1;
    const int nr = xw_ranges();
    const int rpr = cdiv(cdiv(M, nr), XW_ROWS) * XW_ROWS;
    k_wgrad_x6<true><<<32 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, XwGen{x4, W0, ldw0, b0});
    return clift_check_launch("clift_xyz_head_first2_x6_wgrad");
}
