// layer_x6.hip -- the 256 -> 256 hidden layers of the xyz heads as PERSISTENT fp32-FAITHFUL kernels on the bf16 matrix cores
// ("fp32x6", clift_gemm precision = 2; forward C = relu(A W^T + b) and masked dgrad C = mask . (A W)).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TFLOP/s) and the exact kernel (layer_f32.hip) sits at 0.77 of it, limited by
// the clock the chip holds under dense fp32 MFMA load.  v_mfma_f32_32x32x16_bf16 is 16x faster.  Every fp32 value splits EXACTLY into
// three bf16 terms x = x1 + x2 + x3 (round-to-nearest each: 8 significant bits apiece, 24 together), so
//     x w = x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1) + O(2^-26 |x w|)
// is six bf16 products, each exact in the MFMA's fp32 accumulator; the dropped terms are below one fp32 rounding of the product.
// (gemm_split.hip is the same arithmetic as a tiled kernel -- load-latency bound.)
//
// Shape of the kernel.  The three bf16 planes of a 256 x 256 weight matrix are 384 KB: they only fit in the register files of TWO CUs.
// So a row range is shared by a PAIR of workgroups (block ids 8 apart = the same XCD, so the second read of an activation row comes from
// that L2), each owning 128 output columns.  A workgroup is EIGHT waves = 4 column groups x 2 k-halves, two per SIMD: wave (c, kh) holds the
// weight planes of columns 32 c .. +31 for k = 128 kh .. +127 (3 x 32 registers), split from the fp32 weights once per workgroup.
// Activation rows (fp32 in memory) travel by LDS-DMA into a 32 KB staging area one tile ahead and are split COOPERATIVELY: per 32-row tile
// each wave reads back four 1 KB rows (lane = 16-byte piece), splits them (22 VALU per piece) and writes three 8-byte pieces into the
// tile's bf16 plane images in LDS (2 stages x 3 planes x 16 KB; 16-byte chunk c of row r in slot c ^ (r & 15): conflict-free for reads
// and writes); every wave then reads ITS k-half of the tile as MFMA fragments.  The two k-halves of an output tile meet through LDS: at the
// end of a tile each wave sums its two accumulator chains, SENDS the half of the 16 values per lane that its partner will finish (2 x 1 KB
// into a double-buffered exchange area) and keeps the other half; after the next barrier it adds what it received, applies bias / ReLU /
// mask and stores its 16 columns.  Symmetric on purpose: both waves of a pair issue the same instruction stream, so the hand-counted waits
// are shared.  ONE s_barrier per tile.  LDS: 96 KB plane images + 32 KB staging + 32 KB exchange = all 160 KB; 204 VGPRs, no AGPRs.
//
// A four-wave form of this kernel (one wave per SIMD, ~400 of the SIMD's 512 registers, no exchange) measured the same time and was REMOVED: with
// two processes sharing the GPU, small kernels of the OTHER process that were scheduled onto the leftover registers beside its MFMA stream
// returned corrupted lanes 48..63.  The mechanism (bisected on the weight-gradient kernel, layer_x6w.hip / profiles/r03_x6_notes.txt) is
// co-residency with a dense bf16 MFMA stream; this form carries the remedy, a register ballast: every wave allocates 256 registers, two fill the
// file.  The four-wave form's fused narrow output layer (clift_xyz_head_last2_x6_fwd, ABI 9) went with it: no LDS left here for its cross-wave sum.
//
// Memory-instruction bookkeeping.  All vector-memory instructions of the loop are issued by hand so that the waits can be COUNTED (the
// compiler's own accounting stops at the loop's back edge and falls back to vmcnt(0), which drained the newest prefetch every tile).
// Nothing in the loop is conditional: rows past the end of a range are CLAMPED to its last row on the way in (DMA, mask), so their
// results are copies of that row's and the store simply writes them to that row again; the first tile "stores" zeros to its own rows,
// which its real results overwrite later (same wave, same addresses, program order).  tools/x6_vmcnt_model.py replays the streams.
#include "gemm_common.h"
CLIFT_ROWS_LIMIT_BINDER(layer_x6)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Timing probes (tools/x6_ablation.sh builds variants of the library with -DX6_ABL=<bits>; results are garbage by construction, never shipped):
// 1 no fragment reads after a tile's first, 2 no activation split (staging reads, VALU, plane writes), 4 no k-half exchange / finish,
// 8 no LDS-DMA and no waits for it, 16 no output stores, 32 no barrier; fixed-cost probes (tools/x6_fixed_probe.py): 64 weights not loaded (synthesised),
// 128 weights not split, 256 no staging of tile 0 in the prologue, 512 no tiles at all, 1024 the kernel returns at once (launch + dispatch only).
#ifndef X6_ABL
#define X6_ABL 0
#endif
// cache-policy probes: bit 1 = the row DMA non-temporal (aux = 2), bit 2 = the output stores non-temporal
#ifndef X6_NT
#define X6_NT 0
#endif
constexpr int X6_ROWS = 32;                         // rows per tile
constexpr int X6_PLANE = X6_ROWS * 512;             // bytes of one bf16 plane image of a tile (32 rows x 256 k x 2 B)
constexpr int X6_STAGE = 3 * X6_PLANE;              // 48 KB

static __device__ __forceinline__ unsigned x6_pk(float lo, float hi) {        // two fp32 -> packed bf16 (RNE), `lo` in the low half
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
static __device__ __forceinline__ float x6_uniform(float v) {                  // a wave-uniform value, kept in a scalar register
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
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
constexpr int X6_RAW = 2 * X6_STAGE;                // fp32 staging: wave w owns rows w + 4 i as 1 KB slots (w * 8 + i), 32 KB

template <int N>
static __device__ __forceinline__ void x6_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int X6_XCH = X6_RAW + X6_ROWS * 1024;     // exchange area: [tile parity][wave][2][lane] float4 = 2 x 16 KB
// vmcnt before the read-back of staged row i (tools/x6_vmcnt_model.py replays each stream and prints these tables), by variant:
#ifndef X6_VM00
#define X6_VM00 5           // (tests/test_abi.py builds the listing once with 6 here: the guard must fail it)
#endif
constexpr int X8_VM[6][4] = {
    {X6_VM00, 4, 3, 3}, // 0 forward:                        D0 D1 S0 D2 S1 D3 per tile
    {5, 5, 5, 5},       // 1 dgrad:                    m0 m1 D0 D1 S0 D2 S1 D3; the masks are needed after 6 younger instructions
    {6, 4, 4, 4},       // 2 forward + output layer:         D0 P D1 S0 D2 S1 D3 (P = the store of the output layer's partial sums)
    {4, 2, 3, 3},       // 3 the same, hidden not written:   D0 P D1 D2 D3
    {3, 3, 3, 3},       // 4 dgrad consumed in-kernel (K3W): p D0 D1 D2 D3 (p = the rows' positions; needed after 4 younger instructions)
    {5, 5, 4, 4},       // 5 dgrad with the mask as sign bytes (BM): b D0 D1 S0 D2 S1 D3 (b = one byte per lane; needed after 6 younger)
};                      // (forward that WRITES sign bytes: D0 B D1 S0 D2 S1 D3 = row 2 with B in P's place)

template <int V>
static __device__ __forceinline__ void x8_wait_piece(int i) {
    if (i == 0) x6_wait_vm<X8_VM[V][0]>();
    if (i == 1) x6_wait_vm<X8_VM[V][1]>();
    if (i == 2) x6_wait_vm<X8_VM[V][2]>();
    if (i == 3) x6_wait_vm<X8_VM[V][3]>();
}

// GEN (forward only): the layer is the SECOND layer of an xyz head and its input is generated, A[m][k] = relu(W0[k] . x_m + b0[k]) with K = 3
// (tensoRF.py:475,576), instead of being written by a separate launch (clift_linear_k3_fwd) and read back.  Nothing of the row pipeline is
// needed then: the 32 KB staging area only carries the POSITIONS (a wave DMAs the four it needs, 64 bytes, two tiles ahead into a parity
// slot), the "read-back" of a staged row becomes one broadcast ds_read_b128 of its position -- the same LDS instruction in the same place, so
// every counted lgkmcnt wait stays as it is -- and the lane computes its four k values (its 12 weights + 4 biases live in registers, 13 VALU)
// where it would have unpacked them.  The activation itself is never written: the backward re-derives it (clift_xyz_head_first2_bwd / _wgrad).
struct X6Gen {
    const float* x4;      // (M, 4) normalised sample positions
    const float* W0;      // (256, 3), row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
};

// OUTV (forward only, 1 = hidden activation written as well, 2 = not written): the layer is the LAST hidden layer of a head whose output layer
// is narrow (E <= 4: the instance heads, tensoRF.py:478-481; the semantic head of a 2-class scene) and that layer is applied to the finished
// tile IN REGISTERS: a lane holds 8 finished columns of one row and the output weights of those columns (32 registers), 32 FMAs give its share
// of the row's E outputs, one v_permlane32_swap pair folds the two half-waves (lower + upper, a fixed order).  There is no LDS left to sum the
// shares of the 8 waves, and the other 128 columns live on another CU: every wave STORES its 32 rows x 4 partial sums (8 bytes per lane) into its
// own slot 8 * half + wave of a 16-slot workspace and a small second launch (k_x6_out_sum) adds the 16 slots of a row in slot order + bias.
// 256 B per row written and read once instead of the 1 KB hidden activation written here and re-streamed by a 256 -> E GEMM: this kernel is
// bound by its output stores (profiles/r03_x6_notes.txt), so what it does not write is what it gains.  A row's bits depend on nothing but the
// row (fixed column partition, fixed order), i.e. not on how many rows share the launch.
struct X6Out {
    const float* Wo;      // (E, 256), row pitch ldwo
    int ldwo, E;
    float* part;          // workspace: [16][pstride] float4
    long pstride;         // rows per slot (>= M)
};

// K3W (dgrad only; the fp32x6 counterpart of k_layer_f32<dgrad, K3W>): the layer is the SECOND layer of an xyz head, so its input gradient
// dH1 = (W0 x + b0 > 0) . (dH2 W1) has one consumer, the K = 3 first layer's weight gradient gW0[n][0..2] += sum_m dH1[m][n] x_m, gb0[n] += sum_m
// dH1[m][n] -- a sum over ROWS, and a lane of this kernel owns one row of every tile (8 finished columns of it).  So nothing is exchanged per
// tile: the lane keeps 8 x 4 running sums over ITS rows for the whole row range (32 registers), re-deriving the ReLU mask from the row's
// position in the forward's operation order (the forward never wrote the activation); the coefficients of the wave's 16 columns live in
// SGPRs (the two half-waves own different columns: both candidates are evaluated and one selected).  dH1 is never written, the mask never read:
// one position per row (16 B) replaces the two mask loads and both stores.  The 32 sums are folded across the lanes once per block, through
// the plane images' LDS after the last tile, and leave as one atomic per thread (into this XCD's gradient shard when a pass has them on).
struct X6K3 {
    const float* x4;      // (M, 4) normalised sample positions
    const float* W0;      // (256, 3), row pitch ldw0
    int ldw0;
    const float* b0;      // (256)
    float* gW0;           // (256, 3), row pitch ldgw0
    int ldgw0;
    float* gb0;           // (256)
};

// BM (sign bytes, g.sign_bits): a forward also writes, a masked dgrad reads INSTEAD of the fp32 mask, ONE byte per lane and tile: bit 4 q + e = the
// sign of the lane's finished column fcol + 8 q + e.  Forward and dgrad of consecutive layers use the same (block half, wave, lane) -> column map
// and tiles are 32-row aligned from row 0, so byte ((row / 32) * 16 + 8 half + wave) * 64 + lane is read by the lane that needs it: one
// coalesced 64-byte access per wave and tile, 32 B per row instead of the 1 KB fp32 mask row.  Bit-identical results (the stored activation
// is the value whose sign was taken).
template <bool DGRAD, bool GEN = false, int OUTV = 0, bool K3W = false, bool BM = false>
__global__ __launch_bounds__(512, 2) void k_layer_x6(GemmP g, int rows_per_range, int nranges, X6Gen gx, X6Out go, X6K3 gk) {
    static_assert(!(DGRAD && GEN), "GEN is a forward form");
    static_assert(!(OUTV && (DGRAD || GEN)), "OUTV is a plain forward form");
    static_assert(!K3W || DGRAD, "K3W is a dgrad form");
    static_assert(!(BM && (OUTV || K3W)), "sign bytes go with the plain / generating forward and the plain dgrad");
    constexpr int VMV = K3W ? 4 : OUTV ? 1 + OUTV : (DGRAD ? (BM ? 5 : 1) : (BM ? 2 : 0));      // row of X8_VM
    __shared__ __attribute__((aligned(1024))) unsigned char lds[X6_XCH + 2 * 16384];                // 160 KB, the only LDS object
    // register ballast: the wave allocates its whole 256-register budget, so that two of them fill the SIMD's file and no wave of another kernel
    // (another PROCESS sharing the device) is scheduled beside this MFMA stream -- see layer_x6w.hip for what happens otherwise
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if (X6_ABL & 1024) return;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cg = wave & 3, kh = wave >> 2;       // column group, k-half (scalars)
    const int b = blockIdx.x, half = (b >> 3) & 1, range = (b & 7) + 8 * (b >> 4);
    if (rows_limited()) {
        g.M = limit_rows(g.M);
        rows_per_range = ((g.M + nranges - 1) / nranges + X6_ROWS - 1) / X6_ROWS * X6_ROWS;
    }
    if (range >= nranges) return;
    const int rbeg = range * rows_per_range, rend = min(g.M, rbeg + rows_per_range);
    if (rbeg >= rend) return;
    const int ntiles = (X6_ABL & 512) ? 0 : (rend - rbeg + X6_ROWS - 1) / X6_ROWS;
    const int ncol = 128 * half + 32 * cg;                   // first output column of this wave's MFMA tile
    const int fcol = ncol + 16 * kh + 4 * lh;                // this lane's first FINISHED column: groups q = 2 kh, 2 kh + 1 -> fcol, fcol + 8
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;

    // ---- cooperative split: this wave provides rows wave + 8 i (i = 0..3) of every tile, lane = 16-byte piece (k = 4 lane .. +3)
    auto dma_piece = [&](int t, int i) {
        if (X6_ABL & 8) return;
        const int gr = min(rbeg + t * X6_ROWS + wave + 8 * i, rend - 1);
        if (i == 0) CLIFT_MARK_DMA("piece0"); if (i == 1) CLIFT_MARK_DMA("piece1"); if (i == 2) CLIFT_MARK_DMA("piece2"); if (i == 3) CLIFT_MARK_DMA("piece3");
        __builtin_amdgcn_global_load_lds(g.A + (size_t)gr * g.lda + 4 * lane, (lds_ptr_t)(lds + X6_RAW + (wave * 4 + i) * 1024), 16, 0, (X6_NT & 1) ? 2 : 0);
    };
    // GEN: positions of this wave's rows wave + 8 i of tile t -> staging slot (t & 1): lanes 0..15 = (row i, component), the other lanes repeat them
    const unsigned posa = lds0 + (unsigned)(X6_RAW + wave * 256);
    auto pos_dma = [&](int t) {
        if (X6_ABL & 8) return;
        const int gr = min(rbeg + t * X6_ROWS + wave + 8 * ((lane >> 2) & 3), rend - 1);
        CLIFT_MARK_DMA("pos");
        __builtin_amdgcn_global_load_lds(gx.x4 + (size_t)gr * 4 + (lane & 3), (lds_ptr_t)(lds + X6_RAW + (t & 1) * 2048 + wave * 256), 4, 0, 0);
    };
    float gw0[4] = {0.f, 0.f, 0.f, 0.f}, gw1[4] = {0.f, 0.f, 0.f, 0.f}, gw2[4] = {0.f, 0.f, 0.f, 0.f}, gbb[4] = {0.f, 0.f, 0.f, 0.f};
    if (GEN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wr0 = gx.W0 + (size_t)(4 * lane + e) * gx.ldw0;
            gw0[e] = wr0[0]; gw1[e] = wr0[1]; gw2[e] = wr0[2]; gbb[e] = gx.b0[4 * lane + e];
        }
    }
    const unsigned rawa = lds0 + (unsigned)(X6_RAW + wave * 4096 + lane * 16);
    // LDS address (stage 0, plane 0) of this lane's 8-byte piece of row wave + 8 i: (r & 15) = wave + 8 (i & 1); rows i and i + 2 differ by 8 KB
    unsigned wofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave + 8 * i;
        wofs[i] = lds0 + (unsigned)(r * 512 + (((lane >> 1) ^ (r & 15)) << 4) + (lane & 1) * 8);
    }
    f32x4 sp_x;
    unsigned sp_h0, sp_h1, sp_m0, sp_m1;
    float sp_r0, sp_r1, sp_r2, sp_r3, sp_s0, sp_s1, sp_s2, sp_s3;
    // `tn` = the tile whose rows are being split (GEN: selects the position slot)
    auto raw_read = [&](int i, int tn) {
        if (X6_ABL & 2) return;
        if (GEN) {
            const unsigned a = posa + (unsigned)((tn & 1) * 2048);
            if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(sp_x) : "v"(a) : "memory");
            if (i == 1) asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(sp_x) : "v"(a) : "memory");
            if (i == 2) asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(sp_x) : "v"(a) : "memory");
            if (i == 3) asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(sp_x) : "v"(a) : "memory");
            return;
        }
        if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(sp_x) : "v"(rawa) : "memory");
        if (i == 3) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(sp_x) : "v"(rawa) : "memory");
    };
    // GEN: sp_x holds the row's position (after the wait that publishes it): replace it by the lane's four generated values
    auto gen_values = [&]() {
        if (!GEN || (X6_ABL & 2)) return;
        const f32x4 p = sp_x;
        sp_x[0] = fmaxf(fmaf(gw2[0], p[2], fmaf(gw1[0], p[1], fmaf(gw0[0], p[0], gbb[0]))), 0.f);      // same order as k_linear_k3_fwd
        sp_x[1] = fmaxf(fmaf(gw2[1], p[2], fmaf(gw1[1], p[1], fmaf(gw0[1], p[0], gbb[1]))), 0.f);
        sp_x[2] = fmaxf(fmaf(gw2[2], p[2], fmaf(gw1[2], p[1], fmaf(gw0[2], p[0], gbb[2]))), 0.f);
        sp_x[3] = fmaxf(fmaf(gw2[3], p[2], fmaf(gw1[3], p[1], fmaf(gw0[3], p[0], gbb[3]))), 0.f);
        asm volatile("" : "+v"(sp_x));
    };
    auto split_a = [&]() {
        if (X6_ABL & 2) return;
        sp_h0 = x6_pk(sp_x[0], sp_x[1]); sp_h1 = x6_pk(sp_x[2], sp_x[3]);
        asm volatile("" : "+v"(sp_h0), "+v"(sp_h1));
    };
    auto split_b0 = [&]() { if (X6_ABL & 2) return; sp_r0 = sp_x[0] - x6_lo(sp_h0); sp_r1 = sp_x[1] - x6_hi(sp_h0); };
    auto split_b1 = [&]() { if (X6_ABL & 2) return; sp_r2 = sp_x[2] - x6_lo(sp_h1); sp_r3 = sp_x[3] - x6_hi(sp_h1); };
    auto wr = [&](unsigned addr, int plane, int i, unsigned a, unsigned c) {
        const u32x2 d = {a, c};
        const int off = plane * X6_PLANE + (i >> 1) * 8192;
        if (off == 0) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(d) : "memory");
        if (off == 8192) asm volatile("ds_write_b64 %0, %1 offset:8192" : : "v"(addr), "v"(d) : "memory");
        if (off == 16384) asm volatile("ds_write_b64 %0, %1 offset:16384" : : "v"(addr), "v"(d) : "memory");
        if (off == 24576) asm volatile("ds_write_b64 %0, %1 offset:24576" : : "v"(addr), "v"(d) : "memory");
        if (off == 32768) asm volatile("ds_write_b64 %0, %1 offset:32768" : : "v"(addr), "v"(d) : "memory");
        if (off == 40960) asm volatile("ds_write_b64 %0, %1 offset:40960" : : "v"(addr), "v"(d) : "memory");
    };
    auto split_c = [&](unsigned stage, int i) {
        if (X6_ABL & 2) return;
        sp_m0 = x6_pk(sp_r0, sp_r1); sp_m1 = x6_pk(sp_r2, sp_r3);
        asm volatile("" : "+v"(sp_m0), "+v"(sp_m1));
        wr(wofs[i & 1] + stage, 0, i, sp_h0, sp_h1);
    };
    auto split_d0 = [&]() { if (X6_ABL & 2) return; sp_s0 = sp_r0 - x6_lo(sp_m0); sp_s1 = sp_r1 - x6_hi(sp_m0); };
    auto split_d1 = [&]() { if (X6_ABL & 2) return; sp_s2 = sp_r2 - x6_lo(sp_m1); sp_s3 = sp_r3 - x6_hi(sp_m1); };
    auto split_e = [&](unsigned stage, int i) {
        if (X6_ABL & 2) return;
        wr(wofs[i & 1] + stage, 1, i, sp_m0, sp_m1);
        wr(wofs[i & 1] + stage, 2, i, x6_pk(sp_s0, sp_s1), x6_pk(sp_s2, sp_s3));
    };

    // ---- prologue: tile 0's rows -> staging -> split into stage 0; tile 1's rows on their way while the weights are prepared
    if (!(X6_ABL & 256)) {
    if (GEN) { pos_dma(0); pos_dma(1); }
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_piece(0, i);
    }
    x6_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        raw_read(i, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sp_x) : : "memory");
        gen_values();
        split_a(); split_b0(); split_b1(); split_c(0u, i); split_d0(); split_d1(); split_e(0u, i);
    }
    }
    {   // what tile 0 "receives" (exchange parity 1, this wave's slot): zeros
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const unsigned a = lds0 + (unsigned)(X6_XCH + 16384 + wave * 2048 + lane * 16);
        asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(z) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:1024" : : "v"(a), "v"(z) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!GEN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_piece(1, i);
    }

    // ---- weight fragments of this wave's k-half: w?[j] = planes of W(n = ncol + li, k = 128 kh + 16 j + 8 lh .. +7)
    u32x4 wh[8], wm[8], wl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v[8];
        if (X6_ABL & 64) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)(lane + 3 * e + 7 * j) * 0.001f;
        } else if (!DGRAD) {
            const float* q = g.B + (size_t)(ncol + li) * g.ldb + 128 * kh + 16 * j + 8 * lh;
            const float4 a = *reinterpret_cast<const float4*>(q), c = *reinterpret_cast<const float4*>(q + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        } else {
            const float* q = g.B + (size_t)(128 * kh + 16 * j + 8 * lh) * g.ldb + ncol + li;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = q[(size_t)e * g.ldb];
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            unsigned h, m, l;
            if (X6_ABL & 128) { h = __float_as_uint(v[2 * pr]); m = __float_as_uint(v[2 * pr + 1]); l = h ^ m; }
            else x6_split_pair(v[2 * pr], v[2 * pr + 1], h, m, l);
            wh[j][pr] = h; wm[j][pr] = m; wl[j][pr] = l;
        }
    }
    // bias of the 8 columns this lane FINISHES (fcol + 0..3, fcol + 8 + 0..3)
    float bfin[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bfin[e] = (!DGRAD && g.bias) ? g.bias[fcol + 8 * (e >> 2) + (e & 3)] : 0.f;
    x6_wait_vm<0>();                     // tile 1's rows have landed: from here on every wait is counted

    const unsigned adk = lds0 + (unsigned)(li * 512 + ((lh ^ (li & 15)) << 4) + kh * 256);          // chunk 16 kh + 2 j + lh of row li
    const unsigned xrd = lds0 + (unsigned)(X6_XCH + wave * 2048 + lane * 16);                       // what the partner sends to this wave
    const unsigned xwr = lds0 + (unsigned)(X6_XCH + (wave ^ 4) * 2048 + lane * 16);                 // what this wave sends to its partner
    f32x4 prev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 keep[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, recv[2];
    f32x4 mk[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    unsigned mbyte = 0xffu, sbyte = 0;                                       // BM: the tile's sign byte of this lane (dgrad: read; forward: to write)
    unsigned char* const sb0 = BM ? g.sign_bits + (size_t)(8 * half + wave) * 64 + lane : nullptr;
    int tb_prev = rbeg / X6_ROWS, tb_done = tb_prev;                         // tile (global index) of `prev` / of the tile just multiplied
    int prev_m = min(rbeg + li, rend - 1), m_done = prev_m;                                          // (tile 0 "stores" zeros to its own rows first)
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // sum of the two chains; the half the partner finishes goes to the exchange area (parity par), the other half stays in `keep`
    auto send = [&](int par) {
        if (X6_ABL & 4) return;
        f32x4 sendv[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lo8 = acc0[e] + acc1[e], hi8 = acc0[8 + e] + acc1[8 + e];                   // registers 0..7 <-> q = 0, 1; 8..15 <-> q = 2, 3
            keep[e >> 2][e & 3] = kh ? hi8 : lo8;
            sendv[e >> 2][e & 3] = kh ? lo8 : hi8;
        }
        const unsigned a = xwr + (unsigned)(par * 16384);
        asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(sendv[0]) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:1024" : : "v"(a), "v"(sendv[1]) : "memory");
    };
    auto finish = [&]() {                // keep + received + bias, activation / mask -> prev
        if (X6_ABL & 4) return;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o = (keep[q][e] + recv[q][e]) + bfin[4 * q + e];
                if (K3W) { }                                                      // (masked where it is consumed: k3_col)
                else if (DGRAD && BM) o = ((mbyte >> (4 * q + e)) & 1u) ? o : 0.f;
                else if (DGRAD) o = mk[q][e] > 0.f ? o : 0.f;
                else if (g.act == 1) o = fmaxf(o, 0.f);
                prev[q][e] = o;
            }
    };

    // K3W: coefficients of the wave's 16 columns (wave-uniform addresses: scalar loads, SGPRs), this lane's running sums, positions
    float sw[16][4];
    float kacc[8][4];
    f32x4 xq = {0.f, 0.f, 0.f, 0.f}, xcur = {0.f, 0.f, 0.f, 0.f};
    bool ok_done = false, prev_ok = false;                   // whether the lane's row of the tile is a real row (not a clamped copy of the last one)
    if (K3W) {
        const int colbase = 128 * half + 32 * cg + 16 * kh;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* wr0 = gk.W0 + (size_t)(colbase + i) * gk.ldw0;
            sw[i][0] = x6_uniform(wr0[0]); sw[i][1] = x6_uniform(wr0[1]); sw[i][2] = x6_uniform(wr0[2]); sw[i][3] = x6_uniform(gk.b0[colbase + i]);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) { kacc[c][0] = 0.f; kacc[c][1] = 0.f; kacc[c][2] = 0.f; kacc[c][3] = 0.f; }
    }
    // finished column c = 4 q + e of this lane is column 8 q + e (+ 4 for the upper half-wave) of the wave's 16
    auto k3_col = [&](int c) {
        const int q = c >> 2, e = c & 3, ia = 8 * q + e, ib = ia + 4;
        const float pa = fmaf(sw[ia][2], xcur[2], fmaf(sw[ia][1], xcur[1], fmaf(sw[ia][0], xcur[0], sw[ia][3])));      // the forward's order
        const float pb = fmaf(sw[ib][2], xcur[2], fmaf(sw[ib][1], xcur[1], fmaf(sw[ib][0], xcur[0], sw[ib][3])));
        const float pre = lh ? pb : pa;
        const float md = (pre > 0.f && prev_ok) ? prev[q][e] : 0.f;
        kacc[c][0] = fmaf(md, xcur[0], kacc[c][0]);
        kacc[c][1] = fmaf(md, xcur[1], kacc[c][1]);
        kacc[c][2] = fmaf(md, xcur[2], kacc[c][2]);
        kacc[c][3] += md;
        asm volatile("" : "+v"(kacc[c][0]), "+v"(kacc[c][1]), "+v"(kacc[c][2]), "+v"(kacc[c][3]));      // (pinned to its gap)
    };

    // OUTV: output weights of this lane's 8 finished columns (rows e >= E are zero), its share of the row's outputs, the half-wave fold
    float wo[4][8], po[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2 pfold = {0.f, 0.f};
    if (OUTV) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 8; ++c) wo[e][c] = e < go.E ? go.Wo[(size_t)e * go.ldwo + fcol + 8 * (c >> 2) + (c & 3)] : 0.f;
    }
    auto out_fma = [&](int e) {          // po[e] = sum over the lane's 8 columns, ascending, one FMA chain
        float a = prev[0][0] * wo[e][0];
        a = fmaf(prev[0][1], wo[e][1], a); a = fmaf(prev[0][2], wo[e][2], a); a = fmaf(prev[0][3], wo[e][3], a);
        a = fmaf(prev[1][0], wo[e][4], a); a = fmaf(prev[1][1], wo[e][5], a); a = fmaf(prev[1][2], wo[e][6], a); a = fmaf(prev[1][3], wo[e][7], a);
        asm volatile("" : "+v"(a));      // (pinned to its gap)
        po[e] = a;
    };
    auto out_fold = [&]() {              // lanes of the lower half end up with outputs 0, 1 of their row, the upper half with 2, 3: (lower + upper) each
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(po[0]), __float_as_uint(po[2]), false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(po[1]), __float_as_uint(po[3]), false, false);
        pfold[0] = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
        pfold[1] = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
    };
    float* const pslot = OUTV ? go.part + ((size_t)(8 * half + wave) * go.pstride) * 4 + 2 * lh : nullptr;

    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // this wave's plane and exchange writes are done ...
        if (!(X6_ABL & 32)) __builtin_amdgcn_s_barrier();                        // ... and everyone's; everyone is done reading the other stage
        asm volatile("" ::: "memory");
        const unsigned cur = (unsigned)((t & 1) * X6_STAGE), nxt = (unsigned)(((t + 1) & 1) * X6_STAGE);
        const int m = min(rbeg + t * X6_ROWS + li, rend - 1);
        u32x4 fa[2][3];
        auto rd = [&](int j, u32x4 (&f)[3]) {
            const unsigned a = (adk + cur) ^ (unsigned)(32 * j);
            asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(f[1]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(f[2]) : "v"(a) : "memory");
        };
        rd(0, fa[0]);
        if (!(X6_ABL & 4)) {   // the partner's share of the previous tile (parity (t - 1) & 1 = (t + 1) & 1); tile 0 reads the zeros written in the prologue
            const unsigned a = xrd + (unsigned)(((t + 1) & 1) * 16384);
            asm volatile("ds_read_b128 %0, %1" : "=v"(recv[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(recv[1]) : "v"(a) : "memory");
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u32x4 (&f)[3] = fa[j & 1];
            const int i = j >> 1;
            const bool even = (j & 1) == 0;
            // LDS instructions issued since this step's fragment reads (x6_vmcnt_model.py): j = 0: the two exchange reads; j = 1: the staging
            // read; other even steps: one plane write; other odd steps: the staging read and two plane writes
            if (j == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            else if (even || j == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 0
            if (j + 1 < 8 && !(X6_ABL & 1)) rd(j + 1, fa[(j + 1) & 1]);
            if (even) {
                // GEN: the positions of tile t+1 left during tile t-1 (one DMA, before that tile's two stores); younger than it at any of the four
                // read-backs: those two stores, this tile's position DMA (from i = 2 on) and first store (i = 3) -- vmcnt(2) covers all four
                if (X6_ABL & 8) { } else if (GEN) x6_wait_vm<BM ? 3 : 2>(); else x8_wait_piece<VMV>(i);
                // (marks for the build's listing check: the staged slot read next must not have its DMA in flight -- GEN: the positions of
                //  tile t + 2 may be, from i = 2 on)
                if (GEN) { if (i < 2) CLIFT_MARK_USE("pos", "0"); else CLIFT_MARK_USE("pos", "1"); }
                else { if (i == 0) CLIFT_MARK_USE("piece0", "0"); if (i == 1) CLIFT_MARK_USE("piece1", "0"); if (i == 2) CLIFT_MARK_USE("piece2", "0"); if (i == 3) CLIFT_MARK_USE("piece3", "0"); }
                raw_read(i, t + 1);
            } else {
                if (j == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(sp_x) : : "memory");
                else if (j == 7) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(sp_x) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(sp_x) : : "memory");
                gen_values();
                split_a();
            }
            acc0 = x6_mfma(wh[j], f[0], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 1
            if (j == 0) {                                                        // the previous tile's results: what was kept + what the partner sent
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(recv[0]), "+v"(recv[1]) : : "memory");
                finish();
                prev_m = m_done;
                m_done = m;
                if (BM) { tb_prev = tb_done; tb_done = rbeg / X6_ROWS + t; }
                if (K3W) { xcur = xq; prev_ok = ok_done; ok_done = rbeg + t * X6_ROWS + li < rend; }
            } else if (even) split_e(nxt, i - 1);
            else split_b0();
            acc1 = x6_mfma(wh[j], f[1], acc1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 2
            if (even) {
                if (DGRAD && !K3W && !BM && i < 2) {                             // the ReLU mask of the columns this lane finishes: steps 0, 2
                    const float* mp = g.mask + (size_t)m * g.ldmask + fcol + 8 * i;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(mk[i]) : "v"(mp) : "memory");
                }
                if (DGRAD && BM && j == 0) {                                     // "b": ... or its sign byte of THIS tile (consumed next tile)
                    const unsigned char* bp = sb0 + (size_t)(rbeg / X6_ROWS + t) * 1024;
                    asm volatile("global_load_ubyte %0, %1, off" : "=v"(mbyte) : "v"(bp) : "memory");
                }
                if (!DGRAD && BM && j == 0) {                                    // signs of the previous tile's finished values
                    sbyte = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sbyte |= prev[e >> 2][e & 3] > 0.f ? (1u << e) : 0u;
                    asm volatile("" : "+v"(sbyte));
                }
                if (K3W && j == 0) {                                             // "p": the position of this lane's row of THIS tile (consumed next tile)
                    const float* pp = gk.x4 + (size_t)m * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq) : "v"(pp) : "memory");
                }
                if (K3W && j == 2) k3_col(3);
                if (K3W && j == 4) k3_col(6);
                if (OUTV && j == 0) out_fma(0);                                  // the previous tile's rows through the output layer: steps 0, 2
                if (OUTV && j == 2) out_fold();
            } else split_b1();
            acc0 = x6_mfma(wm[j], f[1], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 3
            if (even) {
                if (GEN) { if (i == 1) pos_dma(t + 2); }
                else if (i > 0) dma_piece(t + 2, i - 1);
                if (OUTV && j == 0) out_fma(1);
                if (K3W && j == 0) k3_col(0);
            } else split_c(nxt, i);
            acc1 = x6_mfma(wm[j], f[0], acc1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 4
            if (even) {
                if (OUTV != 2 && !K3W && !(X6_ABL & 16) && i >= 2)
                    { if (X6_NT & 2) __builtin_nontemporal_store(prev[i - 2], reinterpret_cast<f32x4*>(g.C + (size_t)prev_m * g.ldc + fcol + 8 * (i - 2)));
                      else *reinterpret_cast<f32x4*>(g.C + (size_t)prev_m * g.ldc + fcol + 8 * (i - 2)) = prev[i - 2]; }      // steps 4, 6
                if (K3W && j == 0) k3_col(1);
                if (K3W && j == 2) k3_col(4);
                if (K3W && j == 4) k3_col(7);
                if (OUTV && j == 0) out_fma(2);
                if (!DGRAD && BM && j == 2) sb0[(size_t)tb_prev * 1024] = (unsigned char)sbyte;                                              // "B"
                if (OUTV && j == 2 && !(X6_ABL & 16)) *reinterpret_cast<f32x2*>(pslot + (size_t)prev_m * 4) = pfold;       // "P"
            } else { split_d0(); if (j == 7) split_d1(); }
            acc0 = x6_mfma(wh[j], f[2], acc0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- gap 5
            if (!even) {
                if (j < 7) split_d1();
                else { split_e(nxt, 3); if (!GEN) dma_piece(t + 2, 3); }
            } else {
                if (OUTV && j == 0) out_fma(3);
                if (K3W && j == 0) k3_col(2);
                if (K3W && j == 2) k3_col(5);
            }
            acc1 = x6_mfma(wl[j], f[0], acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (K3W) asm volatile("s_waitcnt vmcnt(4)" : "+v"(xq) : : "memory");
        else if (DGRAD && BM) asm volatile("s_waitcnt vmcnt(6)" : "+v"(mbyte) : : "memory");
        else if (DGRAD) asm volatile("s_waitcnt vmcnt(6)" : "+v"(mk[0]), "+v"(mk[1]) : : "memory");
        send(t & 1);
    }
    // the last tile: exchange once more, finish, store
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const unsigned a = xrd + (unsigned)(((ntiles + 1) & 1) * 16384);
        asm volatile("ds_read_b128 %0, %1" : "=v"(recv[0]) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(recv[1]) : "v"(a) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(recv[0]), "+v"(recv[1]) : : "memory");
    }
    // (the loop's stores wrote the results of tiles 0 .. ntiles - 2 during tiles 1 .. ntiles - 1; prev now holds nothing unsaved)
    // The DMAs of the two tiles past the end (clamped rows) were issued unconditionally: a wave must not end with them in flight (see layer_x6w.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    finish();
    if (K3W) {
        xcur = xq; prev_ok = ok_done;
#pragma unroll
        for (int c = 0; c < 8; ++c) k3_col(c);
        // fold the 32 sums across the 32 lanes of each half-wave: through LDS (the plane images: every wave is past its last fragment read --
        // the barrier above -- and the DMAs in flight target the staging area, not this), rows padded to 65 words (conflict-free both ways)
        float* const red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int a = 0; a < 32; ++a) red[(wave * 32 + a) * 65 + lane] = kacc[a >> 2][a & 3];
        __syncthreads();
        const int rw = tid >> 6, rr = tid & 63, hx = rr >> 5, a = rr & 31;
        const float* src = red + (rw * 32 + a) * 65 + hx * 32;
        float ssum = 0.f;
#pragma unroll
        for (int l = 0; l < 32; ++l) ssum += src[l];
        const int c = a >> 2, k = a & 3;
        const int n = 128 * half + 32 * (rw & 3) + 16 * (rw >> 2) + 4 * hx + 8 * (c >> 2) + (c & 3);
        if (k < 3) unsafeAtomicAdd(grad_target(gk.gW0) + (size_t)n * gk.ldgw0 + k, ssum);
        else unsafeAtomicAdd(grad_target(gk.gb0) + n, ssum);
        return;
    }
    if (OUTV != 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(g.C + (size_t)m_done * g.ldc + fcol + 8 * q) = prev[q];
    }
    if (!DGRAD && BM) {
        sbyte = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) sbyte |= prev[e >> 2][e & 3] > 0.f ? (1u << e) : 0u;
        sb0[(size_t)tb_done * 1024] = (unsigned char)sbyte;
    }
    if (OUTV) {
        out_fma(0); out_fma(1); out_fma(2); out_fma(3);
        out_fold();
        *reinterpret_cast<f32x2*>(pslot + (size_t)m_done * 4) = pfold;
    }
}

// out[m][0..E) = bias + the 16 slot partial sums of row m, added in slot order (second launch of the OUTV form)
__global__ __launch_bounds__(256) void k_x6_out_sum(const f32x4* __restrict__ part, long pstride, int M, const float* __restrict__ bo, int E,
                                                    float* __restrict__ out, int ldo) {
    if (rows_limited()) M = limit_rows(M);
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        f32x4 s = part[m];
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            const f32x4 v = part[(size_t)p * pstride + m];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        for (int e = 0; e < E; ++e) out[(size_t)m * ldo + e] = s[e] + bo[e];
    }
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
    const X6Gen none = {nullptr, nullptr, 0, nullptr};
    const X6Out no_out = {nullptr, 0, 0, nullptr, 0};
    if (p.sign_bits) {          // (gemm.hip checked the form: forward writes the sign bytes, dgrad with mask = NULL reads them)
        if (b_trans) k_layer_x6<true, false, 0, false, true><<<grid, 512, 0, st>>>(p, rpr, nr, none, no_out, X6K3{});
        else k_layer_x6<false, false, 0, false, true><<<grid, 512, 0, st>>>(p, rpr, nr, none, no_out, X6K3{});
        return clift_check_launch("clift_gemm(fp32x6 layer, sign bytes)");
    }
    if (b_trans) k_layer_x6<true><<<grid, 512, 0, st>>>(p, rpr, nr, none, no_out, X6K3{});
    else k_layer_x6<false><<<grid, 512, 0, st>>>(p, rpr, nr, none, no_out, X6K3{});
    return clift_check_launch("clift_gemm(fp32x6 layer)");
}

// First TWO layers of an xyz head in one launch, fp32x6 arithmetic for the 256 x 256 layer (the K = 3 layer is exact fp32 FMAs as in
// clift_linear_k3_fwd): h2 = relu(W1 relu(W0 x + b0) + b1)   (tensoRF.py:475-478, 576-579).  The first layer's activation is not written.
// bytes of the sign-byte record of an M-row activation (clift_gemm_t::sign_bits): 16 waves x 64 lanes per 32-row tile
extern "C" long clift_sign_bits_bytes(int M) { return M > 0 ? (long)cdiv(M, X6_ROWS) * 1024 : 0; }

extern "C" int clift_xyz_head_first2_x6_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1, const float* b1,
                                            int M, float* h2, int ldh2, void* sign_bits, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)x4) & 15) == 0 && (((uintptr_t)W1) & 15) == 0 && (((uintptr_t)h2) & 15) == 0 && ldw1 % 4 == 0 && ldw1 >= 256 &&
                      ldh2 % 4 == 0 && ldh2 >= 256 && ldw0 >= 3,
                  "clift_xyz_head_first2_x6_fwd: x4 / W1 / h2 must be 16-byte aligned, pitches >= 256 and multiples of 4");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = nullptr; p.lda = 256; p.B = W1; p.ldb = ldw1; p.C = h2; p.ldc = ldh2; p.bias = b1; p.act = 1;
    const int tiles = cdiv(M, X6_ROWS);
    const int pairs = clift_persistent_cus() / 2;
    const int nranges = tiles < pairs ? tiles : pairs;
    const int rpr = cdiv(cdiv(M, nranges), X6_ROWS) * X6_ROWS;
    const int nr = cdiv(M, rpr);
    p.sign_bits = (unsigned char*)sign_bits;
    if (sign_bits) k_layer_x6<false, true, 0, false, true><<<16 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, X6Gen{x4, W0, ldw0, b0}, X6Out{nullptr, 0, 0, nullptr, 0}, X6K3{});
    else k_layer_x6<false, true><<<16 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, X6Gen{x4, W0, ldw0, b0}, X6Out{nullptr, 0, 0, nullptr, 0}, X6K3{});
    return clift_check_launch("clift_xyz_head_first2_x6_fwd");
}

// LAST hidden layer of an xyz head together with its narrow output layer (E <= 4), fp32x6 arithmetic for the 256 x 256 layer, exact fp32
// FMAs for the output layer (the fp32x6 counterpart of clift_xyz_head_last2_fwd; tensoRF.py:478-481, 591-594 at C <= 4):
//   h = relu(W A^T + b) (written to `hidden` only if it is non-null), out[:, 0:E] = h Wout^T + bout.
// workspace: clift_xyz_head_last2_x6_workspace_bytes(M) bytes of device memory, 16-byte aligned, private to this call until it has run.
extern "C" long clift_xyz_head_last2_x6_workspace_bytes(int M) { return M > 0 ? 256L * M : 0; }

extern "C" int clift_xyz_head_last2_x6_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                           const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, void* workspace,
                                           long workspace_bytes, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE(E >= 1 && E <= 4, "clift_xyz_head_last2_x6_fwd: E must be in [1,4] (got %d)", E);
    CLIFT_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= 256 && ldw >= 256,
                  "clift_xyz_head_last2_x6_fwd: A / W must be 16-byte aligned with pitches >= 256 that are multiples of 4");
    CLIFT_REQUIRE(hidden == nullptr || ((((uintptr_t)hidden) & 15) == 0 && ldh % 4 == 0 && ldh >= 256), "clift_xyz_head_last2_x6_fwd: hidden must be 16-byte aligned, pitch >= 256");
    CLIFT_REQUIRE(workspace != nullptr && (((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= clift_xyz_head_last2_x6_workspace_bytes(M),
                  "clift_xyz_head_last2_x6_fwd: needs a 16-byte aligned workspace of clift_xyz_head_last2_x6_workspace_bytes(M) bytes");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = A; p.lda = lda; p.B = W; p.ldb = ldw; p.C = hidden; p.ldc = ldh; p.bias = b; p.act = 1;
    const int tiles = cdiv(M, X6_ROWS);
    const int pairs = clift_persistent_cus() / 2;
    const int nranges = tiles < pairs ? tiles : pairs;
    const int rpr = cdiv(cdiv(M, nranges), X6_ROWS) * X6_ROWS;
    const int nr = cdiv(M, rpr);
    const X6Gen none = {nullptr, nullptr, 0, nullptr};
    const X6Out op = {Wout, ldwo, E, (float*)workspace, (long)M};
    if (hidden) k_layer_x6<false, false, 1><<<16 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, none, op, X6K3{});
    else k_layer_x6<false, false, 2><<<16 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, none, op, X6K3{});
    int rc = clift_check_launch("clift_xyz_head_last2_x6_fwd");
    if (rc) return rc;
    const int blocks = cdiv(M, 256) < 2048 ? cdiv(M, 256) : 2048;
    k_x6_out_sum<<<blocks, 256, 0, as_stream(s)>>>((const f32x4*)workspace, (long)M, M, bout, E, out, ldo);
    return clift_check_launch("clift_xyz_head_last2_x6_fwd(sum)");
}

// Backward of the first TWO layers of an xyz head in one launch, fp32x6 arithmetic for the 256 x 256 product (the fp32x6 counterpart of
// clift_xyz_head_first2_bwd; the forward is clift_xyz_head_first2_x6_fwd): with dH2 (M, ldd) the gradient at the second layer's output
// (already masked by its ReLU), dH1 = (W0 x + b0 > 0) . (dH2 W1) is formed tile by tile and consumed on the spot:
//   gW0[n][0..2] += sum_m dH1[m][n] x4[m][0..2],   gb0[n] += sum_m dH1[m][n]          (tensoRF.py:475-476, 576-577)
extern "C" int clift_xyz_head_first2_x6_bwd(const float* dH2, int ldd, const float* W1, int ldw1, const float* W0, int ldw0, const float* b0,
                                            const float* x4, int M, float* gW0, int ldgw0, float* gb0, clift_stream_t s) {
    if (M <= 0) return 0;
    CLIFT_REQUIRE((((uintptr_t)dH2) & 15) == 0 && (((uintptr_t)x4) & 15) == 0 && ldd % 4 == 0 && ldd >= 256 && ldw1 >= 256 && ldw0 >= 3 && ldgw0 >= 3,
                  "clift_xyz_head_first2_x6_bwd: dH2 / x4 must be 16-byte aligned, ldd a multiple of 4 and >= 256");
    GemmP p = {};
    p.M = M; p.N = 256; p.K = 256; p.A = dH2; p.lda = ldd; p.B = W1; p.ldb = ldw1; p.C = nullptr; p.ldc = 256;
    const int tiles = cdiv(M, X6_ROWS);
    const int pairs = clift_persistent_cus() / 2;
    const int nranges = tiles < pairs ? tiles : pairs;
    const int rpr = cdiv(cdiv(M, nranges), X6_ROWS) * X6_ROWS;
    const int nr = cdiv(M, rpr);
    k_layer_x6<true, false, 0, true><<<16 * cdiv(nr, 8), 512, 0, as_stream(s)>>>(p, rpr, nr, X6Gen{nullptr, nullptr, 0, nullptr}, X6Out{nullptr, 0, 0, nullptr, 0},
                                                                                  X6K3{x4, W0, ldw0, b0, gW0, ldgw0, gb0});
    return clift_check_launch("clift_xyz_head_first2_x6_bwd");
}
