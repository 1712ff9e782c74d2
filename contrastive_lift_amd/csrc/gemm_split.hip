// gemm_split.hip -- fp32-FAITHFUL nn.Linear forward / dgrad on the bf16 matrix cores ("fp32x6", clift_gemm precision = 2).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at the vector rate (157 TFLOP/s) but v_mfma_f32_32x32x16_bf16 16x faster.  Every fp32
// value splits EXACTLY into three bf16 terms a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 8
// significant bits each, 24 together = the fp32 significand), so
//     a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a b|)
// -- six bf16 products, each exact in the fp32 accumulator of the MFMA; the three dropped terms are below fp32 product
// rounding.  Six MFMAs cost 6/16 of the fp32-MFMA time; the result matches the exact-fp32 kernel to fp32 round-off
// (tests/test_gpu_parity.py compares both against fp64 with the same tolerance).
//
// Operands: A (activations / output gradients, M x K, streamed, row-major) is loaded as fp32 and split on its way INTO LDS
// (once per element: 5.5 VALU ops per value); B (the weight matrix, N x K, reused by every block) is split ONCE per launch by
// k_split_weights into three bf16 planes [3][Np][Kp] (k-contiguous whatever the storage order of B, zero padded), which
// blocks copy into LDS unchanged.  Both operands therefore sit in LDS as three bf16 planes [plane][row][16 k] with a 40-byte
// row pitch, and a fragment is two conflict-free ds_read_b64 per plane.  (A first version kept A as fp32 in LDS and split it at
// fragment time: 4x redundant VALU work across the waves of a block row plus 32 spilled VGPRs made it slower than the exact
// fp32 kernel.)  The wgrad form (both operands streamed, k-major) stays on the exact-fp32 kernel.
// LDS per block (BK = 16): 3 x (128 + 256) x 40 B = 45 KB.
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BKS = 16;       // k depth of one staged tile = one bf16 MFMA k-step
constexpr int PBD = 10;       // dword row pitch of a plane image (16 bf16 + 4 pad = 40 B: conflict-free ds_read_b64)

static __device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo; p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}

// ---------------------------------------------------------------------------------------------- weight split (once per launch)
// planes[p][n][k], p = 0 (leading) .. 2, n < Np, k < Kp; B(n,k) = b_trans ? B[k*ldb+n] : B[n*ldb+k]; zero outside N x K.
__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ B, int ldb, int b_trans, int N, int K, int Np, int Kp,
                                                        unsigned short* __restrict__ planes) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)Np * Kp) return;
    const int n = (int)(gid / Kp), k = (int)(gid - (long)n * Kp);
    float v = 0.f;
    if (n < N && k < K) v = b_trans ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    const size_t plane = (size_t)Np * Kp;
    planes[gid] = __builtin_bit_cast(unsigned short, h);
    planes[plane + gid] = __builtin_bit_cast(unsigned short, m);
    planes[2 * plane + gid] = __builtin_bit_cast(unsigned short, l);
}

struct SplitP {
    GemmP g;
    const unsigned short* planes;   // [3][Np][Kp]
    int Np, Kp;
};

template <int BM, int BN, int WM, int WN, bool SWAP>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8 ? 4 : 1)) void k_gemm_split(SplitP sp) {
    const GemmP& g = sp.g;
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LAP = BM * PBD, LBP = BN * PBD;                 // dwords per A / B plane image
    __shared__ __attribute__((aligned(16))) unsigned lds_a[3 * LAP];
    __shared__ __attribute__((aligned(16))) unsigned lds_b[3 * LBP];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int ntn = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging registers: A = NVA float4 (4 consecutive k of one row, fp32), B = NVB uint4 (8 bf16 of one plane row)
    constexpr int NEA = BM * BKS / 4, NVA = (NEA + NT - 1) / NT;
    constexpr int NEB = 3 * BN * 2, NVB = (NEB + NT - 1) / NT;  // 2 sixteen-byte chunks per row per plane
    float4 va0[NVA];                    // next A tile (fp32), one tile ahead
    uint4 vb[NVB];                      // next B tile (pre-split planes), one tile ahead
    // (prefetching A two tiles ahead -- a second register set and a 2x unrolled loop -- spilled 36 VGPRs into the loop: 509 us
    //  instead of 319 us for the 265 k x 256 x 256 forward layer)
    const size_t plane_elems = (size_t)sp.Np * sp.Kp;

    auto load_a = [&](int k0, float4 (&va)[NVA]) {
        const bool full = (k0 + BKS <= g.K) && (m0 + BM <= g.M);
#pragma unroll
        for (int p = 0; p < NVA; ++p) {
            const int e = tid + p * NT;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NEA % NT == 0 || e < NEA) {
                const int m = m0 + (e >> 2), k = k0 + (e & 3) * 4;
                const float* q = g.A + (size_t)m * g.lda + k;
                if (full) x = *reinterpret_cast<const float4*>(q);
                else if (m < g.M) {
                    if (k + 3 < g.K) x = *reinterpret_cast<const float4*>(q);
                    else { if (k < g.K) x.x = q[0]; if (k + 1 < g.K) x.y = q[1]; if (k + 2 < g.K) x.z = q[2]; }
                }
            }
            va[p] = x;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NVB; ++p) {
            const int e = tid + p * NT;
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (NEB % NT == 0 || e < NEB) {
                const int pl = e / (2 * BN), r = (e - pl * 2 * BN) >> 1, c = e & 1;        // plane, row, 8-k chunk
                // planes are zero padded to Np >= n0 + BN rows and Kp >= k0 + BKS columns: no bounds checks
                x = *reinterpret_cast<const uint4*>(sp.planes + pl * plane_elems + (size_t)(n0 + r) * sp.Kp + k0 + c * 8);
            }
            vb[p] = x;
        }
    };
    auto store_tiles = [&](const float4 (&va)[NVA]) {
#pragma unroll
        for (int p = 0; p < NVA; ++p) {
            const int e = tid + p * NT;
            if (NEA % NT == 0 || e < NEA) {
                // exact three-way split of 4 consecutive k, 8 bytes into each plane
                const float a0 = va[p].x, a1 = va[p].y, a2 = va[p].z, a3 = va[p].w;
                const unsigned h0 = pack_bf16(a0, a1), h1 = pack_bf16(a2, a3);
                const float r0 = a0 - __uint_as_float(h0 << 16), r1 = a1 - __uint_as_float(h0 & 0xffff0000u);
                const float r2 = a2 - __uint_as_float(h1 << 16), r3 = a3 - __uint_as_float(h1 & 0xffff0000u);
                const unsigned m0_ = pack_bf16(r0, r1), m1_ = pack_bf16(r2, r3);
                const float s0 = r0 - __uint_as_float(m0_ << 16), s1 = r1 - __uint_as_float(m0_ & 0xffff0000u);
                const float s2 = r2 - __uint_as_float(m1_ << 16), s3 = r3 - __uint_as_float(m1_ & 0xffff0000u);
                unsigned* q = lds_a + (e >> 2) * PBD + (e & 3) * 2;
                *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(q + LAP) = make_uint2(m0_, m1_);
                *reinterpret_cast<uint2*>(q + 2 * LAP) = make_uint2(pack_bf16(s0, s1), pack_bf16(s2, s3));
            }
        }
#pragma unroll
        for (int p = 0; p < NVB; ++p) {
            const int e = tid + p * NT;
            if (NEB % NT == 0 || e < NEB) {
                const int pl = e / (2 * BN), r = (e - pl * 2 * BN) >> 1, c = e & 1;
                uint2* q = reinterpret_cast<uint2*>(lds_b + pl * LBP + r * PBD + c * 4);     // 8-byte aligned (pitch 40 B)
                q[0] = make_uint2(vb[p].x, vb[p].y);
                q[1] = make_uint2(vb[p].z, vb[p].w);
            }
        }
    };
    auto frag = [&](const unsigned* img, int row) {
        const uint2* q2 = reinterpret_cast<const uint2*>(img + row * PBD + 4 * lh);
        const uint2 lo = q2[0], hi = q2[1];
        return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
    };

    const int arow = wm * (BM / WM) + li, brow = wn * (BN / WN) + li;
    auto step = [&](int k0, float4 (&va)[NVA]) {
        store_tiles(va);
        __syncthreads();
        if (k0 + BKS < g.K) { load_a(k0 + BKS, va); load_b(k0 + BKS); }
#pragma unroll
        for (int x = 0; x < TM; ++x) {
            const bf16x8 ah = frag(lds_a, arow + x * 32), am = frag(lds_a + LAP, arow + x * 32), al = frag(lds_a + 2 * LAP, arow + x * 32);
#pragma unroll
            for (int y = 0; y < TN; ++y) {
                const bf16x8 bh = frag(lds_b, brow + y * 32), bm = frag(lds_b + LBP, brow + y * 32), bl = frag(lds_b + 2 * LBP, brow + y * 32);
                f32x16 c = acc[x][y];
                // smallest terms first: (3,1) (2,2) (1,3) | (2,1) (1,2) | (1,1)
#define CLIFT_MM(A_, B_) c = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(B_, A_, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, c, 0, 0, 0)
                CLIFT_MM(al, bh); CLIFT_MM(am, bm); CLIFT_MM(ah, bl);
                CLIFT_MM(am, bh); CLIFT_MM(ah, bm);
                CLIFT_MM(ah, bh);
#undef CLIFT_MM
                acc[x][y] = c;
                // one accumulator chain at a time: hoisting the next tile's fragments above these MFMAs (to interleave two
                // chains) costs 12 more live VGPRs and spilled the prefetch registers -- the other waves of the SIMD fill the
                // dependent-issue gaps instead
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    };
    load_a(0, va0);
    load_b(0);
    for (int k0 = 0; k0 < g.K; k0 += BKS) step(k0, va0);
    gemm_epilogue<BM, BN, WM, WN, SWAP>(g, acc, m0, n0, wm, wn, li, lh, tid, false, 0.f);
}

template <int BM, int BN, int WM, int WN>
static int launch_split(const SplitP& sp, int a_trans, hipStream_t st) {
    const GemmP& p = sp.g;
    const dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, 1), block(WM * WN * 64);
    const bool vec = gemm_vector_epilogue_ok(p);
    (void)a_trans;      // the dispatcher only sends row-major A here (forward / dgrad)
    if (vec) k_gemm_split<BM, BN, WM, WN, true><<<grid, block, 0, st>>>(sp);
    else k_gemm_split<BM, BN, WM, WN, false><<<grid, block, 0, st>>>(sp);
    return clift_check_launch("clift_gemm(fp32x6)");
}

long clift_gemm_split_workspace_bytes(int N, int K) {
    const long Np = (long)cdiv(N, 256) * 256, Kp = (long)cdiv(K, BKS) * BKS;
    return 3 * Np * Kp * 2;
}

int clift_gemm_split_launch(const GemmP& p, int a_trans, int b_trans, void* workspace, hipStream_t st) {
    SplitP sp;
    sp.g = p;
    sp.Np = cdiv(p.N, 256) * 256;            // every block tile (BN <= 256) stays inside the padded planes
    sp.Kp = cdiv(p.K, BKS) * BKS;
    sp.planes = reinterpret_cast<const unsigned short*>(workspace);
    const long n = (long)sp.Np * sp.Kp;
    k_split_weights<<<cdiv(n, 256), 256, 0, st>>>(p.B, p.ldb, b_trans, p.N, p.K, sp.Np, sp.Kp, reinterpret_cast<unsigned short*>(workspace));
    if (int rc = clift_check_launch("clift_gemm(fp32x6 weight split)")) return rc;
    if (p.N > 128) return launch_split<128, 256, 2, 4>(sp, a_trans, st);
    if (p.N > 32) return launch_split<128, 128, 2, 2>(sp, a_trans, st);
    return launch_split<256, 32, 4, 1>(sp, a_trans, st);
}
