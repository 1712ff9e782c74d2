// pmc_harness.cpp -- a torch-free launcher for counter passes (rocprofv3 --pmc) over single libclift.so kernels.
// A counter pass serialises every dispatch and re-runs nothing else, so the process under the profiler should do as little as
// possible: this program allocates the operands of ONE 256x256 hidden-layer launch shape with hipMalloc, fills them with a
// pseudo-random pattern, and issues `reps` launches of the chosen form through the C ABI.
//   pmc_harness fwd|dgrad|wgrad|gen|outv M reps [precision]     (gen: clift_xyz_head_first2_fwd, activation kept; outv: clift_xyz_head_last2_fwd, E = 3, hidden dropped)
// Build: hipcc -O2 --offload-arch=gfx950 tools/pmc_harness.cpp -Iinclude -Lcontrastive_lift_amd -lclift -Wl,-rpath,'$ORIGIN/../contrastive_lift_amd' -o tools/pmc_harness.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "clift.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static float* dev_random(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * scale; }
    float* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s fwd|dgrad|wgrad M reps\n", argv[0]); return 1; }
    const char* mode = argv[1];
    const int M = atoi(argv[2]), reps = atoi(argv[3]);
    float* A = dev_random((size_t)M * 256, 1, 1.0f);
    float* X = dev_random((size_t)M * 256, 2, 1.0f);
    float* W = dev_random(256 * 256, 3, 0.06f);
    float* b = dev_random(256, 4, 0.1f);
    float* Cc = nullptr; CK(hipMalloc(&Cc, (size_t)M * 256 * sizeof(float)));
    float* gW = nullptr; CK(hipMalloc(&gW, 256 * 256 * sizeof(float))); CK(hipMemset(gW, 0, 256 * 256 * sizeof(float)));
    float* gb = nullptr; CK(hipMalloc(&gb, 256 * sizeof(float))); CK(hipMemset(gb, 0, 256 * sizeof(float)));
    if (!A || !X || !W || !b) { fprintf(stderr, "alloc failed\n"); return 2; }
    const bool gen = !strcmp(mode, "gen"), outv = !strcmp(mode, "outv");
    clift_gemm_t g; memset(&g, 0, sizeof g);
    const int precision = argc > 4 ? atoi(argv[4]) : 0;          // 2 = fp32x6 (persistent split kernels for the 256 x 256 forward / dgrad)
    if (!strcmp(mode, "fwd")) {
        g.M = M; g.N = 256; g.K = 256; g.A = A; g.lda = 256; g.B = W; g.ldb = 256; g.C = Cc; g.ldc = 256; g.bias = b; g.act = 1; g.split_k = 1;
    } else if (!strcmp(mode, "dgrad")) {
        g.M = M; g.N = 256; g.K = 256; g.A = A; g.lda = 256; g.B = W; g.ldb = 256; g.b_trans = 1; g.C = Cc; g.ldc = 256; g.mask = X; g.ldmask = 256; g.split_k = 1;
    } else if (gen || outv) {
    } else {        // wgrad: gW (256,256) += dY^T X over M rows
        g.M = 256; g.N = 256; g.K = M; g.A = A; g.lda = 256; g.a_trans = 1; g.B = X; g.ldb = 256; g.b_trans = 1; g.C = gW; g.ldc = 256; g.accumulate = 1;
        g.colsum = gb;
        int tiles = 2 * 1;
        int sp = (512 + tiles - 1) / tiles; if (sp > (M + 255) / 256) sp = (M + 255) / 256; if (sp < 1) sp = 1;
        g.split_k = sp;
    }
    g.precision = precision;
    float* x4 = dev_random((size_t)M * 4, 5, 1.0f);
    float* W0 = dev_random(256 * 4, 6, 0.7f);
    float* Wo = dev_random(4 * 256, 7, 0.1f);
    float* out3 = nullptr; CK(hipMalloc(&out3, (size_t)M * 4 * sizeof(float)));
    auto launch = [&]() -> int {
        if (gen) return clift_xyz_head_first2_fwd(x4, W0, 4, b, W, 256, b, M, X, 256, Cc, 256, nullptr);
        if (outv) return clift_xyz_head_last2_fwd(A, 256, W, 256, b, Wo, 256, b, 3, M, nullptr, 256, out3, 4, nullptr);
        return clift_gemm(&g, nullptr);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (launch()) { fprintf(stderr, "launch: %s\n", clift_last_error()); return 3; }     // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r)
        if (launch()) { fprintf(stderr, "launch: %s\n", clift_last_error()); return 3; }
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, fl = 2.0 * M * 65536.0;
    printf("%s M=%d reps=%d: %.1f us/launch, %.1f TFLOP/s; algorithmic bytes/launch = %.0f\n", mode, M, reps, us, fl / us / 1e6,
           !strcmp(mode, "fwd") ? 8.0 * M * 256 + 262144.0 : !strcmp(mode, "dgrad") ? 12.0 * M * 256 + 262144.0 :
           gen ? 8.0 * M * 256 + 16.0 * M + 262144.0 : outv ? 4.0 * M * 256 + 12.0 * M + 262144.0 : 8.0 * M * 256 + 262144.0);
    return 0;
}
