/* clift.h -- C ABI of libclift.so: the MI355X (gfx950) kernels of the Contrastive-Lift render hot path.
 *
 * The reference (yashbhalgat/Contrastive-Lift) is pure Python/PyTorch: it has no FFI layer.  The
 * boundary a maintainer would bind is therefore the set of ATen-op groups its renderer/field/loss
 * objects bottom out in.  Every entry point below cites the reference
 * file:line whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 / int32 data unless its name starts with h_ (host);
 *   - matrices are row-major with an explicit leading dimension where one is given;
 *   - VM tables are channels-last: plane i is [res[b_i]][res[a_i]][comps], line i is [res[v_i]][comps]
 *     with (a_i,b_i) = (0,1),(0,2),(1,2) and v_i = 2,1,0 (reference tensoRF.py:61-62,104-105; the PyTorch
 *     tensors keep the reference shape (1,C,H,W) and use channels_last strides);
 *   - the library never allocates or frees device memory, never synchronises the device and launches
 *     only on the caller's stream; the caller owns all buffers (torch tensors);
 *   - every function returns 0 on success, non-zero on error (message via clift_last_error());
 *     no C++ exception crosses this boundary;
 *   - one host thread per process drives the library (same threading model as the reference: one
 *     Python thread per DDP rank).
 */
#ifndef CLIFT_H
#define CLIFT_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* clift_stream_t; /* hipStream_t */

/* ABI version (bumped on any signature change) and last error string of the calling process. */
int clift_version(void);
/* Data-parallel runs (reference: DDP buckets overlapped with backward, trainer/__init__.py:93-108): leave k of the 256 CUs to
 * the collective's kernels while an asynchronous all-reduce is in flight -- the persistent launches (one block per CU, held for the
 * whole launch) then use 256 - k blocks.  k = 0 restores the default.  Host state; takes effect at the next launch. */
int clift_set_cu_reserve(int k);
const char* clift_last_error(void);

/* One VM-decomposed table set (3 planes + 3 lines, equal component count). */
typedef struct {
    const float* plane[3];
    const float* line[3];
    int res[3]; /* Rx, Ry, Rz */
    int comps;  /* components per plane (16 density / 48 appearance in the reference configs) */
} clift_vm_t;

/* Gradient accumulators for a table set (same layout; accumulated into with atomics).
 * xcd_stride == 0: the pointers are the final gradient tables, accumulated with device-scope atomics.
 * xcd_stride  > 0: the pointers address copy 0 of EIGHT zero-initialised accumulation copies, xcd_stride floats
 *   apart; every XCD of the MI355X accumulates into its own copy with atomics that execute in its private L2
 *   (selected by the hardware XCC id, placement-independent); the caller then folds the copies into the real
 *   gradient with clift_xcd_reduce (which also re-zeroes them). */
typedef struct {
    float* plane[3];
    float* line[3];
    long xcd_stride;
} clift_vm_grad_t;

/* dst[i] += sum_{x<8} work[x*xcd_stride + i] for i < n; work is re-zeroed. */
int clift_xcd_reduce(float* work, long xcd_stride, long n, float* dst, clift_stream_t s);

/* Renderer state: reference model/renderer/panopli_tensoRF_renderer.py:42-71 (buffers bbox_aabb,
 * inv_box_extent; python attrs step_size, n_samples, distance_scale, raymarch_weight_thres) and
 * tensoRF.py:35 (splus_density_shift). */
typedef struct {
    float lo[3];
    float hi[3];
    float inv_ext2[3]; /* 2 / (hi - lo) */
    float step_size;
    int n_samples;
    float distance_scale;
    float density_shift;
    float weight_thres;
} clift_march_t;

/* ---- a1-a3: util/ray.py:8-12,25-31,46-54,81-99.  rays (H*W, 8) = [o, d, near, far].
 * *bad_count is incremented for every ray whose sphere discriminant is negative (reference asserts). */
int clift_gen_rays(int H, int W, const float* h_K9, const float* h_c2w16, float near_plane, float* rays,
                   int* bad_count, clift_stream_t s);

/* ---- a4-a6: renderer.py:800-817 (sampling; jitter = perturb*rand per ray, NULL for none), :633-634
 * (normalise), tensoRF.py:108-125 (density VM lookup + shift + softplus).  sigma (N, S), 0 outside the box. */
int clift_density_fwd(const clift_march_t* h_m, const clift_vm_t* h_dens, const float* rays, const float* jitter,
                      int N, float* sigma, clift_stream_t s);

/* Point-wise field API (reference TensorVMSplit.compute_density / compute_density_without_activation,
 * tensoRF.py:114-125; used by the dense alpha grid of the bbox shrink, renderer.py:717-754).  xn (n, ldx) normalised
 * coordinates; out (n). */
int clift_density_points(const clift_vm_t* h_dens, const float* xn, int ldx, long n, float shift, int activation,
                         float* out, clift_stream_t s);
/* Alpha-mask bounding box of the epoch-boundary shrink (renderer.py:669-680,717-729,752-754): alpha = 1 - exp(-softplus(density +
 * shift) * step) on the (g0,g1,g2) lattice of the box [h_lo3, h_hi3] (lattice coordinate of axis a at index i =
 * lo_a (1 - s_a[i]) + hi_a s_a[i]; s0/s1/s2 are DEVICE arrays holding the caller's linspace(0,1,g_a)), clamp to [0,1], 3^3 max-pool
 * (stride 1, padding 1), >= threshold.  alpha_work: g0*g1*g2 floats of scratch (left holding the un-pooled alpha, x-major);
 * box7 (device, 7 ints) = [min index per axis (3), max index per axis (3), number of voxels above the threshold]; min = INT_MAX
 * and max = -1 when none is. */
int clift_alpha_bbox(const clift_vm_t* h_dens, const float* h_lo3, const float* h_hi3, const float* h_inv_ext2, const float* s0,
                     const float* s1, const float* s2, int g0, int g1, int g2, float shift, float step, float threshold,
                     float* alpha_work, int* box7, clift_stream_t s);
/* Plane x line products at arbitrary points, F (n, 3*comps) (compute_appearance_feature before the basis, :127-134). */
int clift_vm_products_points(const clift_vm_t* h_vm, const float* xn, int ldx, long n, float* F, clift_stream_t s);
/* Appearance-MLP input assembly from explicit view directions (render_appearance_mlp(viewdirs, features), :400-408). */
int clift_app_encode_points(const float* feat, int ldf, int nf, int pe_feat, int pe_view, const float* dirs, int ldd,
                            long n, float* X, int ldx, clift_stream_t s);

/* ---- a7-a8: renderer.py:83-84,100-103,137,173-174,626-631 + eff_distloss (renderer.py:101).
 * Per sample alpha, T (transmittance before the sample), w = alpha*T, all (N, S).
 * ray_out (N, 8) = [opacity, depth, bg, w_total, wm_total, dist_loss, t_min, 0]; n_active (N) = #(w > thres). */
int clift_march_fwd(const clift_march_t* h_m, const float* rays, const float* jitter, int N, const float* sigma,
                    float* alpha, float* T, float* w, float* ray_out, int* n_active, clift_stream_t s);

/* Backward of a7-a8: g_w (N,S) = dL/dw contributions from compositing (zero where inactive),
 * g_opacity (N) = dL/d(sum w), g_dist = device scalar dL/d(dist_reg) (nullable) -> dsigma (N,S) = dL/dsigma. */
int clift_march_bwd(const clift_march_t* h_m, const float* rays, const float* jitter, int N, const float* alpha,
                    const float* T, const float* w, const float* ray_out, const float* g_w,
                    const float* g_opacity, const float* g_dist, float* dsigma, clift_stream_t s);

/* Backward of a6: scatter dsigma through softplus and the VM products into the table gradients.  sigma (nullable) = the (N,S) output of
 * clift_density_fwd for the same rays: with it the softplus derivative is 1 - exp(-sigma) per sample; without it the sample's feature is
 * summed again over planes and channels (same value to fp32 round-off). */
int clift_density_bwd(const clift_march_t* h_m, const clift_vm_t* h_dens, const clift_vm_grad_t* h_grad,
                      const float* rays, const float* jitter, int N, const float* dsigma, const float* sigma, clift_stream_t s);

/* ---- compaction of active samples (replaces the boolean-mask gathers renderer.py:103-108):
 * ray_start (N+1) = exclusive scan of n_active; act_idx[ray_start[r] .. ray_start[r+1]) = r*S + k in
 * increasing k for every k with w > thres. */
int clift_scan_counts(const int* n_active, int N, int* ray_start, clift_stream_t s);
int clift_compact_fill(const float* w, const int* ray_start, int N, int S, float thres, int* act_idx,
                       clift_stream_t s);

/* ---- sync-free steps (no read-back of the active-sample count; the reference syncs at every boolean-mask gather,
 * renderer.py:97,105).  The caller sizes the compacted buffers by a capacity `cap` and binds ONE device int as the
 * library's dynamic row limit: every per-sample kernel then clamps its row count to min(its argument, *limit) and the
 * persistent kernels re-balance their row ranges over the true count.  The caller keeps INT_MAX in *limit outside a
 * pass; clift_scan_counts_capped writes min(total, cap) into it (limit_out = the bound address), clamps the offsets to
 * cap and records total in *overflow when total > cap (samples past the capacity are dropped -- memory-safe, and the
 * caller is expected to raise the capacity); clift_compact_fill_capped writes only rows < cap. */
int clift_bind_rows_limit(const int* dev_limit);

/* XCD-private gradient shards (ABI 12).  The kernels that end by adding a per-block partial of a small gradient tensor to memory (the weight
 * gradients of the MLP layers, the K = 3 / output layers' gradients) are slowed by contention: all eight XCDs read-modify-write the same few
 * cache lines at the same moment (8 - 20 us per launch).  With shards a block adds into the copy of the gradient range that belongs to its
 * XCD, and the eight copies are folded into the gradients once per pass.  Record (40 bytes of device memory, little endian):
 *   int64 lo      address of the first float of the sharded gradient range      int64 bytes   its length
 *   int64 shard0  address of shard 0 (shard x at shard0 + x * stride; 8 shards, zero-initialised by the caller ONCE)
 *   int64 stride  bytes between shards                                           int32 enabled, int32 pad
 * clift_bind_grad_shards(record) publishes the record's address once per process (NULL unbinds).  A pass is bracketed, stream-ordered, by
 * clift_grad_shards_begin(record, src, zero, zero_n): record := *src with enabled = 1, and zero_n floats at `zero` are cleared (the launch
 * takes the place of the caller's gradient fill), and clift_grad_shards_fold(record, first, n, disable): gradients[first .. first + n) +=
 * the eight shards, which are cleared again; disable != 0 ends the bracket.  Outside a bracket every kernel adds straight into the
 * gradients (the autograd path, direct C callers: nothing changes for them). */
int clift_bind_grad_shards(const void* dev_record);
int clift_grad_shards_begin(void* dev_record, const void* dev_src, float* zero, long zero_n, clift_stream_t s);
int clift_grad_shards_fold(void* dev_record, long first, long n, int disable, clift_stream_t s);   /* NULL unbinds; one-time, synchronous */
int clift_scan_counts_capped(const int* n_active, int N, int* ray_start, int cap, int* limit_out, int* overflow,
                             clift_stream_t s);
int clift_compact_fill_capped(const float* w, const int* ray_start, int N, int S, float thres, int* act_idx, int cap,
                              clift_stream_t s);

/* ---- a9: tensoRF.py:127-134 (plane x line products, plane-major concat) on the compacted samples.
 * F (M, 3*comps); xa (M, 4) = [xn.x, xn.y, xn.z, 0] (input of the xyz MLP heads, tensoRF.py:142-156). */
int clift_app_gather_fwd(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter,
                         const int* act_idx, int M, float* F, float* xa, clift_stream_t s);
/* xa only (instance / segment passes: renderer.py:204,285 evaluate the xyz heads without appearance). */
int clift_active_xyz(const clift_march_t* h_m, const float* rays, const float* jitter, const int* act_idx, int M,
                     float* xa, clift_stream_t s);
/* xa (nullable): the (M, 4) normalised positions clift_app_gather_fwd / clift_active_xyz produced for the same act_idx -- when
 * given (and comps <= 64), the backward runs in its wave-per-(segment, plane) form: the index work of a 32-sample segment is done once, in
 * parallel, and the serial walk keeps only the per-channel arithmetic (same per-sample terms, merged over 32 instead of 16 samples, i.e. a
 * different fp32 summation order). */
int clift_app_gather_bwd(const clift_march_t* h_m, const clift_vm_t* h_app, const clift_vm_grad_t* h_grad,
                         const float* rays, const float* jitter, const int* act_idx, int M, const float* dF,
                         const float* xa, clift_stream_t s);

/* ---- a10 input assembly: tensoRF.py:400-408,413-418.  X (M, ldx) = [feat(nf), dir(3), sin/cos PE(feat),
 * sin/cos PE(dir), zero pad]; ldx >= nf + 3 + 2*pe_feat*nf + 2*pe_view*3. */
int clift_app_encode_fwd(const float* feat, int ldf, int nf, int pe_feat, int pe_view, const float* rays,
                         const int* act_idx, int S, int M, float* X, int ldx, int x_bf16 /* X is bf16-stored */, clift_stream_t s);
int clift_app_encode_bwd(const float* feat, int ldf, int nf, int pe_feat, const float* dX, int ldx, int M,
                         float* dfeat, int lddf, clift_stream_t s);
/* The appearance head's front end in ONE launch (ABI 16): clift_app_gather_fwd, the basis Linear (tensoRF.py:65,127-134: feat = F Wb^T,
 * Wb (nf, 3*comps) with row pitch ldb, no bias) and clift_app_encode_fwd for tiles of 64 active samples, the products and the features
 * staying on the CU.  Writes xa (M, 4; nullable), feat (M, ldf; pad columns zero -- the encode backward reads it), X (M, ldx) fp32 and --
 * only when F is not NULL (a backward pass will want the products) -- F (M, 3*comps).  feat = F Wb^T runs on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32, two k halves added: fp32 round-off apart from the GEMM of the unfused path).  nf <= 28, nf <= ldf <= 28, ldx % 4 == 0, ldb % 4 == 0. */
int clift_app_front_fwd(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter, const int* act_idx,
                        int M, const float* Wb, int ldb, int nf, int pe_feat, int pe_view, float* xa, float* feat, int ldf, float* X,
                        int ldx, float* F, clift_stream_t s);
/* ... with the choice of X's storage (ABI 17): x_bf16 = 1 writes X as (M, ldx) bf16, ldx even (bf16 mode: the appearance MLP's input is bf16-stored
 * like every other streamed activation of that mode); x_bf16 = 0 is clift_app_front_fwd. */
int clift_app_front_fwd_x(const clift_march_t* h_m, const clift_vm_t* h_app, const float* rays, const float* jitter, const int* act_idx,
                          int M, const float* Wb, int ldb, int nf, int pe_feat, int pe_view, float* xa, float* feat, int ldf, void* X,
                          int ldx, float* F, int x_bf16, clift_stream_t s);

/* ---- nn.Linear building block on the matrix cores (tensoRF.py:65,393-397,475-491,576-582 and their
 * backward).  C[m][n] (+)= act( sum_k A(m,k) * B(n,k) + bias[n] ) * (mask[m][n] > 0)
 *   A(m,k) = a_trans ? A[k*lda+m] : A[m*lda+k];   B(n,k) = b_trans ? B[k*ldb+n] : B[n*ldb+k].
 * precision 0: fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32, exact fp32 products).
 * precision 1: "bf16 mode" -- A and B are rounded to bf16 (RNE) on their way into LDS, products on
 *   v_mfma_f32_32x32x16_bf16 with fp32 accumulation; all tensors stay fp32 in memory (BASELINE config 3).
 * precision 2: "fp32x6" -- fp32-faithful products on the bf16 matrix cores: every operand value is split exactly into three
 *   bf16 terms and the six leading cross products are accumulated in fp32 (error below fp32 product rounding).  Applies to
 *   the forward / dgrad forms (a_trans = 0, no accumulate); other forms run as precision 0.  Needs `workspace` -- except for the forms that
 *   have persistent split kernels, which need none: N = K = 256 forward / masked dgrad and the 256 x 256 weight gradient (csrc/layer_x6*.hip)
 *   and, ABI 18, the 128-wide appearance layers (csrc/layer_n6.hip: forward K in {128, 160} -> N = 128, masked dgrad 128 -> 128, unmasked dgrad
 *   128 -> 160, weight gradients 128 x {128, 160} with a_trans = b_trans = accumulate = 1).
 * K % 4 == 0, lda/ldb % 4 == 0, 16-byte aligned bases.
 * split_k > 1 partitions K over blockIdx.z and requires accumulate = 1 (atomic add into C).
 * Dispatch (no change of contract): the N = K = 256 hidden-layer forms with M >= 4096 -- forward (plain A, [n][k] weights, no
 *   mask) and masked dgrad (b_trans, fp32 mask, no bias/act) -- run as persistent kernels (csrc/layer_f32.hip: weights in registers,
 *   rows streamed by LDS-DMA); in precision 1 with bf16-stored A / C (/ mask) the same forms and the 256 x 256 wgrad run as
 *   streaming kernels (csrc/layer_bf16.hip).  Everything else takes the tiled kernels (csrc/gemm*.hip).  The environment variable
 *   CLIFT_NO_PERSISTENT=1 forces the tiled fp32 kernels (A/B comparisons in tests/test_gpu_parity.py). */
typedef struct {
    int M, N, K;
    const float* A; int lda; int a_trans;
    const float* B; int ldb; int b_trans;
    float* C; int ldc;
    const float* bias;              /* nullable */
    int act;                        /* 0 none, 1 relu */
    const float* mask; int ldmask;  /* nullable */
    int accumulate;
    int split_k;
    int c_trans;                    /* write C[n*ldc + m] (lets a narrow-M wgrad run as a narrow-N problem) */
    float* colsum;                  /* nullable; a_trans only: colsum[m] += sum_k A(m,k)  (bias gradient) */
    int precision;                  /* 0 fp32 operands, 1 bf16 operands (fp32 accumulate), 2 fp32x6 split (see above) */
    void* workspace;                /* precision 2 only: device scratch for the split weight planes, 16-byte aligned */
    long workspace_bytes;           /* >= clift_gemm_workspace_bytes(N, K) */
    int a_bf16, b_bf16, c_bf16, mask_bf16;  /* precision 1 only: the tensor is STORED as bf16 (2-byte elements, pitches in elements);
                                     * the pointers are passed through the float* fields */
    void* sign_bits;                /* nullable; precision 2, N = K = 256 persistent forms only (ABI 15): clift_sign_bits_bytes(M) bytes.  A forward
                                     * (act = 1) also WRITES the signs of its output there (one byte per lane of the kernel and 32-row tile: 32 B per
                                     * row); the masked dgrad of the NEXT layer (b_trans, mask = NULL) READS them instead of the 1 KB-per-row fp32 mask --
                                     * same results bit for bit.  The layout is private to csrc/layer_x6.hip: only pass what such a forward wrote for
                                     * the same M. */
} clift_gemm_t;
long clift_gemm_workspace_bytes(int N, int K);
long clift_sign_bits_bytes(int M);

/* Backward of a narrow output layer (no <= 32 outputs: 22 classes / 3 instance dims) over a 256-wide ReLU hidden layer in ONE pass over
 * the hidden activation H (tensoRF.py:480-481, 593-594 backward): dX = (H > 0) . (dOut W), gW += dOut^T H, gb += column sums of dOut.
 * dOut (M, ldd) fp32 with zero pad columns (no <= ldd <= 32, ldd % 4 == 0), W (no, 256) pitch ldw, H (M, 256) pitch ldh, dX (M, 256)
 * pitch ldx, gW (no, 256) pitch ldgw (accumulated), gb (no) nullable (accumulated); 16-byte aligned rows.  Replaces one clift_wgrad_narrow +
 * one masked clift_gemm(b_trans) call, which each stream H from memory. */
int clift_out_layer_bwd(const float* dOut, int ldd, int no, const float* W, int ldw, const float* H, int ldh, int M,
                        float* dX, int ldx, float* gW, int ldgw, float* gb, clift_stream_t s);
/* ... over a hidden layer of nh units, nh % 32 == 0, nh <= 256 (ABI 16: the 128-wide appearance head, tensoRF.py:393-397 backward; its
 * dOut is the gradient of the pre-sigmoid colours, (M, 4) with a zero pad column).  Pitches >= nh. */
int clift_out_layer_bwd_nh(const float* dOut, int ldd, int no, const float* W, int ldw, const float* H, int ldh, int nh, int M,
                           float* dX, int ldx, float* gW, int ldgw, float* gb,
                           int h_bf16 /* H and dX are bf16-stored (bf16 mode, nh = 256): the products stay fp32 */, clift_stream_t s);
/* Forward of a narrow output layer over a 256-wide hidden activation, with the row activation that follows it, in ONE pass over H (ABI 15;
 * tensoRF.py:591-594 with :37 for the semantic head):  out[m][0..no) = act( H[m][0..256) W^T + b ),  no <= 32, act 0 = none, 2 = softmax over
 * the row.  H (M, ldh) fp32, W (no, 256) pitch ldw, b (no), out (M, ldo) -- columns [no, ldo) untouched.  Replaces clift_gemm (which streams H at
 * less than half the rate for so narrow an output) + clift_rows_act_fwd; results differ from that pair by fp32 summation order only. */
int clift_out_layer_fwd(const float* H, int ldh, const float* W, int ldw, const float* b, int no, int M, float* out, int ldo, int act,
                        clift_stream_t s);
int clift_gemm(const clift_gemm_t* h_g, clift_stream_t s);

/* First layer of the xyz heads (in_features == 3): out (M, Nout) = act(x[:, :3] W^T + b); x is (M, 4), W (Nout, 3)
 * with row pitch ldw. */
int clift_linear_k3_fwd(const float* x4, const float* W, int ldw, const float* b, int M, int Nout, int relu,
                        float* out, int ldo, int out_bf16 /* out is bf16-stored (bf16 mode) */, clift_stream_t s);
/* First TWO layers of an xyz head in one launch (tensoRF.py:475-478, 576-579; 256-wide hidden layers, fp32):
 * h2 (M, ldh2) = relu(W1 relu(W0 x + b0) + b1).  The K = 3 layer is generated inside the second layer's persistent kernel
 * instead of being written (255 MB at the bench shape) and re-read.  h1 (nullable; (M, ldh1)) receives the first layer's
 * activation when a backward pass will need it.  Results are bit-identical to clift_linear_k3_fwd followed by clift_gemm. */
int clift_xyz_head_first2_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1,
                              const float* b1, int M, float* h1, int ldh1, float* h2, int ldh2, clift_stream_t s);
/* clift_xyz_head_first2_fwd in fp32x6 arithmetic (ABI 13): the 256 x 256 layer as six bf16 products per fp32 product on the bf16 matrix
 * cores (csrc/layer_x6.hip), the K = 3 layer exact; the first layer's activation is never written (the backward does not need it:
 * clift_xyz_head_first2_bwd / clift_xyz_head_first2_wgrad).  Results within 1e-6 (row-max relative) of clift_xyz_head_first2_fwd. */
int clift_xyz_head_first2_x6_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1,
                                 const float* b1, int M, float* h2, int ldh2, void* sign_bits /* nullable, see clift_gemm_t (ABI 15) */,
                                 clift_stream_t s);
/* Backward of the first TWO layers of an xyz head in bf16 mode, the part behind the second layer's weight gradient (ABI 16; the bf16 counterpart
 * of clift_xyz_head_first2_bwd, tensoRF.py:475-478, 576-579): dH2 (M, ldd) and the first activation h1 (M, ldm; the ReLU mask) bf16-STORED, W1
 * (256, 256) and the positions x4 (M, 4) fp32.  gW0 (256, ldg) += ((h1 > 0) . bf16(dH2 W1))^T x4[:, :3], gb0 += its column sums: the second
 * layer's input gradient is formed, rounded to bf16 and masked exactly as clift_gemm would store it, and consumed in the kernel -- never written. */
int clift_xyz_head_first2_bf16_bwd(const void* dH2, int ldd, const float* W1, int ldw1, const void* h1, int ldm, const float* x4, int M,
                                   float* gW0, int ldg, float* gb0, clift_stream_t s);
/* The fp32x6 counterparts of the three fused ends of an xyz head's backward / forward (ABI 14; csrc/layer_x6.hip, csrc/layer_x6w.hip): the
 * 256 x 256 products as six bf16 products per fp32 product of exactly three-way-split operands (fp32 accumulate; within 1e-6 row-max relative
 * of the exact-fp32 entry points they mirror), everything narrow -- the K = 3 layer, the ReLU masks, the E <= 4 output layer -- in exact fp32.
 *   clift_xyz_head_first2_x6_bwd    = clift_xyz_head_first2_bwd   (tensoRF.py:475-476, 576-577): dH1 never written, mask never read;
 *   clift_xyz_head_first2_x6_wgrad  = clift_xyz_head_first2_wgrad (tensoRF.py:477-478, 578-579): the first activation regenerated in the split;
 *   clift_xyz_head_last2_x6_fwd     = clift_xyz_head_last2_fwd    (tensoRF.py:478-481): the output layer applied to the tile in registers; the
 *     16 column-group partial sums of a row go through `workspace` (clift_xyz_head_last2_x6_workspace_bytes(M) bytes, 16-byte aligned,
 *     the caller's scratch until the call has run on the stream) and are added in a fixed order by a second small launch: deterministic,
 *     and a row's bits do not depend on how many rows share the launch. */
int clift_xyz_head_first2_x6_bwd(const float* dH2, int ldd, const float* W1, int ldw1, const float* W0, int ldw0, const float* b0,
                                 const float* x4, int M, float* gW0, int ldgw0, float* gb0, clift_stream_t s);
int clift_xyz_head_first2_x6_wgrad(const float* dH2, int ldd, const float* W0, int ldw0, const float* b0, const float* x4, int M,
                                   float* gW1, int ldgw1, float* gb1, clift_stream_t s);
long clift_xyz_head_last2_x6_workspace_bytes(int M);
int clift_xyz_head_last2_x6_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, void* workspace,
                                long workspace_bytes, clift_stream_t s);
/* Backward of the first TWO layers of an xyz head in one launch (tensoRF.py:475-478, 576-579; fp32, ABI 11).  dH2 (M, ldd) is the
 * gradient at the second layer's output, already masked by that layer's ReLU; W0 (256, 3) / b0 the first layer; x4 (M, 4) the sample
 * positions.  The second layer's input gradient dH1 = (W0 x + b0 > 0) . (dH2 W1) is formed tile by tile and consumed in place:
 *   gW0[n][0..2] += sum_m dH1[m][n] x4[m][0..2],   gb0[n] += sum_m dH1[m][n]
 * -- dH1 (1 KB per row) is never written, and the first layer's activation is not read: its sign is re-derived from the positions in the
 * forward's operation order (the same bits clift_linear_k3_fwd / clift_xyz_head_first2_fwd produced).  Replaces clift_gemm(b_trans, mask)
 * + clift_linear_k3_bwd; results differ from that pair by summation order only (both accumulate per-block partial sums with fp32
 * atomics).  The second layer's own weight gradient stays a clift_gemm call. */
int clift_xyz_head_first2_bwd(const float* dH2, int ldd, const float* W1, int ldw1, const float* W0, int ldw0, const float* b0,
                              const float* x4, int M, float* gW0, int ldgw0, float* gb0, clift_stream_t s);
/* Weight / bias gradient of the SECOND layer of an xyz head with its input -- the first layer's activation -- generated in-kernel from
 * the positions (ABI 11):  gW1[n][k] += sum_m dH2[m][n] relu(W0[k] . x4[m] + b0[k]),  gb1[n] += sum_m dH2[m][n]  (gb1 nullable).
 * With clift_xyz_head_first2_bwd this makes the first layer's activation (1 KB per sample) unnecessary for the backward: the forward
 * (clift_xyz_head_first2_fwd with h1 = NULL) never writes it.  Same bits as clift_gemm's weight gradient over the stored activation up
 * to the summation order of the per-block partial sums. */
int clift_xyz_head_first2_wgrad(const float* dH2, int ldd, const float* W0, int ldw0, const float* b0, const float* x4, int M,
                                float* gW1, int ldgw1, float* gb1, clift_stream_t s);
/* LAST hidden layer of an xyz head together with its narrow output layer (E <= 4 outputs: the instance heads,
 * tensoRF.py:478-481): h = relu(A W^T + b), out[:, 0:E] = h Wout^T + bout, in one launch -- the output layer is applied to the
 * tile while it is in registers instead of re-reading the 1 KB-per-row activation.  `hidden` (nullable, (M, ldh)) receives h when
 * a backward pass will need it.  The E sums are formed in a fixed order (deterministic); they differ from clift_gemm's by
 * summation order only. */
int clift_xyz_head_last2_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                             const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, clift_stream_t s);

/* Last hidden layer of the appearance MLP + its output layer + sigmoid in one launch (tensoRF.py:395-397,410; 128-wide, fp32):
 * h = relu(A W^T + b) (written to `hidden` if non-null), pre = h Wout^T + bout (written if `pre` non-null), out = sigmoid ?
 * 1/(1+exp(-pre)) : pre.  E <= 4.  Deterministic; differs from clift_gemm + clift_rows_act_fwd by summation order only. */
int clift_app_head_last2_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                             const float* bout, int E, int M, float* hidden, int ldh, float* pre, int ldp, float* out, int ldo,
                             int sigmoid, clift_stream_t s);
/* fp32x6 (ABI 18; csrc/layer_n6.hip): the same pair of layers with the 128 x 128 product as six bf16 products of exactly three-way-split
 * operands (fp32-faithful, see clift_gemm precision 2), the output layer in exact fp32 FMAs on the finished tile, its four column-group shares
 * summed in a fixed order; `hidden` (M, ldh) fp32 or NULL (not written).  E <= 4. */
int clift_app_head_last2_x6_fwd(const float* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                const float* bout, int E, int M, float* hidden, int ldh, float* out, int ldo, int sigmoid,
                                clift_stream_t s);
/* bf16 mode (ABI 17; csrc/layer_nb16.hip): the same pair of layers over a bf16-STORED input activation A (M, lda), W (128, 128) fp32 rounded to
 * bf16 in the kernel, fp32 accumulate; `hidden` (M, ldh) bf16-stored or NULL; the output layer takes the bf16-rounded hidden activation (what
 * the unfused pair would read back) against fp32 output weights, its four column-group shares summed in a fixed order.  E <= 4. */
int clift_app_head_last2_bf16_fwd(const void* A, int lda, const float* W, int ldw, const float* b, const float* Wout, int ldwo,
                                  const float* bout, int E, int M, void* hidden, int ldh, float* out, int ldo, int sigmoid,
                                  clift_stream_t s);
/* bf16 mode (ABI 17): backward of the no <= 4 wide output layer over the bf16-stored 128-wide hidden activation H (tensoRF.py:397 backward), one
 * pass over H:  dX (M, ldx) bf16-stored = (H > 0) . (dOut W),  gW (no, 128) += dOut^T H,  gb (no) += column sums of dOut (nullable).
 * dOut (M, ldd) fp32 with ldd >= 4 (pad columns ignored), products fp32. */
int clift_out_layer_bwd_n128_bf16(const float* dOut, int ldd, int no, const float* W, int ldw, const void* H, int ldh, int M,
                                  void* dX, int ldx, float* gW, int ldgw, float* gb, clift_stream_t s);
/* bf16 mode: the first three layers of an xyz head (K = 3 layer + two 256 x 256 hidden layers, tensoRF.py:475-479, 576-579) and,
 * for E in [1,4], its E-wide output layer, in ONE launch with the activations resident in LDS (bf16 operands, fp32 accumulate --
 * the arithmetic of clift_linear_k3_fwd + clift_gemm(precision 1) with bf16-stored activations).  h1 / h2 / h3: nullable bf16-stored
 * (M, 256) destinations of the hidden activations, passed only when a backward pass will need them; E = 0: no output layer, h3
 * (required) is the result; E > 0: out (M, ldo) fp32. */
int clift_xyz_head_bf16_fwd(const float* x4, const float* W0, int ldw0, const float* b0, const float* W1, int ldw1,
                            const float* b1, const float* W2, int ldw2, const float* b2, const float* Wout, int ldwo,
                            const float* bout, int E, int M, void* h1, void* h2, void* h3, float* out, int ldo,
                            clift_stream_t s);
/* dW (Nout,3; pitch ldw) += dH^T x ; db (Nout) += colsum(dH). */
int clift_linear_k3_bwd(const float* x4, const float* dH, int ldh, int M, int Nout, float* dW, int ldw, float* db,
                        int dh_bf16 /* dH is bf16-stored */, clift_stream_t s);
/* Narrow weight gradient (out_features no <= 32: the last layer of every head, tensoRF.py:395,480,581, and the basis
 * Linear :65 seen from its narrow side): gW (no, ni; pitch ldw) += dY^T X over M samples, gb (no; nullable) += colsum(dY).
 * A streaming VALU reduction at HBM rate instead of a mostly-padding matrix-core tile. */
int clift_wgrad_narrow(const float* dY, int ldd, int no, const float* X, int ldx, int ni, int M, float* gW, int ldw,
                       float* gb, int x_bf16 /* X is bf16-stored */, clift_stream_t s);
/* db (N) += colsum(dY (M,N)). */
int clift_colsum(const float* dY, int ld, int M, int N, float* db, clift_stream_t s);

/* Row activations of the head outputs: kind 1 = sigmoid (tensoRF.py:385,410), 2 = softmax (tensoRF.py:37,593);
 * the backward also accepts kind 0 = identity (re-pitches dout (M,C) into the 4-float-aligned dpre (M,ldd)). */
int clift_rows_act_fwd(const float* pre, int ldp, int M, int C, int kind, float* out, int ldo, clift_stream_t s);
int clift_rows_act_bwd(const float* out, int ldo, const float* dout, int lddo, int M, int C, int kind, float* dpre,
                       int ldd, clift_stream_t s);

/* ---- a13 compositing: renderer.py:137-167.  rgb_s (M,3) sem_s (M,C) inst_s (M,D) are per-active-sample
 * head outputs (NULL to skip a head).  Outputs rgb_raw (N,3) (pre-clamp), rgb_map (N,3), sem_raw (N,C)
 * (weighted sums), sem_map (N,C) (log-normalised when softmax_mode), inst_map (N,D). */
int clift_composite_fwd(const float* w, const int* ray_start, const int* act_idx, int N, int C, int D,
                        const float* rgb_s, const float* sem_s, const float* inst_s, const float* ray_out,
                        int softmax_mode, int white_bg, float* rgb_raw, float* rgb_map, float* sem_raw,
                        float* sem_map, float* inst_map, clift_stream_t s);
/* Backward.  stop_grad = renderer.stop_semantic_grad (w detached for semantics/instances, renderer.py:144-147).
 * Writes d_rgb_s/d_sem_s/d_inst_s (per active sample, M rows), g_w at the active positions of a ZEROED (N,S)
 * buffer, g_opacity (N).  Any of g_rgb/g_sem/g_inst (and the matching head buffers) may be NULL (treated as
 * zero / skipped).  ge_work: scratch of N*(3+C+D) floats. */
int clift_composite_bwd(const float* w, const int* ray_start, const int* act_idx, int N, int S, int M, int C, int D,
                        const float* rgb_s, const float* sem_s, const float* inst_s, const float* rgb_raw,
                        const float* sem_raw, int softmax_mode, int white_bg, int stop_grad, const float* g_rgb,
                        const float* g_sem, const float* g_inst, float* ge_work, float* d_rgb_s, float* d_sem_s,
                        float* d_inst_s, float* g_w, float* g_opacity, clift_stream_t s);
/* The same with the heads' output activations folded in (ABI 16): instead of d_rgb_s / d_sem_s / d_inst_s it writes the gradients w.r.t. the
 * PRE-activation outputs of the heads' last layers, in the zero-padded rows their backward kernels take -- dpre_rgb (M, ld_rgb): sigmoid backward
 * (tensoRF.py:398); dpre_sem (M, ld_sem): sem_kind 2 = softmax backward over the row (tensoRF.py:37,593), 0 = identity; dpre_i0 / dpre_i1
 * (M, ld_inst): instance columns [0, E) / [E, 2E) (fast / slow half, tensoRF.py:505-511; each nullable) -- so no clift_rows_act_bwd launch
 * follows.  The head buffers rgb_s / sem_s / inst_s must be given for the heads whose gradient is wanted. */
int clift_composite_bwd_act(const float* w, const int* act_idx, int N, int S, int M, int C, int D, const float* rgb_s, const float* sem_s,
                            const float* inst_s, const float* rgb_raw, const float* sem_raw, int softmax_mode, int white_bg, int stop_grad,
                            const float* g_rgb, const float* g_sem, const float* g_inst, float* ge_work, int sem_kind, float* dpre_rgb,
                            int ld_rgb, float* dpre_sem, int ld_sem, float* dpre_i0, float* dpre_i1, int ld_inst, int E, float* g_w,
                            float* g_opacity, clift_stream_t s);

/* ---- a18: model/loss/loss.py:14-22 on one channels-last plane (H,W,C); loss_accum[0] += weight*TV(x);
 * grad (nullable) += weight * dTV/dx. */
int clift_tv_fwd_bwd(const float* plane, int H, int W, int C, float weight, float* grad, float* loss_accum,
                     clift_stream_t s);
/* The same for up to CLIFT_TV_MAX planes in ONE launch (total_tv_loss, tensoRF.py:248-290: three density + three appearance
 * planes every step; each of those passes is a few MB, i.e. launch-latency-sized on this chip). */
#define CLIFT_TV_MAX 8
typedef struct {
    int n;
    const float* plane[CLIFT_TV_MAX];
    float* grad[CLIFT_TV_MAX];      /* nullable per plane */
    int H[CLIFT_TV_MAX], W[CLIFT_TV_MAX], C[CLIFT_TV_MAX];
    float weight[CLIFT_TV_MAX];
} clift_tv_set_t;
int clift_tv_fwd_bwd_multi(const clift_tv_set_t* set, float* loss_accum, clift_stream_t s);

/* ---- a19 pixel losses of training_step (trainer/train_panopli_tensorf.py:155-160,177-178):
 * out2[0] += mean((mask*(rgb-gt))^2); out2[1] += mean_i( mask_i conf_i * -sum_c cw_c p_ic log_softmax(sem_i)_c ).
 * g_rgb = w_rgb * d out2[0]/d rgb ; g_sem = w_sem * d out2[1]/d sem (either nullable); maskf (N) 0/1, nullable. */
int clift_pixel_losses(const float* rgb, const float* rgb_gt, const float* sem, const float* probs,
                       const float* conf, const float* class_w, const float* maskf, int N, int C, float w_rgb,
                       float w_sem, float* out2, float* g_rgb, float* g_sem, clift_stream_t s);

/* The same with SCELoss in place of the cross entropy (config use_symmetric_ce, trainer T:74-77; model/loss/loss.py:36-59):
 * per pixel alpha * CE + beta * RCE, RCE = -sum_c clamp(softmax(sem . cw), 1e-8, 1)_c log(clamp(p_c, 1e-8, 1)) cw_c. */
int clift_pixel_losses_sce(const float* rgb, const float* rgb_gt, const float* sem, const float* probs,
                           const float* conf, const float* class_w, const float* maskf, int N, int C, float w_rgb,
                           float w_sem, float alpha, float beta, float* out2, float* g_rgb, float* g_sem,
                           clift_stream_t s);
/* Per-pixel (reduction='none') form of the two semantic losses, as the reference's loss callables return them
 * (CrossEntropyLoss(weight, reduction='none') with soft targets: sce = 0; SCELoss: sce = 1): loss_rows (N),
 * grad_rows (N, C; nullable) = d loss_rows[i] / d pred[i, :]. */
int clift_semantic_loss_rows(const float* pred, const float* probs, const float* class_w, int N, int C, int sce,
                             float alpha, float beta, float* loss_rows, float* grad_rows, clift_stream_t s);

/* ---- segment-consistency term of training_step (trainer/train_panopli_tensorf.py:185-197): feats (B, ld) rendered semantic
 * features of the rays of G 2D segments, group (B) segment index of each ray.  target class of a segment = argmax of the mean
 * feature row (torch_scatter.scatter_mean); loss[0] += mean_i( class_w[t_i] conf_i CE(feats_i, t_i) ); grad (B, ldg), nullable,
 * = scale * d loss / d feats.  work >= G*C + G floats (zeroed by the call). */
int clift_segment_loss(const float* feats, int ld, const int* group, const float* conf, const float* class_w, int B,
                       int C, int G, float scale, float* work, float* loss, float* grad, int ldg, clift_stream_t s);

/* ---- a16: model/loss/loss.py:62-82.  loss[0] = value; g_feat (B,E) = d loss / d features (nullable). */
int clift_contrastive(const float* feat, const int* labels, int B, int E, float temperature, float* loss,
                      float* g_feat, float* work /* >= 4*B floats */, clift_stream_t s);

/* ---- a17: trainer/train_panopli_tensorf.py:261-309 (use_proj = False).  inst (B, 2E) = [fast | slow];
 * loss[0] = value; g_inst (B,2E) = gradient (zero for the slow half and the slow ray set).
 * work >= 8*B + 2*B*E floats. */
int clift_slow_fast(const float* inst, const int* labels, const float* conf, int B, int E, float* loss,
                    float* g_inst, float* work, clift_stream_t s);

/* ---- inference post-processing: inference/render_panopli.py:371-419 (assign_clusters: per-class cdist + argmin against
 * cached centroids).  labels[i] = argmin_c |feat_i - centroid_c| where valid[i] != 0 (NULL = all), else -1. */
int clift_nearest_centroid(const float* feat, int ldf, int E, const float* centroids, int K, const unsigned char* valid,
                           long n, int* labels, clift_stream_t s);

/* ---- optimiser plumbing on flat fp32 ranges: torch.optim.Adam semantics (L2 weight decay folded into the
 * gradient; bias correction with step >= 1) and the slow-net EMA (trainer T:325-329). */
int clift_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, clift_stream_t s);
int clift_ema(float* slow, const float* fast, long n, float momentum, clift_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* CLIFT_H */
