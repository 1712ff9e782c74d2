"""Kernel sequencing for the render hot path: explicit forward / backward over libclift.so entry points.

No torch autograd inside: ``render_forward`` returns the outputs plus a context holding the saved
activations, ``render_backward`` turns output gradients into parameter gradients written straight into the
caller's gradient views (the field's gradient arena in training).  ``renderer.py`` wraps this pair in a
``torch.autograd.Function`` for drop-in use.

Per chunk of N rays (reference renderer.py:80-176):
  density_fwd -> march_fwd -> scan/compact (one host read of the active count) ->
  appearance gather -> basis GEMM -> encode -> 2 GEMM+ReLU -> GEMM -> sigmoid
  xyz heads: K=3 first layer -> GEMM+ReLU chain -> narrow GEMM (-> softmax)
  composite.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import Gemm, March, VM, VMGrad, call, ptr, stream


# ----------------------------------------------------------------------------- struct builders
def march_struct(renderer, model):
    m = March()
    lo, hi = renderer.bbox_aabb_host
    inv = renderer.inv_box_extent_host
    for i in range(3):
        m.lo[i], m.hi[i], m.inv_ext2[i] = lo[i], hi[i], inv[i]
    m.step_size = renderer.step_size_host
    m.n_samples = int(renderer.n_samples)
    m.distance_scale = float(renderer.distance_scale)
    m.density_shift = float(model.splus_density_shift)
    m.weight_thres = float(renderer.raymarch_weight_thres)
    return m


def vm_struct(views, prefix, res):
    v = VM()
    for i in range(3):
        v.plane[i] = views[f"{prefix}_plane.{i}"].data_ptr()
        v.line[i] = views[f"{prefix}_line.{i}"].data_ptr()
        v.res[i] = int(res[i])
    v.comps = int(views[f"{prefix}_plane.0"].shape[1])
    return v


_GROUP_OF = {"density": "grid_density", "appearance": "grid_app", "semantic": "grid_sem", "instance": "grid_inst"}


def vm_grad_struct(model, gviews, prefix):
    """Gradient target of a scatter kernel: eight per-XCD accumulation copies (model-owned, persistent, zero between uses: XCD-local L2
    atomics) that ``vm_grad_finish`` folds into ``gviews``.  (xcd_stride = 0 in the C struct means "the final tables directly".)"""
    g = VMGrad()
    a, b = model.arena.range_of(_GROUP_OF[prefix])
    n = b - a
    work = model.xcd_workspace(prefix, 8 * n)
    base = work.data_ptr()
    for i in range(3):
        g.plane[i] = base + 4 * (model.arena.by_name[f"{prefix}_plane.{i}"].offset - a)
        g.line[i] = base + 4 * (model.arena.by_name[f"{prefix}_line.{i}"].offset - a)
    g.xcd_stride = n
    return g


def vm_grad_finish(model, gviews, prefix, g):
    if g.xcd_stride > 0:
        a, b = model.arena.range_of(_GROUP_OF[prefix])
        work = model.xcd_workspace(prefix, 8 * (b - a))
        # the tables of one group are contiguous in every arena-layout buffer, starting at plane 0
        dst = gviews[f"{prefix}_plane.0"]
        call("clift_xcd_reduce", ptr(work), g.xcd_stride, b - a, C.c_void_p(dst.data_ptr()), stream())


def grid_res(views):
    """(Rx,Ry,Rz) from the density table shapes: plane0 is (1,C,Ry,Rx), line0 is (1,C,Rz,1)."""
    p0, l0 = views["density_plane.0"], views["density_line.0"]
    return (p0.shape[3], p0.shape[2], l0.shape[2])


def _pitch(t):
    return t.stride(0) if t.dim() == 2 else 1


# MLP arithmetic of the matrix-core launches: 0 = "fp32" (exact fp32 products on the fp32 MFMA), 1 = "bf16" (operands rounded to bf16 in the
# kernel, fp32 accumulate, hidden activations bf16-stored -- BASELINE configs[2] / SURVEY 7.9), 2 = "fp32x6": every fp32 operand split
# EXACTLY into three bf16 terms (8 + 8 + 8 significant bits), the six leading cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate --
# fp32-faithful (the dropped terms are below one fp32 rounding of the product) for the 256-wide layers; narrow / 128-wide layers stay exact.
_PRECISIONS = {"fp32": 0, "f32": 0, "bf16": 1, "fp32x6": 2}
# The default since round 4 is fp32x6: fp32 results to fp32 round-off (row-max relative error against float64 2e-7 .. 8e-7, the exact kernels'
# own; whole-step gradients: no more entries outside the fp64 band than the exact path or the fp32 CPU oracle, profiles/r04_fp64_outliers.txt),
# 1.3 x the step rate of the exact-fp32 MFMA kernels.  "fp32" selects those (v_mfma_f32_32x32x2_f32: bit-identical to an fmaf chain).
DEFAULT_MLP_DTYPE = "fp32x6"
MLP_PRECISION = _PRECISIONS[os.environ.get("CLIFT_MLP_DTYPE", DEFAULT_MLP_DTYPE).lower()]


class _Precision:
    """Temporarily run the matrix-core launches at another operand precision (wave-uniform host state, restored on exit)."""

    def __init__(self, p):
        self.p = p

    def __enter__(self):
        global MLP_PRECISION
        self.prev, MLP_PRECISION = MLP_PRECISION, self.p

    def __exit__(self, *a):
        global MLP_PRECISION
        MLP_PRECISION = self.prev


def exact_fp32():
    """Context: matrix-core launches on the exact-fp32 kernels whatever mode is set or forced (tests of the exact kernels' fusions)."""
    return _Precision(0)


# bf16 mode and the 128-wide appearance MLP.  Rounds 2 - 4 kept it on the exact-fp32 persistent kernels (layer_n128.hip): faster than the TILED bf16
# kernels on these short-K layers, but bound by the fp32 matrix pipe (6 launches, 635 us of a 3.4 ms step).  Round 5: streaming bf16 kernels with
# bf16-STORED input / hidden activations / hidden gradients (csrc/layer_nb16.hip), like the xyz heads of that mode.  Tests clear APP_BF16 to get the
# old arrangement back.
APP_BF16 = True
# fp32x6 mode and the same layers.  Rounds 3 - 4 kept them exact as well (the only split kernel for their shapes was the tiled one: slower than the
# exact persistent kernels).  Round 5: persistent split kernels for the 128-wide forms (csrc/layer_n6.hip) -- 6 / 16 of the matrix time, the same
# fp32 streams.  Tests clear APP_X6 to get the exact arrangement back.
APP_X6 = True


def _app_precision():
    # fp32x6 (2): only the 256 x 256 layers have persistent split kernels; the 128-wide appearance layers would fall to the TILED split kernel
    # (gemm_split.hip), which is slower than the exact persistent kernels and -- seen with two processes sharing the GPU -- the one kernel of that
    # mode whose results were disturbed by the other process's fp32x6 launches (profiles/r03_x6_notes.txt).  Exact fp32 there.
    if MLP_PRECISION == 1 and APP_BF16 and os.environ.get("CLIFT_NO_PERSISTENT") is None:
        return _Precision(1)
    if MLP_PRECISION == 2 and APP_X6 and os.environ.get("CLIFT_NO_PERSISTENT") is None and os.environ.get("CLIFT_X6_TILED") is None:
        return _Precision(2)
    if MLP_PRECISION in (1, 2):
        return _Precision(0)
    return _Precision(MLP_PRECISION)


def act_dtype():
    """Storage type of the hidden activations / hidden gradients of the xyz-MLP heads: bf16 in bf16 mode (they are only ever
    consumed as bf16 matrix-core operands or as ReLU masks there), fp32 otherwise."""
    return torch.bfloat16 if MLP_PRECISION == 1 else torch.float32


def set_mlp_precision(name):
    """'fp32', 'bf16' or 'fp32x6'; returns the previous setting's name."""
    global MLP_PRECISION
    prev = {0: "fp32", 1: "bf16", 2: "fp32x6"}[MLP_PRECISION]
    MLP_PRECISION = _PRECISIONS[str(name).lower()]
    return prev


def sign_bits_for(M, dev):
    """Scratch for the sign bytes of an (M, 256) activation written by a persistent fp32x6 forward (clift_gemm_t.sign_bits): 32 B per row."""
    return torch.empty(((M + 31) // 32 * 1024,), dtype=torch.uint8, device=dev)


def gemm(M, N, K, A, lda, B, ldb, Cm, ldc, a_trans=0, b_trans=0, bias=None, act=0, mask=None, ldmask=0,
         accumulate=0, split_k=1, a_off=0, c_off=0, c_trans=0, colsum=None, sign_bits=None):
    """One clift_gemm launch.  Pitches and offsets are in ELEMENTS of the respective tensor; a tensor of dtype bfloat16 is
    passed as bf16-stored (bf16 mode only: hidden activations and their gradients)."""
    g = Gemm()
    g.M, g.N, g.K = int(M), int(N), int(K)
    g.A, g.lda, g.a_trans = A.data_ptr() + A.element_size() * a_off, int(lda), int(a_trans)
    g.B, g.ldb, g.b_trans = B.data_ptr(), int(ldb), int(b_trans)
    g.C, g.ldc = Cm.data_ptr() + Cm.element_size() * c_off, int(ldc)
    g.bias = bias.data_ptr() if bias is not None else None
    g.act = int(act)
    g.mask = mask.data_ptr() if mask is not None else None
    g.ldmask = int(ldmask)
    g.accumulate, g.split_k = int(accumulate), int(split_k)
    g.c_trans = int(c_trans)
    g.colsum = colsum.data_ptr() if colsum is not None else None
    g.sign_bits = sign_bits.data_ptr() if sign_bits is not None else None
    # fp32x6 mode: the 256 x 256 hidden layers (forward / dgrad) run as persistent split kernels (csrc/layer_x6.hip); every other
    # launch -- narrow layers (HBM streams), the 128-wide appearance MLP, the other weight gradients -- stays on its exact-fp32 persistent kernel,
    # which is faster than the tiled split kernel the library would pick for it
    x6 = N == 256 and K == 256 and not a_trans and not accumulate and not c_trans
    # ... and their weight gradients (csrc/layer_x6w.hip)
    x6w = (a_trans and b_trans and M == 256 and N == 256 and K >= 4096 and accumulate and not c_trans and bias is None and mask is None
           and not act and int(lda) % 4 == 0 and int(ldb) % 4 == 0 and A.dtype == torch.float32 and B.dtype == torch.float32)
    # ... and the 128-wide layers of the appearance MLP (csrc/layer_n6.hip): forward K in {128, 160} -> 128, masked dgrad 128 -> 128, unmasked dgrad
    # 128 -> 160, weight gradients 128 x {128, 160}
    f32s = A.dtype == torch.float32 and B.dtype == torch.float32 and Cm.dtype == torch.float32
    n6 = (f32s and not a_trans and not accumulate and not c_trans and int(lda) % 4 == 0 and int(lda) >= K and int(ldc) % 4 == 0 and g.C % 16 == 0 and
          ((not b_trans and N == 128 and K in (128, 160) and mask is None and int(ldb) >= K and act in (0, 1)) or
           (b_trans and K == 128 and bias is None and not act and
            ((N == 128 and mask is not None and int(ldmask) % 4 == 0 and int(ldmask) >= 128 and g.mask % 16 == 0) or (N == 160 and mask is None and int(ldb) >= 160)))))
    n6w = (f32s and a_trans and b_trans and M == 128 and N in (128, 160) and accumulate and not c_trans and bias is None and mask is None and not act
           and int(lda) % 4 == 0 and int(lda) >= 128 and int(ldb) % 4 == 0 and int(ldb) >= N and int(ldc) >= N)
    if os.environ.get("CLIFT_NO_PERSISTENT") is not None or os.environ.get("CLIFT_X6_TILED") is not None:
        n6 = n6w = False
    g.precision = MLP_PRECISION if (MLP_PRECISION != 2 or x6 or x6w or n6 or n6w) else 0
    g.a_bf16, g.b_bf16 = int(A.dtype == torch.bfloat16), int(B.dtype == torch.bfloat16)
    g.c_bf16, g.mask_bf16 = int(Cm.dtype == torch.bfloat16), int(mask is not None and mask.dtype == torch.bfloat16)
    if g.precision == 2 and not (x6w or n6 or n6w):
        # the persistent split kernel takes 16-byte aligned rows; anything else (odd output pitch ...) goes to the library's tiled split
        # kernel, which needs a workspace for the split weight planes
        persistent = (int(lda) % 4 == 0 and int(ldc) % 4 == 0 and g.C % 16 == 0 and (mask is None or (int(ldmask) % 4 == 0 and g.mask % 16 == 0))
                      and ((not b_trans and mask is None) or (b_trans and (mask is not None or sign_bits is not None) and bias is None and not act))
                      and not os.environ.get("CLIFT_X6_TILED"))
        if not persistent:
            nbytes = int(_lib.load().clift_gemm_workspace_bytes(int(N), int(K)))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)      # split weight planes (stream-ordered scratch)
            if _lib._launch_stream is not None:   # launched on a side stream (Branches): the allocator must not recycle it earlier
                ws.record_stream(_lib._launch_stream)
            g.workspace, g.workspace_bytes = ws.data_ptr(), nbytes
    call("clift_gemm", C.byref(g), stream())


def wgrad(no, ni, M, dY, ldd, X, ldx, gW, gb):
    """gW (no, ni) += dY^T X over the M samples; gb (no) += column sums of dY.  Wide layers: matrix cores, the bias sum
    fused into the GEMM (read from the A tile already staged in LDS).  Narrow layers (no <= 32): a streaming VALU
    reduction at HBM rate (clift_wgrad_narrow)."""
    if no > 32:
        gemm(no, ni, M, dY, ldd, X, ldx, gW, _pitch(gW), a_trans=1, b_trans=1, accumulate=1, split_k=_splits(no, ni, M), colsum=gb)
    else:
        if dY.dtype != torch.float32:
            raise _lib.CliftError("narrow wgrad: the output-side gradient must be fp32")
        call("clift_wgrad_narrow", ptr(dY), ldd, no, ptr(X), ldx, ni, M, ptr(gW), _pitch(gW), ptr(gb), int(X.dtype == torch.bfloat16), stream())


def _splits(out_rows, out_cols, K):
    """Split the (huge) sample-reduction of a wgrad over blockIdx.z so that the launch is ONE resident wave of workgroups
    (256 CUs x 2 blocks = 512): measured at 265 k samples, 256x256 / 128x128 / 128x152 outputs: 512 blocks 332 / 125 / 208 us,
    1024 blocks 378 / 165 / 261 us, 384 or 640 blocks worse than either (partial second wave, twice the atomics)."""
    tiles = ((out_rows + 127) // 128) * ((out_cols + 255) // 256 if out_cols > 128 else (out_cols + 127) // 128 if out_cols > 32 else 1)
    return max(1, min((K + 255) // 256, (512 + tiles - 1) // tiles))


def _lin_params(seq, views_prefix, views):
    """[(W, b)] of an nn.Sequential MLP, fetched from a name->tensor dict (parameter or gradient views)."""
    out = []
    j = 0
    while f"{views_prefix}.{j}.weight" in views:
        out.append((views[f"{views_prefix}.{j}.weight"], views[f"{views_prefix}.{j}.bias"]))
        j += 2
    return out


# ----------------------------------------------------------------------------- concurrent head chains
MULTI_STREAM = os.environ.get("CLIFT_STREAMS", "0") == "1"   # opt-in; +4 % with the tiled kernels, -3 % with the persistent ones
_side_streams = {}


class Branches:
    """Fork/join of independent kernel chains over HIP streams.  The appearance, semantic, fast-instance and slow-instance
    heads are independent between the compaction and the compositing, and every layer-wide GEMM has a start-up phase and
    an un-overlapped output burst (see gemm.hip) that another chain's MFMA work can fill.  Launches of a branch go to a
    side stream (via _lib.set_launch_stream); ALL allocations stay on torch's current stream, and every temporary of a
    branch is kept alive in ``self.keep`` until ``join`` so that no block is recycled into another in-flight branch."""

    def __init__(self, enabled=True, model=None):
        # a head on its own VM grid (engine.grid_head_fwd / _bwd) issues torch ops -- zero fills, add_ -- which live on torch's CURRENT stream, not
        # on the branch's side stream: such a field keeps every branch on the main stream
        if model is not None and (getattr(model, "semantic_plane", None) is not None or getattr(model, "instance_plane", None) is not None):
            enabled = False
        self.enabled = enabled and MULTI_STREAM
        self.keep = []
        self.used = []
        if self.enabled:
            self.main = torch.cuda.current_stream()
            self.fork = torch.cuda.Event()
            self.fork.record(self.main)

    def run(self, idx, fn):
        if not self.enabled:
            fn(self.keep)
            return
        dev = torch.cuda.current_device()
        st = _side_streams.get((dev, idx))
        if st is None:
            st = _side_streams[(dev, idx)] = torch.cuda.Stream(device=dev)
        st.wait_event(self.fork)
        _lib.set_launch_stream(st)
        try:
            fn(self.keep)
        finally:
            _lib.set_launch_stream(None)
        self.used.append(st)

    def join(self):
        if self.enabled:
            for st in self.used:
                ev = torch.cuda.Event()
                ev.record(st)
                self.main.wait_event(ev)
        self.keep = []
        self.used = []


# ----------------------------------------------------------------------------- xyz MLP heads (semantic / instance)
def last2(M, h, W, b, Wo, bo, hidden, out, ldo, col_off):
    """One clift_xyz_head_last2_fwd launch: hidden = relu(h W^T + b) (written if not None), out[:, col_off:col_off+E] = hidden Wo^T + bo."""
    call("clift_xyz_head_last2_fwd", ptr(h), h.shape[1], ptr(W), _pitch(W), ptr(b), ptr(Wo), _pitch(Wo), ptr(bo), Wo.shape[0], M, ptr(hidden), 256,
         C.c_void_p(out.data_ptr() + 4 * col_off), ldo, stream())


def app_last2(M, H1, W2, b2, W3, b3, H2, rgb):
    """One clift_app_head_last2_fwd launch: H2 = relu(H1 W2^T + b2) (written if not None), rgb = sigmoid(H2 W3^T + b3)."""
    call("clift_app_head_last2_fwd", ptr(H1), H1.shape[1], ptr(W2), _pitch(W2), ptr(b2), ptr(W3), _pitch(W3), ptr(b3), W3.shape[0], M,
         ptr(H2), 128, None, 0, ptr(rgb), rgb.shape[1], 1, stream())




def head_bf16(M, xa, l0, l1, l2, lout, h1, h2, h3, out, ldo, col_off):
    """One clift_xyz_head_bf16_fwd launch (csrc/head_bf16.hip)."""
    (W0, b0), (W1, b1), (W2, b2) = l0, l1, l2
    Wo, bo = lout if lout is not None else (None, None)
    call("clift_xyz_head_bf16_fwd", ptr(xa), ptr(W0), _pitch(W0), ptr(b0), ptr(W1), _pitch(W1), ptr(b1), ptr(W2), _pitch(W2), ptr(b2),
         ptr(Wo), _pitch(Wo) if Wo is not None else 0, ptr(bo), Wo.shape[0] if Wo is not None else 0, M, ptr(h1), ptr(h2), ptr(h3),
         C.c_void_p(out.data_ptr() + 4 * col_off) if lout is not None else None, ldo, stream())


def first2(M, xa, W0, b0, W1, b1, h1, h2):
    """One clift_xyz_head_first2_fwd launch: h2 = relu(W1 relu(W0 x + b0) + b1); h1 (or None) receives the first layer's activation.
    (A module-level function so that bench.py can bracket these launches with events like it does engine.gemm.)"""
    call("clift_xyz_head_first2_fwd", ptr(xa), ptr(W0), _pitch(W0), ptr(b0), ptr(W1), _pitch(W1), ptr(b1), M, ptr(h1), 256, ptr(h2), 256, stream())


# The fusions of the xyz heads are not switchable (their A/Bs are settled: profiles/r02_*, r03_ab_first2_bwd.txt, r04_ab_x6_fused_ends.txt): the K = 3
# layer is generated inside the second layer's kernel; the first activation is never written -- the backward consumes the second layer's input
# gradient inside its kernel (first2_bwd) and regenerates the activation for the weight gradient (first2_wgrad); an E <= 4 output layer is
# applied inside the last hidden layer's kernel; the output layer's weight and input gradient are one launch.  CLIFT_NO_PERSISTENT=1 selects
# the independent tiled kernels for every layer instead (the cross-check the tests use).
APP_SCATTER_XA = True       # tests clear it: clift_app_gather_bwd without the forward's positions (xa = NULL: the lane-per-(plane, channel) walk)
DENS_BWD_SIGMA = True       # tests clear it: clift_density_bwd without the forward's sigma (sigma = NULL: the softplus derivative re-summed)
FIRST2_BF16_BWD_FUSED = True  # tests clear it: bf16 mode, the second layer's input gradient written and the K = 3 layer's weight gradient as its own launch
OUT_BWD_BF16_FUSED = True   # tests clear it: bf16 mode, output layer's weight gradient and masked input gradient as two passes over the hidden activation
COMPOSITE_ACT_FUSED = True  # tests clear it: the heads' output activations taken back by clift_rows_act_bwd launches after the compositing backward
APP_OUT_BWD_FUSED = True    # tests clear it: the appearance output layer's weight gradient and masked input gradient as two passes over the hidden activation
APP_FRONT_FUSED = True      # tests clear it: appearance gather, basis GEMM and input encoding as three launches (the form the fused front end replaced)
KEEP_FIRST_ACT = False      # tests set this to compare against the stored-activation backward (masked dgrad + K = 3 weight gradient)
SOAK_KEEP = False           # tools/determinism_soak.py sets it: a forward that retains nothing still leaves REFERENCES to the appearance chain's intermediates in ctx.soak
FUSE_HEAD_BF16 = True       # bf16 mode: first three layers (+ E <= 4 output layer) of an xyz head in one launch; tests clear it to compare with the per-layer launches


def first2_bwd(M, d, W1, W0, b0, xa, gW0, gb0):
    """One clift_xyz_head_first2_bwd launch: gW0 += ((W0 x + b0 > 0) . (d W1))^T xa[:, :3], gb0 += column sums of the same masked gradient.
    (Module-level for the same reason as first2.)"""
    call("clift_xyz_head_first2_bwd", ptr(d), d.shape[1], ptr(W1), _pitch(W1), ptr(W0), _pitch(W0), ptr(b0), ptr(xa), M, ptr(gW0), _pitch(gW0), ptr(gb0),
         stream())


def first2_x6(M, xa, W0, b0, W1, b1, h2, sign_bits=None):
    """One clift_xyz_head_first2_x6_fwd launch (fp32x6 mode): h2 = relu(W1 relu(W0 x + b0) + b1), the first activation not written;
    ``sign_bits`` (optional uint8 scratch from sign_bits_for) receives the signs of h2 for the next layer's masked dgrad."""
    call("clift_xyz_head_first2_x6_fwd", ptr(xa), ptr(W0), _pitch(W0), ptr(b0), ptr(W1), _pitch(W1), ptr(b1), M, ptr(h2), 256, ptr(sign_bits), stream())


def first2_x6_bwd(M, d, W1, W0, b0, xa, gW0, gb0):
    """One clift_xyz_head_first2_x6_bwd launch (fp32x6 mode): first2_bwd with the 256 x 256 product on the split kernels."""
    call("clift_xyz_head_first2_x6_bwd", ptr(d), d.shape[1], ptr(W1), _pitch(W1), ptr(W0), _pitch(W0), ptr(b0), ptr(xa), M, ptr(gW0), _pitch(gW0), ptr(gb0),
         stream())


def first2_x6_wgrad(M, d, W0, b0, xa, gW1, gb1):
    """One clift_xyz_head_first2_x6_wgrad launch (fp32x6 mode): first2_wgrad on the split weight-gradient kernel."""
    call("clift_xyz_head_first2_x6_wgrad", ptr(d), d.shape[1], ptr(W0), _pitch(W0), ptr(b0), ptr(xa), M, ptr(gW1), _pitch(gW1), ptr(gb1), stream())


def last2_x6(M, h, W, b, Wo, bo, hidden, out, ldo, col_off):
    """One clift_xyz_head_last2_x6_fwd launch (fp32x6 mode): last2 with the 256 x 256 layer on the split kernel; the output layer's partial
    sums travel through a stream-ordered scratch buffer (256 B per row)."""
    nbytes = 256 * M
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=h.device)
    if _lib._launch_stream is not None:       # launched on a side stream (Branches): the allocator must not recycle it earlier
        ws.record_stream(_lib._launch_stream)
    call("clift_xyz_head_last2_x6_fwd", ptr(h), h.shape[1], ptr(W), _pitch(W), ptr(b), ptr(Wo), _pitch(Wo), ptr(bo), Wo.shape[0], M, ptr(hidden), 256,
         C.c_void_p(out.data_ptr() + 4 * col_off), ldo, ptr(ws), nbytes, stream())


def first2_wgrad(M, d, W0, b0, xa, gW1, gb1):
    """One clift_xyz_head_first2_wgrad launch: gW1 += d^T relu(xa[:, :3] W0^T + b0), gb1 += column sums of d."""
    call("clift_xyz_head_first2_wgrad", ptr(d), d.shape[1], ptr(W0), _pitch(W0), ptr(b0), ptr(xa), M, ptr(gW1), _pitch(gW1), ptr(gb1), stream())


def _row_softmax_inplace(out, M, n_out, ldo, col_off):
    o = C.c_void_p(out.data_ptr() + 4 * col_off)
    call("clift_rows_act_fwd", o, ldo, M, n_out, 2, o, ldo, stream())


def out_layer_fwd(M, h, W, b, out, ldo, col_off, act):
    """One clift_out_layer_fwd launch: out[:, col_off:col_off+no] = act(h W^T + b), no <= 32, act 0 / 2 (row softmax) -- one read of h."""
    call("clift_out_layer_fwd", ptr(h), h.shape[1], ptr(W), _pitch(W), ptr(b), W.shape[0], M, C.c_void_p(out.data_ptr() + 4 * col_off), ldo, act, stream())


class HeadActs(list):
    """The hidden activations an xyz head keeps for its backward (acts[i] = output of layer i, or None where the backward re-derives it), plus
    -- fp32x6 mode -- the sign bytes a persistent forward left for activation i (``signs[i]``, from sign_bits_for): held HERE, next to the tensor
    they describe and for exactly as long, instead of as an ad-hoc attribute on the tensor object (which a view or a clone would drop silently)."""

    def __init__(self, *a):
        super().__init__(*a)
        self.signs = {}

    def sign_bits_of(self, i, M):
        """The sign bytes of activation i, or None; checked against the activation they were written with."""
        sb = self.signs.get(i)
        if sb is None:
            return None
        h = self[i]
        if h is None or h.shape[0] != M or sb.numel() != (M + 31) // 32 * 1024:
            raise _lib.CliftError("sign bytes and activation of an xyz head do not come from the same forward")
        return sb


def xyz_mlp_fwd(layers, xa, M, out, ldo, col_off=0, keep_first=True, out_act=0):
    """layers = [(W,b)]; xa (M,4); hidden activations returned for the backward; final layer written into
    out[:, col_off:col_off+n_out] with row pitch ldo -- as logits (``out_act`` 0) or after a softmax over the row (2: the semantic head,
    tensoRF.py:37,593; applied inside the output layer's kernel where that exists, else in place by a row-activation launch).
    ``keep_first`` = False: the caller will not run a backward through this head, so the first (K = 3) layer's activation is
    never materialised (acts[0] is None) -- fp32 path, 256-wide heads."""
    dev = xa.device
    acts = HeadActs()
    W0, b0 = layers[0]
    hdt = act_dtype()                     # bf16 mode stores the hidden activations as bf16 (half the HBM stream of these layers)
    rest = layers[1:-1]
    if (FUSE_HEAD_BF16 and MLP_PRECISION == 1 and hdt == torch.bfloat16 and len(layers) >= 4 and W0.shape[0] == 256
            and tuple(layers[1][0].shape) == (256, 256) and tuple(layers[2][0].shape) == (256, 256) and os.environ.get("CLIFT_NO_PERSISTENT") is None):
        # bf16 mode: K = 3 layer + two hidden layers (+ the output layer when it is <= 4 wide and follows directly) in one launch with the
        # activations resident in LDS; they are written (bf16) only for a backward pass
        Wo, bo = layers[-1]
        with_out = len(layers) == 4 and Wo.shape[0] <= 4 and out.dtype == torch.float32
        mk = lambda: torch.empty((M, 256), dtype=torch.bfloat16, device=dev)
        h1, h2 = (mk(), mk()) if keep_first else (None, None)
        h3 = mk() if (keep_first or not with_out) else None
        head_bf16(M, xa, layers[0], layers[1], layers[2], (Wo, bo) if with_out else None, h1, h2, h3, out, ldo, col_off)
        acts += [h1, h2, h3]
        if with_out:
            if out_act == 2:
                _row_softmax_inplace(out, M, Wo.shape[0], ldo, col_off)
            return acts if keep_first else [None]
        h = h3
        rest = layers[3:-1]
    elif (MLP_PRECISION == 0 and len(layers) >= 3 and W0.shape[0] == 256 and tuple(layers[1][0].shape) == (256, 256)
            and os.environ.get("CLIFT_NO_PERSISTENT") is None):
        W1, b1 = layers[1]
        # (with the fused backward the first layer's activation has no reader: both of its uses -- the ReLU mask of the second layer's input
        # gradient and the second layer's weight gradient -- re-derive it from the positions, so it is never written)
        h1 = torch.empty((M, 256), dtype=torch.float32, device=dev) if (keep_first and KEEP_FIRST_ACT) else None
        h = torch.empty((M, 256), dtype=torch.float32, device=dev)
        first2(M, xa, W0, b0, W1, b1, h1, h)
        acts += [h1, h]
        rest = layers[2:-1]
    elif (MLP_PRECISION == 2 and len(layers) >= 3 and W0.shape[0] == 256 and tuple(layers[1][0].shape) == (256, 256)
            and (not keep_first or not KEEP_FIRST_ACT) and os.environ.get("CLIFT_NO_PERSISTENT") is None
            and os.environ.get("CLIFT_X6_TILED") is None):
        # fp32x6: the same fusion with the 256 x 256 layer on the split kernels; the first activation is never written (the backward, if any,
        # re-derives it: first2_bwd / first2_wgrad)
        W1, b1 = layers[1]
        h = torch.empty((M, 256), dtype=torch.float32, device=dev)
        # a backward will run: the kernel also leaves the SIGNS of its output (32 B per row) for the next layer's masked dgrad, which then
        # does not stream the 1 KB-per-row activation a second time just to test it against zero
        sb1 = sign_bits_for(M, dev) if (keep_first and len(layers) >= 4) else None
        first2_x6(M, xa, W0, b0, W1, b1, h, sb1)
        acts += [None, h]
        if sb1 is not None:
            acts.signs[1] = sb1
        rest = layers[2:-1]
    else:
        h = torch.empty((M, W0.shape[0]), dtype=hdt, device=dev)
        call("clift_linear_k3_fwd", ptr(xa), ptr(W0), _pitch(W0), ptr(b0), M, W0.shape[0], 1, ptr(h), h.shape[1], int(hdt == torch.bfloat16), stream())
        acts.append(h)
    Wo, bo = layers[-1]
    fuse_out = (MLP_PRECISION in (0, 2) and len(rest) >= 1 and Wo.shape[0] <= 4 and tuple(rest[-1][0].shape) == (256, 256)
                and h.dtype == torch.float32 and out.dtype == torch.float32 and os.environ.get("CLIFT_NO_PERSISTENT") is None
                and (MLP_PRECISION == 0 or os.environ.get("CLIFT_X6_TILED") is None))
    for li_, (W, b) in enumerate(rest):
        if fuse_out and li_ == len(rest) - 1:
            # last hidden layer + the narrow output layer in one launch; the hidden activation is written only for a backward
            hn = torch.empty((M, 256), dtype=torch.float32, device=dev) if keep_first else None
            (last2_x6 if MLP_PRECISION == 2 else last2)(M, h, W, b, Wo, bo, hn, out, ldo, col_off)
            acts.append(hn)
            if out_act == 2:
                _row_softmax_inplace(out, M, Wo.shape[0], ldo, col_off)
            return acts if keep_first else [None]
        hn = torch.empty((M, W.shape[0]), dtype=hdt, device=dev)
        # (fp32x6, a backward will run, and this is not the last hidden layer -- whose output the output layer's backward reads as values: sign bytes too)
        sb = (sign_bits_for(M, dev) if (keep_first and MLP_PRECISION == 2 and li_ < len(rest) - 1 and tuple(W.shape) == (256, 256) and h.dtype == torch.float32
                                        and os.environ.get("CLIFT_X6_TILED") is None and os.environ.get("CLIFT_NO_PERSISTENT") is None) else None)
        gemm(M, W.shape[0], W.shape[1], h, h.shape[1], W, _pitch(W), hn, hn.shape[1], bias=b, act=1, sign_bits=sb)
        acts.append(hn)
        if sb is not None:
            acts.signs[len(acts) - 1] = sb
        h = hn
    W, b = Wo, bo
    if (MLP_PRECISION in (0, 2) and W.shape[0] <= 32 and W.shape[1] == 256 and h.dtype == torch.float32 and h.shape[1] == 256 and out.dtype == torch.float32
            and _pitch(W) % 4 == 0 and b is not None and os.environ.get("CLIFT_NO_PERSISTENT") is None):
        # the E <= 32 output layer (+ the row softmax) as one stream over the hidden activation
        out_layer_fwd(M, h, W, b, out, ldo, col_off, out_act)
        return acts if keep_first else [None]
    gemm(M, W.shape[0], W.shape[1], h, h.shape[1], W, _pitch(W), out, ldo, bias=b, c_off=col_off)
    if out_act == 2:
        _row_softmax_inplace(out, M, W.shape[0], ldo, col_off)
    # no backward through this head: nothing is retained, every hidden activation goes back to the (stream-ordered) allocator as soon
    # as the next layer has been enqueued -- a frame render at 65536 rays per chunk holds ~9 GiB per hidden layer otherwise
    return acts if keep_first else [None]


def xyz_mlp_bwd(layers, glayers, xa, acts, dpre, M, keep=None):
    """dpre (M, ld) = gradient w.r.t. the last layer's pre-activation output (ld % 4 == 0, pad zero).
    ``keep``: list that receives every temporary (so a caller running this on a side stream controls their lifetime)."""
    dev = xa.device
    d = dpre
    n = len(layers)
    if acts[0] is None and len(acts) != n - 1:
        raise _lib.CliftError("backward through an xyz head whose forward ran with keep_first=False (head not named in grad_heads)")
    keep = keep if keep is not None else []
    keep.append(d)
    for li in range(n - 1, 0, -1):
        W, b = layers[li]
        gW, gb = glayers[li]
        h = acts[li - 1]
        no, ni = W.shape
        if li == 1 and (h is None or (not KEEP_FIRST_ACT and MLP_PRECISION in (0, 2) and no == 256 and ni == 256 and tuple(layers[0][0].shape) == (256, 3) and
                                      d.dtype == torch.float32 and h.dtype == torch.float32 and d.shape[1] % 4 == 0 and
                                      os.environ.get("CLIFT_NO_PERSISTENT") is None)):
            # second layer: its weight gradient (over the first layer's activation -- regenerated from the positions when the forward did not keep
            # it), then its input gradient formed and consumed by the first layer's weight gradient in one launch
            if no != 256 or ni != 256 or tuple(layers[0][0].shape) != (256, 3) or d.shape[1] != 256 or d.dtype != torch.float32:
                raise _lib.CliftError("backward through an xyz head whose forward ran with keep_first=False (head not named in grad_heads)")
            x6_ends = MLP_PRECISION == 2 and os.environ.get("CLIFT_X6_TILED") is None
            if h is None:
                (first2_x6_wgrad if x6_ends else first2_wgrad)(M, d, layers[0][0], layers[0][1], xa, gW, gb)
            else:
                wgrad(no, ni, M, d, d.shape[1], h, h.shape[1], gW, gb)
            (first2_x6_bwd if x6_ends else first2_bwd)(M, d, W, layers[0][0], layers[0][1], xa, *glayers[0])
            return
        if (li == 1 and MLP_PRECISION == 1 and FIRST2_BF16_BWD_FUSED and h is not None and h.dtype == torch.bfloat16 and d.dtype == torch.bfloat16
                and no == 256 and ni == 256 and tuple(layers[0][0].shape) == (256, 3) and d.shape[1] == 256 and h.shape[1] == 256 and M >= 4096
                and os.environ.get("CLIFT_NO_PERSISTENT") is None):
            # bf16 mode, second layer: its weight gradient, then its input gradient formed and consumed by the K = 3 layer's weight gradient in one launch
            wgrad(no, ni, M, d, d.shape[1], h, h.shape[1], gW, gb)
            call("clift_xyz_head_first2_bf16_bwd", ptr(d), 256, ptr(W), _pitch(W), ptr(h), 256, ptr(xa), M, ptr(glayers[0][0]), _pitch(glayers[0][0]),
                 ptr(glayers[0][1]), stream())
            return
        dn = torch.empty((M, ni), dtype=act_dtype(), device=dev)        # bf16 mode: hidden gradients are bf16-stored as well
        if (li == n - 1 and ni == 256 and no <= 32 and d.shape[1] <= 32 and d.shape[1] % 4 == 0 and M >= 4096 and
                MLP_PRECISION in (0, 2) and h.dtype == torch.float32 and d.dtype == torch.float32 and dn.dtype == torch.float32):
            # output layer: weight gradient and masked input gradient in one pass over the hidden activation
            call("clift_out_layer_bwd", ptr(d), d.shape[1], no, ptr(W), _pitch(W), ptr(h), h.shape[1], M, ptr(dn), ni, ptr(gW), _pitch(gW),
                 ptr(gb), stream())
            d = dn
            keep.append(d)
            continue
        if (li == n - 1 and ni == 256 and no <= 32 and d.shape[1] <= 32 and d.shape[1] % 4 == 0 and M >= 4096 and MLP_PRECISION == 1 and OUT_BWD_BF16_FUSED and
                h.dtype == torch.bfloat16 and d.dtype == torch.float32 and dn.dtype == torch.bfloat16 and h.shape[1] == 256 and _pitch(W) >= 256):
            # bf16 mode: the same one pass over the (bf16-stored) hidden activation; the input gradient leaves bf16-stored
            call("clift_out_layer_bwd_nh", ptr(d), d.shape[1], no, ptr(W), _pitch(W), ptr(h), 256, 256, M, ptr(dn), 256, ptr(gW), _pitch(gW),
                 ptr(gb), 1, stream())
            d = dn
            keep.append(d)
            continue
        wgrad(no, ni, M, d, d.shape[1], h, h.shape[1], gW, gb)
        sb = acts.sign_bits_of(li - 1, M) if isinstance(acts, HeadActs) else None
        if (sb is not None and MLP_PRECISION == 2 and no == 256 and ni == 256 and d.dtype == torch.float32 and d.shape[1] % 4 == 0
                and os.environ.get("CLIFT_X6_TILED") is None):
            gemm(M, ni, no, d, d.shape[1], W, _pitch(W), dn, ni, b_trans=1, sign_bits=sb)      # mask = the signs the forward left behind
        else:
            gemm(M, ni, no, d, d.shape[1], W, _pitch(W), dn, ni, b_trans=1, mask=h, ldmask=h.shape[1])
        d = dn
        keep.append(d)
    gW, gb = glayers[0]
    call("clift_linear_k3_bwd", ptr(xa), ptr(d), d.shape[1], M, layers[0][0].shape[0], ptr(gW), _pitch(gW), ptr(gb),
         int(d.dtype == torch.bfloat16), stream())


# ----------------------------------------------------------------------------- heads on their own VM grids (tensoRF.py:70-83,142-156)
def _grid_precision():
    # the grid heads' MLPs are short (27 -> 128 -> 128 -> C, 27 -> 256 -> 256 -> E): in bf16 mode they stay exact (no bf16-stored forms are built
    # for them); in fp32x6 mode engine.gemm sends their one 256 x 256 layer to the split kernels like any other
    return _Precision(0 if MLP_PRECISION == 1 else MLP_PRECISION)


def feat_mlp_fwd(layers, X, M, out, ldo, col_off=0, keep=True, out_act=0):
    """MLP over a feature matrix X (M, ldx) -- ldx = the first layer's row pitch, pad columns zero -- instead of over positions: hidden layers
    through engine.gemm (bias + ReLU), the last layer into out[:, col_off:col_off + n_out] (+ row softmax when ``out_act`` is 2).
    Returns the hidden activations (for feat_mlp_bwd) or None."""
    h, acts = X, []
    for W, b in layers[:-1]:
        hn = torch.empty((M, W.shape[0]), dtype=torch.float32, device=X.device)
        gemm(M, W.shape[0], _pitch(W), h, h.shape[1], W, _pitch(W), hn, hn.shape[1], bias=b, act=1)
        acts.append(hn)
        h = hn
    Wo, bo = layers[-1]
    if (Wo.shape[0] <= 32 and h.shape[1] == 256 and _pitch(Wo) % 4 == 0 and _pitch(Wo) >= 256 and MLP_PRECISION in (0, 2)
            and os.environ.get("CLIFT_NO_PERSISTENT") is None):
        out_layer_fwd(M, h, Wo, bo, out, ldo, col_off, out_act)
    else:
        gemm(M, Wo.shape[0], _pitch(Wo), h, h.shape[1], Wo, _pitch(Wo), out, ldo, bias=bo, c_off=col_off)
        if out_act == 2:
            _row_softmax_inplace(out, M, Wo.shape[0], ldo, col_off)
    return acts if keep else None


def feat_mlp_bwd(layers, glayers, X, acts, dpre, M, keep=None):
    """Backward of feat_mlp_fwd: dpre (M, ld) = gradient w.r.t. the last layer's pre-activation output (ld % 4 == 0, pad zero); accumulates the
    weight / bias gradients and returns the gradient w.r.t. X (M, ldx)."""
    keep = keep if keep is not None else []
    d = dpre
    keep.append(d)
    for li in range(len(layers) - 1, -1, -1):
        W, b = layers[li]
        gW, gb = glayers[li]
        h = acts[li - 1] if li > 0 else X
        no, kin = W.shape[0], h.shape[1]
        wgrad(no, kin, M, d, d.shape[1], h, kin, gW, gb)
        dn = torch.empty((M, kin), dtype=torch.float32, device=X.device)
        if li > 0:
            gemm(M, kin, no, d, d.shape[1], W, _pitch(W), dn, kin, b_trans=1, mask=h, ldmask=kin)
        else:
            gemm(M, kin, no, d, d.shape[1], W, _pitch(W), dn, kin, b_trans=1)
        d = dn
        keep.append(d)
    return d


def grid_head_fwd(model, views, ctx, prefix, nets, keep, out_act=0):
    """A head whose MLP reads the 27 features of its own VM grid (compute_feature, tensoRF.py:127-134, on the semantic / instance tables) for
    the chunk's active samples: product gather (the appearance head's kernel: generic in the component count), basis Linear, then every net of
    ``nets`` = [(layers, out, ldo, col_off)] on the same features (the instance head's fast and slow MLPs).  Returns what the backward needs."""
    M, dev = ctx.M, ctx.rays.device
    with _grid_precision():
        vm = vm_struct(views, prefix, ctx.res)
        nc = 3 * vm.comps
        F = torch.empty((M, nc), dtype=torch.float32, device=dev)
        call("clift_app_gather_fwd", C.byref(ctx.ms), C.byref(vm), ptr(ctx.rays), ptr(ctx.jitter), ptr(ctx.act_idx), M, ptr(F), ptr(ctx.xa), stream())
        Wb = views[f"{prefix}_basis_mat.weight"]
        nf, ldf = Wb.shape[0], (Wb.shape[0] + 3) // 4 * 4
        feat = torch.zeros((M, ldf), dtype=torch.float32, device=dev)          # (pad columns stay zero: the first layer's K is its row pitch)
        with exact_fp32():
            gemm(M, nf, nc, F, nc, Wb, _pitch(Wb), feat, ldf)
        acts = [feat_mlp_fwd(layers, feat, M, out, ldo, col_off, keep, out_act) for layers, out, ldo, col_off in nets]
    return dict(prefix=prefix, F=F if keep else None, feat=feat if keep else None, acts=acts)


def grid_head_bwd(model, views, gviews, ctx, st, nets, keep):
    """Backward of grid_head_fwd.  ``nets`` = [(layers, glayers, acts, dpre)] for the nets that receive a gradient (their input gradients add up
    on the shared features); then the basis matrix's gradient, dF = dfeat Wb and the scatter into the head's tables."""
    M, dev, prefix = ctx.M, ctx.rays.device, st["prefix"]
    with _grid_precision():
        dfeat = None
        for layers, glayers, acts, dpre in nets:
            d = feat_mlp_bwd(layers, glayers, st["feat"], acts, dpre, M, keep)
            dfeat = d if dfeat is None else dfeat.add_(d)
        Wb, gWb = views[f"{prefix}_basis_mat.weight"], gviews[f"{prefix}_basis_mat.weight"]
        nf, nc = Wb.shape
        call("clift_wgrad_narrow", ptr(dfeat), dfeat.shape[1], nf, ptr(st["F"]), nc, nc, M, ptr(gWb), _pitch(gWb), None, 0, stream())
        dF = torch.empty((M, nc), dtype=torch.float32, device=dev)
        with exact_fp32():
            gemm(M, nc, nf, dfeat, dfeat.shape[1], Wb, _pitch(Wb), dF, nc, b_trans=1)
        model.xcd_workspace_for(prefix)
        vm = vm_struct(views, prefix, ctx.res)
        g = vm_grad_struct(model, gviews, prefix)
        call("clift_app_gather_bwd", C.byref(ctx.ms), C.byref(vm), C.byref(g), ptr(ctx.rays), ptr(ctx.jitter), ptr(ctx.act_idx), M, ptr(dF),
             ptr(ctx.xa) if APP_SCATTER_XA else None, stream())
        vm_grad_finish(model, gviews, prefix, g)
        keep.extend([dfeat, dF])


# ----------------------------------------------------------------------------- forward
class RenderCtx:
    pass


INT_MAX = 2 ** 31 - 1
_rows_limit = {}
_limit_owner = None          # the sync-free chunk whose row count the limit currently holds


def rows_limit(dev):
    """The library's dynamic row limit of this device: int32 [limit, overflow], bound once (clift_bind_rows_limit).  limit is INT_MAX
    (no effect) except inside a sync-free pass; ``reset_rows_limit`` restores that."""
    key = (dev.type, dev.index)
    t = _rows_limit.get(key)
    if t is None:
        t = torch.tensor([INT_MAX, 0], dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        call("clift_bind_rows_limit", ptr(t))
        _rows_limit[key] = t
    return t


_grad_shard_record = {}


def grad_shard_record(dev):
    """The library's gradient-shard record of this device (5 x int64: range start, bytes, shard 0, shard stride, enabled), bound once
    (clift_bind_grad_shards).  All zero -- disabled -- except inside a trainer pass (clift_grad_shards_begin ... clift_grad_shards_fold)."""
    key = (dev.type, dev.index)
    t = _grad_shard_record.get(key)
    if t is None:
        t = torch.zeros(5, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        call("clift_bind_grad_shards", ptr(t))
        _grad_shard_record[key] = t
    return t


def reset_rows_limit(dev=None):
    """Back to 'no limit' (stream-ordered).  Called at the end of every sync-free pass, and defensively by the point-wise utilities."""
    global _limit_owner
    _limit_owner = None
    for key, t in _rows_limit.items():
        if dev is None or key == (dev.type, dev.index):
            t[0:1].fill_(INT_MAX)


def _density_march(model, renderer, rays, jitter, cap=None):
    views = model.named_views()
    N = rays.shape[0]
    S = int(renderer.n_samples)
    dev = rays.device
    ms = march_struct(renderer, model)
    res = grid_res(views)
    vd = vm_struct(views, "density", res)
    sigma = torch.empty((N, S), dtype=torch.float32, device=dev)
    alpha, T, w = torch.empty_like(sigma), torch.empty_like(sigma), torch.empty_like(sigma)
    ray_out = torch.empty((N, 8), dtype=torch.float32, device=dev)
    n_active = torch.empty((N,), dtype=torch.int32, device=dev)
    st = stream()
    call("clift_density_fwd", C.byref(ms), C.byref(vd), ptr(rays), ptr(jitter), N, ptr(sigma), st)
    call("clift_march_fwd", C.byref(ms), ptr(rays), ptr(jitter), N, ptr(sigma), ptr(alpha), ptr(T), ptr(w), ptr(ray_out),
         ptr(n_active), st)
    ray_start = torch.empty((N + 1,), dtype=torch.int32, device=dev)
    if cap is None:
        # a synchronising chunk inside a sync-free pass (a pass still learning its capacity, e.g. the segment term whose ray count
        # changes every step): the library-wide row limit may still hold the PREVIOUS capped chunk's count, which would silently clamp
        # every per-sample kernel of this chunk -- put "no limit" back first (stream-ordered)
        global _limit_owner
        lim_t = _rows_limit.get((dev.type, dev.index))
        if lim_t is not None and _limit_owner is not None:
            lim_t[0:1].fill_(INT_MAX)
            _limit_owner = None
        call("clift_scan_counts", ptr(n_active), N, ptr(ray_start), st)
        M = int(ray_start[N].item())        # the one host sync of the chunk: sizes the active-sample buffers
        act_idx = torch.empty((max(M, 1),), dtype=torch.int32, device=dev)
        call("clift_compact_fill", ptr(w), ptr(ray_start), N, S, float(renderer.raymarch_weight_thres), ptr(act_idx), st)
    else:
        # sync-free: buffers and grids are sized by the capacity, the true count stays on the device (rows_limit()[0]) where every
        # per-sample kernel clamps to it; an overflow (count > cap: samples dropped) is recorded in rows_limit()[1] for the caller
        if MLP_PRECISION == 1:
            raise _lib.CliftError("sync-free steps (cap=...) are built for the fp32 / fp32x6 paths only")
        lim = rows_limit(dev)
        M = max(int(cap), 1)
        call("clift_scan_counts_capped", ptr(n_active), N, ptr(ray_start), M, ptr(lim), C.c_void_p(lim.data_ptr() + 4), st)
        act_idx = torch.empty((M,), dtype=torch.int32, device=dev)
        call("clift_compact_fill_capped", ptr(w), ptr(ray_start), N, S, float(renderer.raymarch_weight_thres), ptr(act_idx), M, st)
    ctx = RenderCtx()
    ctx.ms, ctx.res, ctx.N, ctx.S, ctx.M = ms, res, N, S, M
    ctx.capped = cap is not None
    if ctx.capped:
        _limit_owner = ctx
    ctx.rays, ctx.jitter = rays, jitter
    ctx.alpha, ctx.T, ctx.w, ctx.ray_out, ctx.ray_start, ctx.act_idx = alpha, T, w, ray_out, ray_start, act_idx
    ctx.sigma = sigma
    return ctx


def _check_rays(rays, jitter):
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise ValueError(f"rays must be (N,8) [o,d,near,far], got {tuple(rays.shape)}")
    _lib.f32(rays, "rays")
    rays = rays.contiguous()
    if jitter is not None:
        jitter = _lib.f32(jitter, "jitter").reshape(-1).contiguous()
        if jitter.shape[0] != rays.shape[0]:
            raise ValueError("jitter must have one entry per ray")
    return rays, jitter


def render_forward(model, renderer, rays, jitter, white_bg, want_rgb=True, want_sem=True, want_inst=True, grad_heads=("app", "sem", "fast", "slow"),
                   cap=None, want_dist=True):
    """Full renderer.forward (reference renderer.py:80-176).  Returns dict of outputs and the backward context.
    ``grad_heads``: the heads a backward pass may be run through ("app", "sem", "fast", "slow"); a head that is not named retains
    no activations (the training main pass never differentiates the instance heads, T:155; inference none)."""
    rays, jitter = _check_rays(rays, jitter)
    views = model.named_views()
    ctx = _density_march(model, renderer, rays, jitter, cap)
    N, S, M = ctx.N, ctx.S, ctx.M
    dev = rays.device
    st = stream()
    Ccls = model.num_semantic_classes
    D = model.dim_feature_instance if (want_inst and model.render_instance_mlp is not None) else 0
    softmax_mode = 1 if renderer.semantic_weight_mode == "softmax" else 0
    ctx.white_bg, ctx.softmax_mode, ctx.stop_grad = int(bool(white_bg)), softmax_mode, int(bool(renderer.stop_semantic_grad))
    ctx.C, ctx.D = Ccls, D
    ctx.rgb_s = ctx.sem_s = ctx.inst_s = None
    if M > 0:
        xa = torch.empty((M, 4), dtype=torch.float32, device=dev)
        ctx.xa = xa
        front = None
        if want_rgb:       # the gather also produces xa, which every other head reads: it stays on the main stream
            va = vm_struct(views, "appearance", ctx.res)
            nc = 3 * va.comps
            Wb = views["appearance_basis_mat.weight"]
            W1 = views["render_appearance_mlp.mlp.0.weight"]
            nf, ldx = Wb.shape[0], _pitch(W1)
            if (APP_FRONT_FUSED and nf <= 28 and va.comps in (16, 32, 48) and Wb.shape[1] == nc and _pitch(Wb) % 4 == 0 and ldx % 4 == 0 and ldx <= 512
                    and os.environ.get("CLIFT_NO_PERSISTENT") is None):
                # the appearance front end as one launch: gather -> basis -> MLP input rows (the products are written only for a backward)
                ldf = 28
                F = torch.empty((M, nc), dtype=torch.float32, device=dev) if "app" in grad_heads else None
                feat = torch.empty((M, ldf), dtype=torch.float32, device=dev)
                with _app_precision():              # bf16 mode: the MLP's input rows leave the front end bf16-stored
                    x_bf16 = MLP_PRECISION == 1 and ldx % 8 == 0
                X = torch.empty((M, ldx), dtype=torch.bfloat16 if x_bf16 else torch.float32, device=dev)
                call("clift_app_front_fwd_x", C.byref(ctx.ms), C.byref(va), ptr(rays), ptr(jitter), ptr(ctx.act_idx), M, ptr(Wb), _pitch(Wb), nf,
                     model.pe_feat, model.pe_view, ptr(xa), ptr(feat), ldf, ptr(X), ldx, ptr(F), int(x_bf16), st)
                front = (feat, ldf, X)
            else:
                F = torch.empty((M, nc), dtype=torch.float32, device=dev)
                call("clift_app_gather_fwd", C.byref(ctx.ms), C.byref(va), ptr(rays), ptr(jitter), ptr(ctx.act_idx), M, ptr(F), ptr(xa), st)
            ctx.F = F
        else:
            call("clift_active_xyz", C.byref(ctx.ms), ptr(rays), ptr(jitter), ptr(ctx.act_idx), M, ptr(xa), st)

        def app_chain(keep):
            with _app_precision():
                _app_chain_fwd(keep)

        def _app_chain_fwd(keep):
            Wb = views["appearance_basis_mat.weight"]
            nf, nc = Wb.shape
            (W1, b1), (W2, b2), (W3, b3) = _lin_params(None, "render_appearance_mlp.mlp", views)
            ldx = _pitch(W1)
            hdt = act_dtype() if ldx % 8 == 0 else torch.float32      # bf16 mode: encoded input and hidden activations bf16-stored
            if front is not None:
                feat, ldf, X = front
            else:
                ldf = (nf + 3) // 4 * 4
                feat = torch.empty((M, ldf), dtype=torch.float32, device=dev)
                with exact_fp32():          # (the basis Linear is not one of the bf16 layers in any mode)
                    gemm(M, nf, nc, ctx.F, nc, Wb, _pitch(Wb), feat, ldf)
                X = torch.empty((M, ldx), dtype=hdt, device=dev)
                call("clift_app_encode_fwd", ptr(feat), ldf, nf, model.pe_feat, model.pe_view, ptr(rays), ptr(ctx.act_idx), S, M,
                     ptr(X), ldx, int(hdt == torch.bfloat16), stream())
            H1 = torch.empty((M, W1.shape[0]), dtype=hdt, device=dev)
            gemm(M, W1.shape[0], ldx, X, ldx, W1, ldx, H1, H1.shape[1], bias=b1, act=1)
            rgb_s = torch.empty((M, 3), dtype=torch.float32, device=dev)
            if (MLP_PRECISION in (0, 2) and hdt == torch.float32 and tuple(W2.shape) == (128, 128) and W3.shape[0] <= 4 and W3.shape[1] == 128
                    and os.environ.get("CLIFT_NO_PERSISTENT") is None):
                # second hidden layer + output layer + sigmoid in one launch; H2 is written only for a backward
                H2 = torch.empty((M, 128), dtype=torch.float32, device=dev) if "app" in grad_heads else None
                if MLP_PRECISION == 2 and os.environ.get("CLIFT_X6_TILED") is None:        # (only inside _app_precision() with APP_X6)
                    call("clift_app_head_last2_x6_fwd", ptr(H1), 128, ptr(W2), _pitch(W2), ptr(b2), ptr(W3), _pitch(W3), ptr(b3), W3.shape[0], M,
                         ptr(H2), 128, ptr(rgb_s), 3, 1, stream())
                else:
                    app_last2(M, H1, W2, b2, W3, b3, H2, rgb_s)
            elif (MLP_PRECISION == 1 and hdt == torch.bfloat16 and H1.dtype == torch.bfloat16 and tuple(W2.shape) == (128, 128) and W3.shape[0] <= 4
                    and W3.shape[1] == 128 and M >= 64 and os.environ.get("CLIFT_NO_PERSISTENT") is None):
                # bf16 mode: the same pair of layers over the bf16-stored activation (csrc/layer_nb16.hip)
                H2 = torch.empty((M, 128), dtype=torch.bfloat16, device=dev) if "app" in grad_heads else None
                call("clift_app_head_last2_bf16_fwd", ptr(H1), 128, ptr(W2), _pitch(W2), ptr(b2), ptr(W3), _pitch(W3), ptr(b3), W3.shape[0], M,
                     ptr(H2), 128, ptr(rgb_s), 3, 1, stream())
            else:
                H2 = torch.empty((M, W2.shape[0]), dtype=hdt, device=dev)
                gemm(M, W2.shape[0], W2.shape[1], H1, H1.shape[1], W2, _pitch(W2), H2, H2.shape[1], bias=b2, act=1)
                pre = torch.empty((M, 3), dtype=torch.float32, device=dev)
                gemm(M, 3, W3.shape[1], H2, H2.shape[1], W3, _pitch(W3), pre, 3, bias=b3)
                call("clift_rows_act_fwd", ptr(pre), 3, M, 3, 1, ptr(rgb_s), 3, stream())
                keep.append(pre)
            ctx.rgb_s = rgb_s
            if SOAK_KEEP:
                ctx.soak = dict(feat=feat, X=X, H1=H1)
            if "app" in grad_heads:
                ctx.feat, ctx.ldf, ctx.nf, ctx.X, ctx.ldx, ctx.H1, ctx.H2 = feat, ldf, nf, X, ldx, H1, H2
            else:
                ctx.F = None

        def sem_chain(keep):
            sem_layers = _lin_params(None, "render_semantic_mlp.mlp", views)
            sem_s = torch.empty((M, Ccls), dtype=torch.float32, device=dev)
            sm = 2 if model.render_semantic_mlp.softmax else 0
            if model.semantic_plane is not None:           # the head on its own VM grid
                ctx.sem_grid = grid_head_fwd(model, views, ctx, "semantic", [(sem_layers, sem_s, Ccls, 0)], "sem" in grad_heads, sm)
            else:
                ctx.sem_acts = xyz_mlp_fwd(sem_layers, xa, M, sem_s, Ccls, keep_first="sem" in grad_heads, out_act=sm)
            ctx.sem_s = sem_s

        br = Branches(model=model)
        if want_rgb:
            br.run(0, app_chain)
        if want_sem:
            br.run(1, sem_chain)
        if D > 0:
            E = model.render_instance_mlp.output_channels
            ctx.inst_s = torch.empty((M, D), dtype=torch.float32, device=dev)

            def fast_chain(keep):
                ctx.inst_fast_acts = xyz_mlp_fwd(_lin_params(None, "render_instance_mlp.mlp", views), xa, M, ctx.inst_s, D, 0,
                                                 keep_first="fast" in grad_heads)

            def slow_chain(keep):
                ctx.inst_slow_acts = xyz_mlp_fwd(_lin_params(None, "render_instance_mlp.slow_mlp", views), xa, M, ctx.inst_s, D, E,
                                                 keep_first="slow" in grad_heads)
            if model.instance_plane is not None:           # the head on its own VM grid: fast and slow nets read the same features
                nets = [(_lin_params(None, "render_instance_mlp.mlp", views), ctx.inst_s, D, 0)]
                if model.slow_fast_mode:
                    nets.append((_lin_params(None, "render_instance_mlp.slow_mlp", views), ctx.inst_s, D, E))
                br.run(2, lambda keep: setattr(ctx, "inst_grid", grid_head_fwd(model, views, ctx, "instance", nets, "fast" in grad_heads or "slow" in grad_heads)))
            else:
                br.run(2, fast_chain)
                if model.slow_fast_mode:
                    br.run(3, slow_chain)
        br.join()
    # (the sum kernel writes every (ray, channel) when there are heads to sum; only a chunk without active samples needs the zeros)
    fresh = torch.empty if M > 0 else torch.zeros
    rgb_raw = fresh((N, 3), dtype=torch.float32, device=dev) if want_rgb else None
    rgb_map = torch.empty((N, 3), dtype=torch.float32, device=dev) if want_rgb else None
    sem_raw = fresh((N, Ccls), dtype=torch.float32, device=dev) if want_sem else None
    sem_map = torch.empty((N, Ccls), dtype=torch.float32, device=dev) if want_sem else None
    inst_map = fresh((N, D), dtype=torch.float32, device=dev) if D > 0 else None
    ctx.w_feat = None
    if M > 0 and renderer.semantic_weight_mode == "argmax" and (want_sem or D > 0):
        # renderer.py:142-143: the semantic / instance sums take the one-hot of each ray's heaviest sample (of ALL its samples: a heaviest
        # sample below the threshold is not in the list and the ray's sums stay 0, as in the reference, whose heads are 0 there); the colours
        # keep the weights -- two passes of the compositing kernels, one per weight array
        ctx.w_feat = torch.zeros_like(ctx.w).scatter_(1, ctx.w.argmax(dim=1, keepdim=True), 1.0)
        if want_rgb:
            call("clift_composite_fwd", ptr(ctx.w), ptr(ctx.ray_start), ptr(ctx.act_idx), N, 0, 0, ptr(ctx.rgb_s), None, None, ptr(ctx.ray_out), 0,
                 ctx.white_bg, ptr(rgb_raw), ptr(rgb_map), None, None, None, st)
        call("clift_composite_fwd", ptr(ctx.w_feat), ptr(ctx.ray_start), ptr(ctx.act_idx), N, Ccls if want_sem else 0, D, None, ptr(ctx.sem_s),
             ptr(ctx.inst_s), ptr(ctx.ray_out), softmax_mode, 0, None, None, ptr(sem_raw), ptr(sem_map), ptr(inst_map), st)
    elif M > 0:
        call("clift_composite_fwd", ptr(ctx.w), ptr(ctx.ray_start), ptr(ctx.act_idx), N, Ccls if want_sem else 0, D,
             ptr(ctx.rgb_s), ptr(ctx.sem_s), ptr(ctx.inst_s), ptr(ctx.ray_out), softmax_mode, ctx.white_bg,
             ptr(rgb_raw), ptr(rgb_map), ptr(sem_raw), ptr(sem_map), ptr(inst_map), st)
    else:   # no active sample in the chunk (reference: the `if appearance_mask.any()` branch is skipped)
        call("clift_composite_fwd", None, ptr(ctx.ray_start), ptr(ctx.act_idx), N, Ccls if want_sem else 0, D,
             None, None, None, ptr(ctx.ray_out), softmax_mode, ctx.white_bg,
             ptr(rgb_raw), ptr(rgb_map), ptr(sem_raw), ptr(sem_map), ptr(inst_map), st)
    ctx.rgb_raw, ctx.sem_raw = rgb_raw, sem_raw
    ctx.want = (want_rgb, want_sem, D > 0)
    out = dict(rgb=rgb_map, semantics=sem_map, instances=inst_map, depth=ctx.ray_out[:, 1],
               dist_reg=ctx.ray_out[:, 5].mean() if want_dist else None, opacity=ctx.ray_out[:, 0])     # (the trainer never reads the VALUE of the regulariser)
    return out, ctx


# ----------------------------------------------------------------------------- backward
def render_backward(model, ctx, gviews, g_rgb=None, g_sem=None, g_inst=None, g_dist=None, density_grad=True,
                    slow_grad=False, before_density=None):
    """Accumulate parameter gradients into ``gviews`` (name -> tensor with the parameter's layout).
    g_rgb (N,3), g_sem (N,C), g_inst (N,D): output gradients or None; g_dist: device scalar tensor or None.
    ``before_density``: called once, after every head chain (MLPs, appearance tables) has been issued on the current stream and before
    the density backward -- the data-parallel trainer starts the all-reduce of those gradients there, under the density backward."""
    views = model.named_views()
    N, S, M = ctx.N, ctx.S, ctx.M
    dev = ctx.rays.device
    st = stream()
    global _limit_owner
    if getattr(ctx, "capped", False):
        if _limit_owner is not ctx:        # another chunk was marched since: put THIS chunk's row count back (device-to-device, 4 bytes)
            rows_limit(dev)[0:1].copy_(ctx.ray_start[N:N + 1])
            _limit_owner = ctx
    elif _limit_owner is not None:         # a synchronising chunk's backward behind a sync-free chunk: its rows are not to be clamped
        reset_rows_limit(dev)
    Ccls, D = ctx.C, ctx.D
    want_rgb, want_sem, want_inst = ctx.want
    g_rgb = g_rgb.contiguous() if (g_rgb is not None and want_rgb) else None
    g_sem = g_sem.contiguous() if (g_sem is not None and want_sem) else None
    g_inst = g_inst.contiguous() if (g_inst is not None and want_inst) else None
    if density_grad:
        # d loss / d weight (N, S) and d loss / d opacity (N): zero except at the active samples -- one fill for both
        gbuf = torch.zeros((N * S + N,), dtype=torch.float32, device=dev)
        g_w, g_op = gbuf[:N * S].view(N, S), gbuf[N * S:]
    else:       # (feature passes: the weights carry no gradient, nobody reads these)
        g_w = torch.empty((N, S), dtype=torch.float32, device=dev)
        g_op = torch.empty((N,), dtype=torch.float32, device=dev)
    if M > 0 and (g_rgb is not None or g_sem is not None or g_inst is not None):
        ge = torch.empty((N, 3 + Ccls + D), dtype=torch.float32, device=dev)
        E_inst = model.render_instance_mlp.output_channels if (D > 0 and model.render_instance_mlp is not None) else 0
        act_fused = COMPOSITE_ACT_FUSED and (D == 0 or (E_inst >= 1 and 2 * E_inst >= D))
        if act_fused:
            # the heads' output activations (sigmoid / softmax / identity) are taken back inside the compositing backward: what it writes are the
            # zero-padded pre-activation gradient rows the heads' backward kernels start from
            ldp_sem, ldp_inst = (Ccls + 3) // 4 * 4, (E_inst + 3) // 4 * 4
            d_rgb = torch.empty((M, 4), dtype=torch.float32, device=dev) if g_rgb is not None else None
            d_sem = torch.empty((M, ldp_sem), dtype=torch.float32, device=dev) if g_sem is not None else None
            d_inst = torch.empty((M, ldp_inst), dtype=torch.float32, device=dev) if g_inst is not None else None
            d_inst_slow = (torch.empty((M, ldp_inst), dtype=torch.float32, device=dev)
                           if (g_inst is not None and model.slow_fast_mode and slow_grad) else None)
            sem_kind = 2 if (model.render_semantic_mlp is not None and model.render_semantic_mlp.softmax) else 0
            if getattr(ctx, "w_feat", None) is not None:
                # "argmax" weights (render_forward): the colours against the weights (with the weight / opacity gradients), then the semantic and
                # instance rows against the one-hot array, which carries no gradient
                call("clift_composite_bwd_act", ptr(ctx.w), ptr(ctx.act_idx), N, S, M, 0, 0, ptr(ctx.rgb_s), None, None, ptr(ctx.rgb_raw), None, 0,
                     ctx.white_bg, 1, ptr(g_rgb), None, None, ptr(ge), 0, ptr(d_rgb), 4, None, 4, None, None, 4, 0, ptr(g_w), ptr(g_op), st)
                call("clift_composite_bwd_act", ptr(ctx.w_feat), ptr(ctx.act_idx), N, S, M, Ccls if want_sem else 0, D, None, ptr(ctx.sem_s),
                     ptr(ctx.inst_s), None, ptr(ctx.sem_raw), ctx.softmax_mode, 0, 1, None, ptr(g_sem), ptr(g_inst), ptr(ge), sem_kind, None, 4,
                     ptr(d_sem), ldp_sem, ptr(d_inst), ptr(d_inst_slow), ldp_inst, E_inst, None, None, st)
            else:
                call("clift_composite_bwd_act", ptr(ctx.w), ptr(ctx.act_idx), N, S, M, Ccls if want_sem else 0, D,
                     ptr(ctx.rgb_s), ptr(ctx.sem_s), ptr(ctx.inst_s), ptr(ctx.rgb_raw), ptr(ctx.sem_raw), ctx.softmax_mode,
                     ctx.white_bg, ctx.stop_grad, ptr(g_rgb), ptr(g_sem), ptr(g_inst), ptr(ge), sem_kind, ptr(d_rgb), 4, ptr(d_sem), ldp_sem,
                     ptr(d_inst), ptr(d_inst_slow), ldp_inst, E_inst, ptr(g_w), ptr(g_op), st)
        else:
            d_rgb = torch.empty((M, 3), dtype=torch.float32, device=dev) if g_rgb is not None else None
            d_sem = torch.empty((M, Ccls), dtype=torch.float32, device=dev) if g_sem is not None else None
            d_inst = torch.empty((M, D), dtype=torch.float32, device=dev) if g_inst is not None else None
            if getattr(ctx, "w_feat", None) is not None:      # "argmax" weights: as above
                call("clift_composite_bwd", ptr(ctx.w), ptr(ctx.ray_start), ptr(ctx.act_idx), N, S, M, 0, 0, ptr(ctx.rgb_s), None, None,
                     ptr(ctx.rgb_raw), None, 0, ctx.white_bg, 1, ptr(g_rgb), None, None, ptr(ge), ptr(d_rgb), None, None, ptr(g_w), ptr(g_op), st)
                call("clift_composite_bwd", ptr(ctx.w_feat), ptr(ctx.ray_start), ptr(ctx.act_idx), N, S, M, Ccls if want_sem else 0, D, None,
                     ptr(ctx.sem_s), ptr(ctx.inst_s), None, ptr(ctx.sem_raw), ctx.softmax_mode, 0, 1, None, ptr(g_sem), ptr(g_inst), ptr(ge), None,
                     ptr(d_sem), ptr(d_inst), None, None, st)
            else:
                call("clift_composite_bwd", ptr(ctx.w), ptr(ctx.ray_start), ptr(ctx.act_idx), N, S, M, Ccls if want_sem else 0, D,
                     ptr(ctx.rgb_s), ptr(ctx.sem_s), ptr(ctx.inst_s), ptr(ctx.rgb_raw), ptr(ctx.sem_raw), ctx.softmax_mode,
                     ctx.white_bg, ctx.stop_grad, ptr(g_rgb), ptr(g_sem), ptr(g_inst), ptr(ge), ptr(d_rgb), ptr(d_sem), ptr(d_inst),
                     ptr(g_w), ptr(g_op), st)
        br = Branches(model=model)
        if density_grad:
            model.xcd_workspace_for("density")
        if d_rgb is not None:
            model.xcd_workspace_for("appearance")

        # ---------------- appearance head
        def app_chain(keep):
            with _app_precision():
                _app_chain_bwd(keep)

        def _app_chain_bwd(keep):
            app = _lin_params(None, "render_appearance_mlp.mlp", views)
            gapp = _lin_params(None, "render_appearance_mlp.mlp", gviews)
            (W1, b1), (W2, b2), (W3, b3) = app
            (gW1, gb1), (gW2, gb2), (gW3, gb3) = gapp
            if act_fused:
                dpre = d_rgb
            else:
                dpre = torch.empty((M, 4), dtype=torch.float32, device=dev)
                call("clift_rows_act_bwd", ptr(ctx.rgb_s), 3, ptr(d_rgb), 3, M, 3, 1, ptr(dpre), 4, stream())
            H1, H2, X, ldx = ctx.H1, ctx.H2, ctx.X, ctx.ldx
            n2 = W3.shape[1]
            dH2 = torch.empty((M, n2), dtype=H2.dtype, device=dev)
            if (APP_OUT_BWD_FUSED and H2.dtype == torch.float32 and n2 % 32 == 0 and n2 <= 256 and W3.shape[0] <= 4 and M >= 4096
                    and _pitch(W3) >= n2 and _pitch(gW3) >= n2 and os.environ.get("CLIFT_NO_PERSISTENT") is None):
                # output layer: weight gradient and masked input gradient in one pass over the hidden activation (as in the xyz heads)
                call("clift_out_layer_bwd_nh", ptr(dpre), 4, W3.shape[0], ptr(W3), _pitch(W3), ptr(H2), n2, n2, M, ptr(dH2), n2, ptr(gW3), _pitch(gW3),
                     ptr(gb3), 0, stream())
            elif (APP_OUT_BWD_FUSED and MLP_PRECISION == 1 and H2.dtype == torch.bfloat16 and n2 == 128 and W3.shape[0] <= 4 and dpre.shape[1] == 4
                    and _pitch(W3) >= 128 and _pitch(gW3) >= 128 and os.environ.get("CLIFT_NO_PERSISTENT") is None):
                # bf16 mode: the same one pass over the bf16-stored activation; the input gradient leaves bf16-stored
                call("clift_out_layer_bwd_n128_bf16", ptr(dpre), 4, W3.shape[0], ptr(W3), _pitch(W3), ptr(H2), 128, M, ptr(dH2), 128, ptr(gW3), _pitch(gW3),
                     ptr(gb3), stream())
            else:
                wgrad(3, n2, M, dpre, 4, H2, n2, gW3, gb3)
                gemm(M, n2, 3, dpre, 4, W3, _pitch(W3), dH2, n2, b_trans=1, mask=H2, ldmask=n2)
            n1 = W2.shape[1]
            wgrad(n2, n1, M, dH2, n2, H1, n1, gW2, gb2)
            dH1 = torch.empty((M, n1), dtype=H1.dtype, device=dev)
            gemm(M, n1, n2, dH2, n2, W2, _pitch(W2), dH1, n1, b_trans=1, mask=H1, ldmask=n1)
            wgrad(n1, ldx, M, dH1, n1, X, ldx, gW1, gb1)
            dX = torch.empty((M, ldx), dtype=torch.float32, device=dev)
            gemm(M, ldx, n1, dH1, n1, W1, ldx, dX, ldx, b_trans=1)
            nf, ldf = ctx.nf, ctx.ldf
            Wb, gWb = views["appearance_basis_mat.weight"], gviews["appearance_basis_mat.weight"]
            nc = Wb.shape[1]
            va = vm_struct(views, "appearance", ctx.res)
            ga = vm_grad_struct(model, gviews, "appearance")
            dfeat = torch.empty((M, ldf), dtype=torch.float32, device=dev)
            call("clift_app_encode_bwd", ptr(ctx.feat), ldf, nf, model.pe_feat, ptr(dX), ldx, M, ptr(dfeat), ldf, stream())
            call("clift_wgrad_narrow", ptr(dfeat), ldf, nf, ptr(ctx.F), nc, nc, M, ptr(gWb), _pitch(gWb), None, 0, stream())
            dF = torch.empty((M, nc), dtype=torch.float32, device=dev)
            with exact_fp32():              # (the basis Linear is not one of the bf16 layers in any mode)
                gemm(M, nc, nf, dfeat, ldf, Wb, _pitch(Wb), dF, nc, b_trans=1)
            call("clift_app_gather_bwd", C.byref(ctx.ms), C.byref(va), C.byref(ga), ptr(ctx.rays), ptr(ctx.jitter),
                 ptr(ctx.act_idx), M, ptr(dF), ptr(ctx.xa) if APP_SCATTER_XA else None, stream())
            vm_grad_finish(model, gviews, "appearance", ga)
            keep.extend([dpre, dH2, dH1, dX, dfeat, dF])

        # ---------------- semantic head
        def sem_chain(keep):
            if act_fused:
                dpre = d_sem
            else:
                ldp = (Ccls + 3) // 4 * 4
                dpre = torch.empty((M, ldp), dtype=torch.float32, device=dev)
                kind = 2 if model.render_semantic_mlp.softmax else 0
                call("clift_rows_act_bwd", ptr(ctx.sem_s), Ccls, ptr(d_sem), Ccls, M, Ccls, kind, ptr(dpre), ldp, stream())
            if getattr(ctx, "sem_grid", None) is not None:
                st_ = ctx.sem_grid
                if st_["F"] is None:
                    raise _lib.CliftError("backward through a grid head whose forward ran without keeping its activations (head not named in grad_heads)")
                grid_head_bwd(model, views, gviews, ctx, st_, [(_lin_params(None, "render_semantic_mlp.mlp", views),
                                                                _lin_params(None, "render_semantic_mlp.mlp", gviews), st_["acts"][0], dpre)], keep)
                return
            xyz_mlp_bwd(_lin_params(None, "render_semantic_mlp.mlp", views), _lin_params(None, "render_semantic_mlp.mlp", gviews),
                        ctx.xa, ctx.sem_acts, dpre, M, keep)

        # ---------------- instance heads
        def inst_chain(prefix, acts, off):
            def run(keep):
                E = model.render_instance_mlp.output_channels
                if act_fused:
                    dpre = d_inst if off == 0 else d_inst_slow
                else:
                    ldp = (E + 3) // 4 * 4
                    dpre = torch.empty((M, ldp), dtype=torch.float32, device=dev)
                    call("clift_rows_act_bwd", None, 0, C.c_void_p(d_inst.data_ptr() + 4 * off), D, M, E, 0, ptr(dpre), ldp, stream())
                xyz_mlp_bwd(_lin_params(None, prefix, views), _lin_params(None, prefix, gviews), ctx.xa, acts, dpre, M, keep)
            return run

        if d_rgb is not None:
            br.run(0, app_chain)
        if d_sem is not None:
            br.run(1, sem_chain)
        if d_inst is not None and getattr(ctx, "inst_grid", None) is not None:
            def inst_grid_chain(keep):
                st_ = ctx.inst_grid
                if st_["F"] is None:
                    raise _lib.CliftError("backward through a grid head whose forward ran without keeping its activations (head not named in grad_heads)")
                E_ = model.render_instance_mlp.output_channels
                ldp = (E_ + 3) // 4 * 4

                def pre(off, fused):
                    if act_fused:
                        return fused
                    dp = torch.empty((M, ldp), dtype=torch.float32, device=dev)
                    call("clift_rows_act_bwd", None, 0, C.c_void_p(d_inst.data_ptr() + 4 * off), D, M, E_, 0, ptr(dp), ldp, stream())
                    return dp
                nets = [(_lin_params(None, "render_instance_mlp.mlp", views), _lin_params(None, "render_instance_mlp.mlp", gviews), st_["acts"][0], pre(0, d_inst))]
                if model.slow_fast_mode and slow_grad:
                    nets.append((_lin_params(None, "render_instance_mlp.slow_mlp", views), _lin_params(None, "render_instance_mlp.slow_mlp", gviews),
                                 st_["acts"][1], pre(E_, d_inst_slow)))
                grid_head_bwd(model, views, gviews, ctx, st_, nets, keep)
            br.run(2, inst_grid_chain)
        elif d_inst is not None:
            br.run(2, inst_chain("render_instance_mlp.mlp", ctx.inst_fast_acts, 0))
            if model.slow_fast_mode and slow_grad:
                br.run(3, inst_chain("render_instance_mlp.slow_mlp", ctx.inst_slow_acts, model.render_instance_mlp.output_channels))
        if density_grad:
            if before_density is not None:
                if br.enabled:
                    br.join()                 # (side-stream mode: the head chains have to be on the current stream first)
                before_density()
                before_density = None
            br.run(4, lambda keep: _density_backward(model, ctx, views, gviews, g_w, g_op, g_dist, keep))
            density_grad = False
        br.join()
    elif g_rgb is not None and ctx.white_bg:
        # no active samples but the white background still routes d rgb into the opacity
        inside = ((ctx.rgb_raw >= 0) & (ctx.rgb_raw <= 1)).to(torch.float32)
        g_op = -(g_rgb * inside).sum(-1)
    if before_density is not None:
        before_density()
    # ---------------- density path (when it was not already run as a branch above)
    if density_grad:
        _density_backward(model, ctx, views, gviews, g_w, g_op, g_dist, [])


def _density_backward(model, ctx, views, gviews, g_w, g_op, g_dist, keep):
    """Transmittance backward (needs only g_w / g_opacity from the compositing backward, so it runs concurrently with
    the MLP chains) and the scatter into the density tables."""
    N, S = ctx.N, ctx.S
    dsigma = torch.empty((N, S), dtype=torch.float32, device=ctx.rays.device)
    call("clift_march_bwd", C.byref(ctx.ms), ptr(ctx.rays), ptr(ctx.jitter), N, ptr(ctx.alpha), ptr(ctx.T), ptr(ctx.w),
         ptr(ctx.ray_out), ptr(g_w), ptr(g_op), ptr(g_dist), ptr(dsigma), stream())
    vd = vm_struct(views, "density", ctx.res)
    gd = vm_grad_struct(model, gviews, "density")
    call("clift_density_bwd", C.byref(ctx.ms), C.byref(vd), C.byref(gd), ptr(ctx.rays), ptr(ctx.jitter), N, ptr(dsigma),
         ptr(ctx.sigma) if DENS_BWD_SIGMA else None, stream())
    vm_grad_finish(model, gviews, "density", gd)
    keep.append(dsigma)


# ----------------------------------------------------------------------------- instance / segment feature passes
def feature_forward(model, renderer, rays, jitter, head, grad_heads=("app", "sem", "fast", "slow"), cap=None, want_xyz=True):
    """renderer.py:178-217 (head='instance') / :259-300 (head='semantic'): density and weights carry no gradient,
    only the head does.  ``grad_heads`` as in render_forward."""
    rays, jitter = _check_rays(rays, jitter)
    views = model.named_views()
    ctx = _density_march(model, renderer, rays, jitter, cap)
    N, S, M = ctx.N, ctx.S, ctx.M
    dev = rays.device
    st = stream()
    ctx.white_bg, ctx.stop_grad = 0, 1
    ctx.softmax_mode = 1 if (head == "semantic" and renderer.semantic_weight_mode == "softmax") else 0
    Ccls = model.num_semantic_classes if head == "semantic" else 0
    D = model.dim_feature_instance if head == "instance" else 0
    ctx.C, ctx.D = Ccls, D
    ctx.rgb_s = ctx.sem_s = ctx.inst_s = None
    ctx.rgb_raw = None
    if M > 0:
        xa = torch.empty((M, 4), dtype=torch.float32, device=dev)
        ctx.xa = xa
        call("clift_active_xyz", C.byref(ctx.ms), ptr(rays), ptr(jitter), ptr(ctx.act_idx), M, ptr(xa), st)
        if head == "semantic":
            layers = _lin_params(None, "render_semantic_mlp.mlp", views)
            ctx.sem_s = torch.empty((M, Ccls), dtype=torch.float32, device=dev)
            sm = 2 if model.render_semantic_mlp.softmax else 0
            if model.semantic_plane is not None:
                ctx.sem_grid = grid_head_fwd(model, views, ctx, "semantic", [(layers, ctx.sem_s, Ccls, 0)], "sem" in grad_heads, sm)
            else:
                ctx.sem_acts = xyz_mlp_fwd(layers, xa, M, ctx.sem_s, Ccls, keep_first="sem" in grad_heads, out_act=sm)
        elif model.instance_plane is not None:
            E = model.render_instance_mlp.output_channels
            ctx.inst_s = torch.empty((M, D), dtype=torch.float32, device=dev)
            nets = [(_lin_params(None, "render_instance_mlp.mlp", views), ctx.inst_s, D, 0)]
            if model.slow_fast_mode:
                nets.append((_lin_params(None, "render_instance_mlp.slow_mlp", views), ctx.inst_s, D, E))
            ctx.inst_grid = grid_head_fwd(model, views, ctx, "instance", nets, "fast" in grad_heads or "slow" in grad_heads)
        else:
            E = model.render_instance_mlp.output_channels
            ctx.inst_s = torch.empty((M, D), dtype=torch.float32, device=dev)
            br = Branches(model=model)

            def fast_chain(keep):
                ctx.inst_fast_acts = xyz_mlp_fwd(_lin_params(None, "render_instance_mlp.mlp", views), xa, M, ctx.inst_s, D, 0,
                                                 keep_first="fast" in grad_heads)

            def slow_chain(keep):
                ctx.inst_slow_acts = xyz_mlp_fwd(_lin_params(None, "render_instance_mlp.slow_mlp", views), xa, M, ctx.inst_s, D, E,
                                                 keep_first="slow" in grad_heads)
            br.run(2, fast_chain)
            if model.slow_fast_mode:
                br.run(3, slow_chain)
            br.join()
    fresh = torch.empty if M > 0 else torch.zeros          # (as in render_forward: the sums are written in full unless the chunk is empty)
    sem_raw = fresh((N, Ccls), dtype=torch.float32, device=dev) if Ccls else None
    sem_map = torch.empty((N, Ccls), dtype=torch.float32, device=dev) if Ccls else None
    inst_map = fresh((N, D), dtype=torch.float32, device=dev) if D else None
    call("clift_composite_fwd", ptr(ctx.w), ptr(ctx.ray_start), ptr(ctx.act_idx), N, Ccls, D, None, ptr(ctx.sem_s), ptr(ctx.inst_s),
         ptr(ctx.ray_out), ctx.softmax_mode, 0, None, None, ptr(sem_raw), ptr(sem_map), ptr(inst_map), st)
    ctx.sem_raw = sem_raw
    ctx.want = (False, head == "semantic", head == "instance")
    if head == "instance":
        # renderer.py:213-215 (the trainer asks for the points only where its loss reads them: two elementwise launches otherwise unused)
        xyz = rays[:, 0:3] + ctx.ray_out[:, 1:2] * rays[:, 3:6] if want_xyz else None
        return (inst_map, xyz), ctx
    return sem_map, ctx


def feature_backward(model, ctx, gviews, g_out, slow_grad=False):
    if ctx.want[1]:
        render_backward(model, ctx, gviews, g_sem=g_out, density_grad=False)
    else:
        render_backward(model, ctx, gviews, g_inst=g_out, density_grad=False, slow_grad=slow_grad)


# ----------------------------------------------------------------------------- point-wise utilities (reference field API)
def _points(xyz):
    x = _lib.f32(xyz, "xyz").reshape(-1, xyz.shape[-1]).contiguous()
    if x.shape[1] < 3:
        raise ValueError("xyz must have >= 3 columns")
    return x


@torch.no_grad()
def density_points(model, xyz, activation=True):
    """TensorVMSplit.compute_density / compute_density_without_activation on normalised points (no gradient)."""
    x = _points(xyz)
    reset_rows_limit(x.device)
    views = model.named_views()
    vd = vm_struct(views, "density", grid_res(views))
    out = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    call("clift_density_points", C.byref(vd), ptr(x), x.shape[1], x.shape[0], float(model.splus_density_shift), int(bool(activation)),
         ptr(out), stream())
    return out


@torch.no_grad()
def appearance_feature_points(model, xyz):
    """TensorVMSplit.compute_appearance_feature: VM products + basis Linear (no gradient)."""
    x = _points(xyz)
    reset_rows_limit(x.device)
    n = x.shape[0]
    views = model.named_views()
    va = vm_struct(views, "appearance", grid_res(views))
    nc = 3 * va.comps
    F = torch.empty((n, nc), dtype=torch.float32, device=x.device)
    call("clift_vm_products_points", C.byref(va), ptr(x), x.shape[1], n, ptr(F), stream())
    Wb = views["appearance_basis_mat.weight"]
    out = torch.empty((n, Wb.shape[0]), dtype=torch.float32, device=x.device)
    gemm(n, Wb.shape[0], nc, F, nc, Wb, _pitch(Wb), out, out.shape[1])
    return out


@torch.no_grad()
def grid_feature_points(model, prefix, xyz):
    """TensorVMSplit.compute_semantic_feature / compute_instance_feature for a head on its own VM grid: VM products + basis Linear (no gradient)."""
    x = _points(xyz)
    reset_rows_limit(x.device)
    n = x.shape[0]
    views = model.named_views()
    vm = vm_struct(views, prefix, grid_res(views))
    nc = 3 * vm.comps
    F = torch.empty((n, nc), dtype=torch.float32, device=x.device)
    call("clift_vm_products_points", C.byref(vm), ptr(x), x.shape[1], n, ptr(F), stream())
    Wb = views[f"{prefix}_basis_mat.weight"]
    out = torch.empty((n, Wb.shape[0]), dtype=torch.float32, device=x.device)
    with exact_fp32():
        gemm(n, Wb.shape[0], nc, F, nc, Wb, _pitch(Wb), out, out.shape[1])
    return out


@torch.no_grad()
def feat_mlp_points(seq, feats):
    """Evaluate a feature-input MLP head (grid heads: MLPRenderSemanticFeature / MLPRenderInstanceFeature on 27 features) on arbitrary rows."""
    f = _lib.f32(feats, "features").reshape(-1, feats.shape[-1]).contiguous()
    reset_rows_limit(f.device)
    M = f.shape[0]
    layers = [(m.weight, m.bias) for m in seq if isinstance(m, torch.nn.Linear)]
    ldx = _pitch(layers[0][0])
    X = torch.zeros((M, ldx), dtype=torch.float32, device=f.device)
    X[:, :f.shape[1]] = f
    out = torch.empty((M, layers[-1][0].shape[0]), dtype=torch.float32, device=f.device)
    with _grid_precision():
        feat_mlp_fwd(layers, X, M, out, out.shape[1], keep=False)
    return out


@torch.no_grad()
def xyz_mlp_points(seq, xyz):
    """Evaluate an xyz MLP head on arbitrary points (inference utility, no gradient)."""
    x = _points(xyz)
    reset_rows_limit(x.device)
    M = x.shape[0]
    xa = torch.zeros((M, 4), dtype=torch.float32, device=x.device)
    xa[:, :3] = x[:, :3]
    layers = [(m.weight, m.bias) for m in seq if isinstance(m, torch.nn.Linear)]
    out = torch.empty((M, layers[-1][0].shape[0]), dtype=torch.float32, device=x.device)
    xyz_mlp_fwd(layers, xa, M, out, out.shape[1], keep_first=False)
    return out


@torch.no_grad()
def appearance_mlp_points(module, viewdirs, features):
    """MLPRenderFeature.forward(viewdirs, features) (tensoRF.py:400-411), no gradient."""
    f = _lib.f32(features, "features").reshape(-1, features.shape[-1]).contiguous()
    d = _lib.f32(viewdirs, "viewdirs").reshape(-1, 3).contiguous()
    reset_rows_limit(f.device)
    n, nf = f.shape
    (W1, b1), (W2, b2), (W3, b3) = [(m.weight, m.bias) for m in module.mlp if isinstance(m, torch.nn.Linear)]
    ldx = _pitch(W1)
    X = torch.empty((n, ldx), dtype=torch.float32, device=f.device)
    call("clift_app_encode_points", ptr(f), nf, nf, module.pe_feat, module.pe_view, ptr(d), 3, n, ptr(X), ldx, stream())
    H1 = torch.empty((n, W1.shape[0]), dtype=torch.float32, device=f.device)
    gemm(n, W1.shape[0], ldx, X, ldx, W1, ldx, H1, H1.shape[1], bias=b1, act=1)
    H2 = torch.empty((n, W2.shape[0]), dtype=torch.float32, device=f.device)
    gemm(n, W2.shape[0], W2.shape[1], H1, H1.shape[1], W2, _pitch(W2), H2, H2.shape[1], bias=b2, act=1)
    pre = torch.empty((n, 3), dtype=torch.float32, device=f.device)
    gemm(n, 3, W3.shape[1], H2, H2.shape[1], W3, _pitch(W3), pre, 3, bias=b3)
    out = torch.empty_like(pre)
    call("clift_rows_act_fwd", ptr(pre), 3, n, 3, 1, ptr(out), 3, stream())
    return out
