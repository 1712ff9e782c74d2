"""GPU tests of round 5, second half: the 128-wide appearance MLP in the default arithmetic (fp32x6) on its own persistent split kernels
(csrc/layer_n6.hip) -- every entry point against float64 next to the exact-fp32 kernel it replaces, the launch-independence of a row's bits,
the device-side row limit, and the appearance chain end to end (tensoRF.py:393-411)."""
import pytest
import torch

from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def _rows(M, K, g, scale=1.5):
    """Rows of very different scale (x e^{scale N(0,1)}), about half of the entries zero (post-ReLU activations), pad columns beyond 150 zero when K = 160."""
    A = torch.relu(torch.randn(M, K, generator=g)) * torch.exp(scale * torch.randn(M, 1, generator=g))
    if K == 160:
        A[:, 150:] = 0
    return A.to(DEV)


def _both(fn):
    from contrastive_lift_amd import engine
    out = {}
    for mode in ("fp32", "fp32x6"):
        with engine._Precision(engine._PRECISIONS[mode]):
            out[mode] = fn()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M", [1, 31, 33, 300, 4097, 66001])
@pytest.mark.parametrize("K", [128, 160])
def test_n6_forward_against_fp64(M, K):
    """k_layer_n6 forward (clift_gemm precision 2, N = 128, K in {128, 160}): relu(A W^T + b) with a padded output pitch left untouched; row-max
    relative error against fp64 <= 2e-6 and <= 4x the exact-fp32 kernel's + 2e-7; rows are independent of how many share the launch."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(11 + M + K)
    A = _rows(M + 77, K, g)
    W = (torch.randn(128, K, generator=g) / 12).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    ref = torch.relu(A[:M].double() @ W.double().T + b.double())

    def run():
        C1 = torch.full((M, 132), -7.0, device=DEV)
        engine.gemm(M, 128, K, A, K, W, K, C1, 132, bias=b, act=1)
        C3 = torch.empty((M + 77, 128), device=DEV)
        engine.gemm(M + 77, 128, K, A, K, W, K, C3, 128, bias=b, act=1)
        return C1, C3
    out = _both(run)
    s = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
    e = {k: float(((v[0][:, :128].double() - ref).abs() / s).max()) for k, v in out.items()}
    assert bool((out["fp32x6"][0][:, 128:] == -7.0).all())
    assert e["fp32x6"] <= 2e-6 and e["fp32x6"] <= 4 * e["fp32"] + 2e-7, e
    assert torch.equal(out["fp32x6"][1][:M], out["fp32x6"][0][:, :128])             # a row's bits do not depend on the launch it is in


@pytest.mark.parametrize("M", [1, 33, 4097, 66001])
def test_n6_input_gradients_against_fp64(M):
    """k_layer_n6 dgrad: the masked 128 -> 128 form (mask = the layer's fp32 input activation) and the unmasked 128 -> 160 form (dX of the first
    layer; the weight's pad columns are zero, so are the result's)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(23 + M)
    d = (torch.randn(M, 128, generator=g) * (torch.rand(M, 128, generator=g) > 0.5) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)
    W2 = (torch.randn(128, 128, generator=g) / 12).to(DEV)
    W1 = (torch.randn(128, 160, generator=g) / 12)
    W1[:, 150:] = 0
    W1 = W1.to(DEV)
    H = torch.randn(M, 128, generator=g).to(DEV)
    ref_m = (d.double() @ W2.double()) * (H > 0)
    ref_u = d.double() @ W1.double()

    def run():
        Cm = torch.empty((M, 128), device=DEV)
        engine.gemm(M, 128, 128, d, 128, W2, 128, Cm, 128, b_trans=1, mask=H, ldmask=128)
        Cu = torch.full((M, 160), -7.0, device=DEV)
        engine.gemm(M, 160, 128, d, 128, W1, 160, Cu, 160, b_trans=1)
        return Cm, Cu
    out = _both(run)
    for i, ref in enumerate((ref_m, ref_u)):
        s = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
        e = {k: float(((v[i].double() - ref).abs() / s).max()) for k, v in out.items()}
        assert e["fp32x6"] <= 2e-6 and e["fp32x6"] <= 4 * e["fp32"] + 2e-7, (i, e)
    assert bool((out["fp32x6"][1][:, 150:] == 0).all())


@pytest.mark.parametrize("M", [1, 31, 4096, 5000, 66001, 249000])
@pytest.mark.parametrize("NX", [128, 160])
def test_n6_weight_gradient_against_fp64(M, NX):
    """k_wgrad_n6: gW (128, NX) += dY^T X, gb += column sums of dY as six bf16 products of exactly split operands against float64, next to the
    exact streaming kernel; accumulating (a second launch doubles the result)."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(41 + M + NX)
    dY = (torch.randn(M, 128, generator=g) * (torch.rand(M, 128, generator=g) > 0.5) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)
    X = _rows(M, NX, g, scale=0.5)
    ref = dY.double().cpu().t() @ X.double().cpu()
    rb = dY.double().cpu().sum(0)

    def run():
        gW, gb = torch.zeros(128, NX, device=DEV), torch.zeros(128, device=DEV)
        engine.wgrad(128, NX, M, dY, 128, X, NX, gW, gb)
        gW2, gb2 = gW.clone(), gb.clone()
        engine.wgrad(128, NX, M, dY, 128, X, NX, gW2, gb2)
        return gW, gb, gW2
    out = _both(run)
    sc, sb = float(ref.abs().max()), float(rb.abs().max()) + 1e-30
    e = {k: (float((v[0].double().cpu() - ref).abs().max()) / sc, float((v[1].double().cpu() - rb).abs().max()) / sb) for k, v in out.items()}
    assert e["fp32x6"][0] <= 2e-6 and e["fp32x6"][0] <= 4 * e["fp32"][0] + 2e-7, e
    assert e["fp32x6"][1] <= 4e-6, e
    assert float((out["fp32x6"][2].double().cpu() - 2 * ref).abs().max()) / sc <= 4e-6
    if NX == 160:
        assert bool((out["fp32x6"][0][:, 150:] == 0).all())


@pytest.mark.parametrize("M", [1, 33, 4097, 66001])
@pytest.mark.parametrize("keep", [True, False])
def test_n6_last_two_layers_fused(M, keep):
    """clift_app_head_last2_x6_fwd: H2 = relu(H1 W2^T + b2) (written only when asked for), rgb = sigmoid(H2 W3^T + b3), against float64 and against
    the exact-fp32 launch; rows independent of the launch."""
    import ctypes as C
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(5 + M)
    H1 = _rows(M + 40, 128, g, scale=0.7)
    W2 = (torch.randn(128, 128, generator=g) / 12).to(DEV)
    b2 = (torch.randn(128, generator=g) / 4).to(DEV)
    W3 = (torch.randn(3, 128, generator=g) / 8).to(DEV)
    b3 = torch.randn(3, generator=g).to(DEV)
    h2 = torch.relu(H1.double() @ W2.double().T + b2.double())
    ref = torch.sigmoid(h2 @ W3.double().T + b3.double())

    def run(rows):
        H2 = torch.empty((rows, 128), device=DEV) if keep else None
        rgb = torch.empty((rows, 3), device=DEV)
        call("clift_app_head_last2_x6_fwd", ptr(H1), 128, ptr(W2), 128, ptr(b2), ptr(W3), 128, ptr(b3), 3, rows, ptr(H2), 128, ptr(rgb), 3, 1, stream())
        return H2, rgb
    H2, rgb = run(M)
    H2b, rgbb = run(M + 40)
    ex2 = torch.empty((M, 128), device=DEV)
    exr = torch.empty((M, 3), device=DEV)
    engine.app_last2(M, H1, W2, b2, W3, b3, ex2, exr)
    torch.cuda.synchronize()
    assert float((rgb.double() - ref[:M]).abs().max()) <= 2e-6 + 4 * float((exr.double() - ref[:M]).abs().max())
    if keep:
        s = h2[:M].abs().amax(1, keepdim=True).clamp_min(1e-30)
        e6, e0 = float(((H2.double() - h2[:M]).abs() / s).max()), float(((ex2.double() - h2[:M]).abs() / s).max())
        assert e6 <= 2e-6 and e6 <= 4 * e0 + 2e-7, (e6, e0)
        assert torch.equal(H2b[:M], H2)
    assert torch.equal(rgbb[:M], rgb)


def test_n6_kernels_under_a_device_side_row_limit():
    """Sync-free passes size every launch by a capacity and leave the true row count on the device: the n6 kernels re-balance their row ranges over
    it, write nothing past it, and sum nothing past it."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(77)
    cap, true = 9000, 5321
    A = _rows(cap, 160, g)
    W = (torch.randn(128, 160, generator=g) / 12).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    dY = torch.randn(cap, 128, generator=g).to(DEV)
    lim = engine.rows_limit(A.device)
    with engine._Precision(2):
        ref = torch.empty((true, 128), device=DEV)
        engine.gemm(true, 128, 160, A, 160, W, 160, ref, 128, bias=b, act=1)
        gW0, gb0 = torch.zeros(128, 160, device=DEV), torch.zeros(128, device=DEV)
        engine.wgrad(128, 160, true, dY, 128, A, 160, gW0, gb0)
        try:
            lim[0:1].fill_(true)
            C1 = torch.full((cap, 128), -7.0, device=DEV)
            engine.gemm(cap, 128, 160, A, 160, W, 160, C1, 128, bias=b, act=1)
            gW, gb = torch.zeros(128, 160, device=DEV), torch.zeros(128, device=DEV)
            engine.wgrad(128, 160, cap, dY, 128, A, 160, gW, gb)
        finally:
            engine.reset_rows_limit(A.device)
    torch.cuda.synchronize()
    assert torch.equal(C1[:true], ref) and bool((C1[true:] == -7.0).all())
    assert float((gW - gW0).abs().max()) <= 1e-5 * float(gW0.abs().max()) and float((gb - gb0).abs().max()) <= 1e-5 * float(gb0.abs().max())


def test_fp32x6_appearance_chain_against_the_exact_chain():
    """The default mode end to end with the appearance MLP on the split kernels (engine.APP_X6) against the same pass with the exact-fp32 appearance
    kernels (rounds 3 - 4): colours to fp32 round-off, every gradient the appearance chain reaches within the noise floating-point atomics of
    sums of this length leave between two runs of the SAME arithmetic."""
    from contrastive_lift_amd import engine
    from test_gpu_round5 import _scene
    model, renderer, pool = _scene()
    g = torch.Generator(device="cpu").manual_seed(3)
    rays = pool[torch.randint(0, pool.shape[0], (4096,), generator=g).to(DEV)].contiguous()
    jit = torch.rand(4096, generator=g).to(DEV)
    cot = torch.randn((4096, 3), generator=g).to(DEV)
    assert engine.MLP_PRECISION == 2
    res = {}
    try:
        for flag in (True, False):
            engine.APP_X6 = flag
            model.grad_flat.zero_()
            o, ctx = engine.render_forward(model, renderer, rays, jit, False, grad_heads=("app",), want_sem=False, want_inst=False)
            engine.render_backward(model, ctx, model.named_grad_views(), g_rgb=cot)
            gv = {k: v.detach().clone() for k, v in model.named_grad_views().items()}
            res[flag] = (o["rgb"].clone(), gv)
    finally:
        engine.APP_X6 = True
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6
    for k in ("render_appearance_mlp.mlp.0.weight", "render_appearance_mlp.mlp.0.bias", "render_appearance_mlp.mlp.2.weight", "render_appearance_mlp.mlp.4.weight",
              "render_appearance_mlp.mlp.4.bias", "appearance_basis_mat.weight", "appearance_plane.0", "appearance_line.2", "density_plane.1"):
        a, b = res[True][1][k].double(), res[False][1][k].double()
        assert float(b.abs().max()) > 0, k
        rel = float((a - b).norm() / b.norm())
        assert rel <= 2e-4, (k, rel)          # (measured 3e-5 on the first layer's weight gradient: sums of 3e5 terms through floating-point atomics)
