"""Round-4 GPU parity tests: the fp32x6 forms of the fused ends of an xyz head (ABI 14; csrc/layer_x6.hip, csrc/layer_x6w.hip)

  * ``clift_xyz_head_last2_x6_fwd``   -- last hidden layer + E <= 4 output layer, the output layer applied to the tile in registers;
  * ``clift_xyz_head_first2_x6_bwd``  -- second layer's input gradient consumed in-kernel by the K = 3 layer's weight gradient;
  * ``clift_xyz_head_first2_x6_wgrad`` -- second layer's weight gradient with its input regenerated from the positions;

each against float64 next to the exact-fp32 entry point it mirrors, on ragged row counts (single row, one short of / one past a 32-row
tile, a row-range boundary, the instance-pass and main-pass bench sizes), with padded pitches, accumulation and the device-side row limit;
and the trainer's exception path: a pass that dies between ``grad_shards_begin`` and ``fold`` leaves nothing behind.
"""
import pytest
import torch

from conftest import grad_close
from test_gpu_parity import DEV
from test_gpu_round3 import _first2_case

pytestmark = pytest.mark.gpu


# ============================================================================ last hidden layer + narrow output layer
def _last2_case(M, E, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, 256, generator=g) * (torch.rand(M, 256, generator=g) > 0.3)
    W = torch.randn(256, 256, generator=g) / 16
    b = 0.3 * torch.randn(256, generator=g)
    Wo = torch.randn(E, 256, generator=g) / 16
    bo = torch.randn(E, generator=g)
    h = torch.relu(A.double() @ W.double().t() + b.double())
    return {k: v.to(DEV) for k, v in dict(A=A, W=W, b=b, Wo=Wo, bo=bo).items()}, h, h @ Wo.double().t() + bo.double()


@pytest.mark.parametrize("M,E", [(1, 3), (31, 3), (32, 1), (33, 4), (4097, 3), (62003, 3), (249001, 2)])
def test_last2_x6_forward_against_fp64(M, E):
    """hidden = relu(A W^T + b) and out = hidden Wo^T + bo in one launch pair.  Against float64: hidden within the band of the plain fp32x6
    layer (2e-6 row-max relative), out within 4e-6 of its row scale next to the exact fused kernel's own error; the hidden activation is
    bit-identical to the plain fp32x6 layer's (same arithmetic, only the epilogue differs); columns of `out` outside [col, col + E) and its
    pad are untouched; with hidden = NULL the outputs have the same bits; a row's bits do not depend on the rows sharing its launch."""
    from contrastive_lift_amd import engine
    t, h_ref, o_ref = _last2_case(M, E, 70 + M + E)
    out = torch.full((M, 6), -7.0, device=DEV)
    hid = torch.empty(M, 256, device=DEV)
    engine.last2_x6(M, t["A"], t["W"], t["b"], t["Wo"], t["bo"], hid, out, 6, 1)
    out2 = torch.full((M, 6), -7.0, device=DEV)
    engine.last2_x6(M, t["A"], t["W"], t["b"], t["Wo"], t["bo"], None, out2, 6, 1)
    oe = torch.full((M, 6), -7.0, device=DEV)
    engine.last2(M, t["A"], t["W"], t["b"], t["Wo"], t["bo"], None, oe, 6, 1)          # exact fused kernel
    plain = torch.empty(M, 256, device=DEV)
    prev = engine.set_mlp_precision("fp32x6")
    try:
        engine.gemm(M, 256, 256, t["A"], 256, t["W"], 256, plain, 256, bias=t["b"], act=1)
    finally:
        engine.set_mlp_precision(prev)
    torch.cuda.synchronize()
    assert torch.equal(hid, plain)
    assert torch.equal(out, out2)
    assert bool((out[:, 0] == -7.0).all()) and bool((out[:, 1 + E:] == -7.0).all())
    sh = h_ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert float(((hid.double().cpu() - h_ref).abs() / sh).max()) <= 2e-6
    so = (h_ref.abs() @ t["Wo"].double().cpu().abs().t() + t["bo"].double().cpu().abs()).clamp_min(1e-30)     # scale of the sum's terms
    e6 = float(((out[:, 1:1 + E].double().cpu() - o_ref).abs() / so).max())
    e0 = float(((oe[:, 1:1 + E].double().cpu() - o_ref).abs() / so).max())
    assert e6 <= 2e-6 and e6 <= 4 * e0 + 2e-7, (e6, e0)
    if M >= 33:          # the first rows again as part of a shorter launch (other tile / range split): same bits
        k = M // 2 + 5
        out3 = torch.full((k, 6), -7.0, device=DEV)
        engine.last2_x6(k, t["A"][:k], t["W"], t["b"], t["Wo"], t["bo"], None, out3, 6, 1)
        torch.cuda.synchronize()
        assert torch.equal(out3, out[:k])


def test_last2_x6_under_a_device_side_row_limit_and_bad_arguments():
    from contrastive_lift_amd import _lib, engine
    from contrastive_lift_amd._lib import call, ptr, stream
    cap, M, E = 9000, 5003, 3
    t, h_ref, o_ref = _last2_case(cap, E, 9)
    t["A"][M:] = float("nan")
    out = torch.full((cap, 4), -7.0, device=DEV)
    lim = engine.rows_limit(out.device)
    lim[0:1].fill_(M)
    try:
        engine.last2_x6(cap, t["A"], t["W"], t["b"], t["Wo"], t["bo"], None, out, 4, 0)
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(out.device)
    assert bool((out[M:] == -7.0).all()) and bool(torch.isfinite(out[:M, :E]).all())
    so = (h_ref[:M].abs() @ t["Wo"].double().cpu().abs().t() + t["bo"].double().cpu().abs()).clamp_min(1e-30)
    assert float(((out[:M, :E].double().cpu() - o_ref[:M]).abs() / so).max()) <= 1e-6
    ws = torch.empty(256 * 100, dtype=torch.uint8, device=DEV)
    o = torch.zeros(100, 4, device=DEV)
    with pytest.raises(_lib.CliftError, match="E must be"):
        call("clift_xyz_head_last2_x6_fwd", ptr(t["A"]), 256, ptr(t["W"]), 256, ptr(t["b"]), ptr(t["Wo"]), 256, ptr(t["bo"]), 5, 100, None, 256, ptr(o), 4,
             ptr(ws), ws.numel(), stream())
    with pytest.raises(_lib.CliftError, match="workspace"):
        call("clift_xyz_head_last2_x6_fwd", ptr(t["A"]), 256, ptr(t["W"]), 256, ptr(t["b"]), ptr(t["Wo"]), 256, ptr(t["bo"]), 3, 100, None, 256, ptr(o), 4,
             ptr(ws), ws.numel() - 16, stream())
    assert int(_lib.load().clift_xyz_head_last2_x6_workspace_bytes(100)) == 25600


# ============================================================================ narrow output layer + row softmax as one stream (ABI 15)
@pytest.mark.parametrize("M,no,act", [(1, 22, 2), (31, 22, 2), (33, 3, 0), (4097, 22, 2), (4099, 32, 2), (62003, 1, 2), (249001, 22, 2), (249001, 22, 0)])
def test_out_layer_fwd_against_fp64_and_the_unfused_pair(M, no, act):
    """clift_out_layer_fwd: out = act(H W^T + b) for no <= 32 outputs over a 256-wide hidden activation, act = none / row softmax, against
    float64 (logits to 2e-6 of the sum's term scale, probabilities to 2e-6 absolute) and against clift_gemm + clift_rows_act_fwd; written with a
    column offset into a wider row whose other columns stay untouched; ragged row counts."""
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(90 + M + no)
    H = torch.relu(torch.randn(M, 256, generator=g)).to(DEV)
    W = (torch.randn(no, 256, generator=g) / 8).to(DEV); b = torch.randn(no, generator=g).to(DEV)
    ldo = no + 5
    out = torch.full((M, ldo), -7.0, device=DEV)
    engine.out_layer_fwd(M, H, W, b, out, ldo, 2, act)
    ref = H.double().cpu() @ W.double().cpu().t() + b.double().cpu()
    scale = (H.double().cpu().abs() @ W.double().cpu().abs().t() + b.double().cpu().abs()).clamp_min(1e-30)
    pair = torch.empty(M, no, device=DEV)
    with engine.exact_fp32():
        engine.gemm(M, no, 256, H, 256, W, 256, pair, no, bias=b)
    if act == 2:
        ref = torch.softmax(ref, -1)
        call("clift_rows_act_fwd", ptr(pair), no, M, no, 2, ptr(pair), no, stream())
    torch.cuda.synchronize()
    got = out[:, 2:2 + no].double().cpu()
    assert bool((out[:, :2] == -7.0).all()) and bool((out[:, 2 + no:] == -7.0).all())
    if act == 2:
        assert float((got - ref).abs().max()) <= 2e-6
        assert float((got.sum(-1) - 1).abs().max()) <= 1e-5
        assert float((got - pair.double().cpu()).abs().max()) <= 2e-6
    else:
        assert float(((got - ref).abs() / scale).max()) <= 2e-6
        assert float(((got - pair.double().cpu()).abs() / scale).max()) <= 2e-6


def test_out_layer_fwd_row_limit_padded_pitches_and_bad_arguments():
    from contrastive_lift_amd import _lib, engine
    from contrastive_lift_amd._lib import call, ptr, stream
    cap, M, no = 9000, 5003, 22
    g = torch.Generator().manual_seed(12)
    Hp = torch.full((cap, 260), float("nan"), device=DEV); Hp[:M, :256] = torch.relu(torch.randn(M, 256, generator=g)).to(DEV)
    Wp = torch.full((no, 264), float("nan"), device=DEV); Wp[:, :256] = (torch.randn(no, 256, generator=g) / 8).to(DEV)
    b = torch.randn(no, generator=g).to(DEV)
    out = torch.full((cap, 24), -7.0, device=DEV)
    lim = engine.rows_limit(out.device)
    lim[0:1].fill_(M)
    try:
        call("clift_out_layer_fwd", ptr(Hp), 260, ptr(Wp), 264, ptr(b), no, cap, ptr(out), 24, 2, stream())
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(out.device)
    ref = torch.softmax(Hp[:M, :256].double().cpu() @ Wp[:, :256].double().cpu().t() + b.double().cpu(), -1)
    assert bool((out[M:] == -7.0).all()) and bool((out[:, no:] == -7.0).all())
    assert float((out[:M, :no].double().cpu() - ref).abs().max()) <= 2e-6
    with pytest.raises(_lib.CliftError, match="out_features"):
        call("clift_out_layer_fwd", ptr(Hp), 260, ptr(Wp), 264, ptr(b), 33, 100, ptr(out), 40, 0, stream())
    with pytest.raises(_lib.CliftError, match="act must be"):
        call("clift_out_layer_fwd", ptr(Hp), 260, ptr(Wp), 264, ptr(b), 22, 100, ptr(out), 24, 1, stream())


# ============================================================================ sign bytes instead of the fp32 mask (ABI 15)
@pytest.mark.parametrize("M", [1, 31, 33, 4097, 62003, 249000])
def test_fp32x6_dgrad_from_sign_bytes_is_the_masked_dgrad_bit_for_bit(M):
    """A persistent fp32x6 forward (plain and with the K = 3 layer generated) that also writes the SIGNS of its output (clift_gemm_t.sign_bits,
    32 B per row) -- same activation bits as without -- and the next layer's masked input gradient reading them instead of the fp32 mask:
    bit-identical to the dgrad that streams the activation as its mask, at ragged row counts and under a row limit."""
    from contrastive_lift_amd import engine
    g = torch.Generator().manual_seed(50 + M)
    A = (torch.randn(M, 256, generator=g) * (torch.rand(M, 256, generator=g) > 0.3)).to(DEV)
    W = (torch.randn(256, 256, generator=g) / 16).to(DEV); b = (0.3 * torch.randn(256, generator=g)).to(DEV)
    W0 = torch.randn(256, 3, generator=g).to(DEV); b0 = (0.5 * torch.randn(256, generator=g)).to(DEV)
    x4 = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).contiguous().to(DEV)
    dY = torch.randn(M, 256, generator=g).to(DEV)
    prev = engine.set_mlp_precision("fp32x6")
    try:
        for gen in (False, True):
            h_ref, h = torch.empty(M, 256, device=DEV), torch.empty(M, 256, device=DEV)
            sb = engine.sign_bits_for(M, DEV)
            sb.fill_(0xA5)
            if gen:
                engine.first2_x6(M, x4, W0, b0, W, b, h_ref)
                engine.first2_x6(M, x4, W0, b0, W, b, h, sb)
            else:
                engine.gemm(M, 256, 256, A, 256, W, 256, h_ref, 256, bias=b, act=1)
                engine.gemm(M, 256, 256, A, 256, W, 256, h, 256, bias=b, act=1, sign_bits=sb)
            d_ref, d = torch.empty(M, 256, device=DEV), torch.empty(M, 256, device=DEV)
            engine.gemm(M, 256, 256, dY, 256, W, 256, d_ref, 256, b_trans=1, mask=h_ref, ldmask=256)
            engine.gemm(M, 256, 256, dY, 256, W, 256, d, 256, b_trans=1, sign_bits=sb)
            torch.cuda.synchronize()
            assert torch.equal(h, h_ref), (M, gen)
            assert torch.equal(d, d_ref), (M, gen)
            assert float((d_ref == 0).float().mean()) > 0.2          # the mask does mask
    finally:
        engine.set_mlp_precision(prev)


def test_fp32x6_sign_bytes_under_a_row_limit_and_bad_arguments():
    from contrastive_lift_amd import _lib, engine
    cap, M = 9000, 5003
    g = torch.Generator().manual_seed(8)
    A = torch.randn(cap, 256, generator=g).to(DEV); A[M:] = float("nan")
    W = (torch.randn(256, 256, generator=g) / 16).to(DEV); b = torch.zeros(256, device=DEV)
    dY = torch.randn(cap, 256, generator=g).to(DEV); dY[M:] = float("nan")
    prev = engine.set_mlp_precision("fp32x6")
    lim = engine.rows_limit(A.device)
    try:
        h = torch.full((cap, 256), -7.0, device=DEV); d = torch.full((cap, 256), -7.0, device=DEV); d_ref = torch.full((cap, 256), -7.0, device=DEV)
        sb = engine.sign_bits_for(cap, DEV)
        lim[0:1].fill_(M)
        engine.gemm(cap, 256, 256, A, 256, W, 256, h, 256, bias=b, act=1, sign_bits=sb)
        engine.gemm(cap, 256, 256, dY, 256, W, 256, d, 256, b_trans=1, sign_bits=sb)
        engine.gemm(cap, 256, 256, dY, 256, W, 256, d_ref, 256, b_trans=1, mask=h, ldmask=256)
        torch.cuda.synchronize()
        assert torch.equal(d, d_ref) and bool((d[M:] == -7.0).all()) and bool(torch.isfinite(d[:M]).all())
    finally:
        engine.reset_rows_limit(A.device)
        engine.set_mlp_precision(prev)
    with pytest.raises(_lib.CliftError, match="sign_bits"):          # exact arithmetic has no such form
        with engine.exact_fp32():
            engine.gemm(100, 256, 256, A, 256, W, 256, h, 256, bias=b, act=1, sign_bits=sb)
    assert int(_lib.load().clift_sign_bits_bytes(33)) == 2048


# ============================================================================ first two layers' backward, fp32x6
@pytest.mark.parametrize("M", [1, 31, 32, 33, 64, 4097, 62003, 249000])
def test_first2_x6_bwd_against_fp64_and_the_exact_kernel(M):
    """clift_xyz_head_first2_x6_bwd: gW0 / gb0 against the float64 sums (mask = sign of the activation the forward kernel produced) within
    the exact kernel's band (2e-5 of each tensor's largest entry: fp32 sums over M rows), next to clift_xyz_head_first2_bwd; accumulates."""
    from contrastive_lift_amd import engine
    t, h1, gW_ref, gb_ref = _first2_case(M, 2000 + M)
    gW, gb = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_x6_bwd(M, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)
    gWe, gbe = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_bwd(M, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gWe, gbe)
    torch.cuda.synchronize()
    for got, ex, ref, nm in ((gW, gWe, gW_ref, "gW0"), (gb, gbe, gb_ref, "gb0")):
        sc = max(float(ref.abs().max()), 1e-30)
        e6 = float((got.double().cpu() - ref).abs().max()) / sc
        e0 = float((ex.double().cpu() - ref).abs().max()) / sc
        assert e6 < 2e-5 and e6 <= 4 * e0 + 2e-6, (M, nm, e6, e0)
    engine.first2_x6_bwd(M, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)
    torch.cuda.synchronize()
    err = float((gW.double().cpu() - 2 * gW_ref).abs().max()) / max(float(gW_ref.abs().max()), 1e-30)
    assert err < 4e-5, (M, "accumulate", err)


@pytest.mark.parametrize("M", [1, 31, 33, 2049, 62003, 249000])
def test_first2_x6_wgrad_against_fp64_and_the_exact_kernel(M):
    """clift_xyz_head_first2_x6_wgrad: gW1 += dH2^T relu(x W0^T + b0), gb1 += column sums, against float64 over the activation the forward
    kernel produces, next to clift_xyz_head_first2_wgrad (band 2e-5 of the largest entry)."""
    from contrastive_lift_amd import engine
    t, h1, _, _ = _first2_case(M, 3000 + M)
    ref = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    refb = t["d"][:M].double().cpu().sum(0)
    gW, gb = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_x6_wgrad(M, t["d"], t["W0"], t["b0"], t["x4"], gW, gb)
    gWe, gbe = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    engine.first2_wgrad(M, t["d"], t["W0"], t["b0"], t["x4"], gWe, gbe)
    torch.cuda.synchronize()
    for got, ex, r, nm in ((gW, gWe, ref, "gW1"), (gb, gbe, refb, "gb1")):
        sc = max(float(r.abs().max()), 1e-30)
        e6 = float((got.double().cpu() - r).abs().max()) / sc
        e0 = float((ex.double().cpu() - r).abs().max()) / sc
        assert e6 < 2e-5 and e6 <= 4 * e0 + 2e-6, (M, nm, e6, e0)
    engine.first2_x6_wgrad(M, t["d"], t["W0"], t["b0"], t["x4"], gW, gb)
    torch.cuda.synchronize()
    assert float((gW.double().cpu() - 2 * ref).abs().max()) / max(float(ref.abs().max()), 1e-30) < 4e-5


def test_first2_x6_backward_kernels_row_limit_and_padded_pitches():
    """Both kernels under the device-side row limit of a sync-free step (rows past it are NaN) and with every pitch larger than its row."""
    from contrastive_lift_amd import engine
    from contrastive_lift_amd._lib import call, ptr, stream
    cap, M = 9000, 5003
    t, h1, gW_ref, gb_ref = _first2_case(M, 6, cap=cap)
    t["d"][M:] = float("nan"); t["x4"][M:] = float("nan")
    gW, gb = torch.zeros(256, 3, device=DEV), torch.zeros(256, device=DEV)
    gW1, gb1 = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
    lim = engine.rows_limit(gW.device)
    lim[0:1].fill_(M)
    try:
        engine.first2_x6_bwd(cap, t["d"], t["W1"], t["W0"], t["b0"], t["x4"], gW, gb)
        engine.first2_x6_wgrad(cap, t["d"], t["W0"], t["b0"], t["x4"], gW1, gb1)
        torch.cuda.synchronize()
    finally:
        engine.reset_rows_limit(gW.device)
    ref1 = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    for got, ref in ((gW, gW_ref), (gb, gb_ref), (gW1, ref1)):
        assert bool(torch.isfinite(got).all())
        assert float((got.double().cpu() - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    # padded pitches through the C ABI: dH2 260, W1 264, W0 / gW0 4, gW1 272
    M = 7001
    t, h1, gW0_ref, gb0_ref = _first2_case(M, 32)
    dp = torch.full((M, 260), float("nan"), device=DEV); dp[:, :256] = t["d"]
    W1p = torch.full((256, 264), float("nan"), device=DEV); W1p[:, :256] = t["W1"]
    W0p = torch.full((256, 4), float("nan"), device=DEV); W0p[:, :3] = t["W0"]
    gW0 = torch.full((256, 4), -7.0, device=DEV); gW0[:, :3] = 0
    gb0 = torch.zeros(256, device=DEV)
    call("clift_xyz_head_first2_x6_bwd", ptr(dp), 260, ptr(W1p), 264, ptr(W0p), 4, ptr(t["b0"]), ptr(t["x4"]), M, ptr(gW0), 4, ptr(gb0), stream())
    gW1 = torch.full((256, 272), -7.0, device=DEV); gW1[:, :256] = 0
    gb1 = torch.zeros(256, device=DEV)
    call("clift_xyz_head_first2_x6_wgrad", ptr(dp), 260, ptr(W0p), 4, ptr(t["b0"]), ptr(t["x4"]), M, ptr(gW1), 272, ptr(gb1), stream())
    torch.cuda.synchronize()
    assert bool((gW0[:, 3] == -7.0).all()) and bool((gW1[:, 256:] == -7.0).all())
    assert float((gW0[:, :3].double().cpu() - gW0_ref).abs().max()) / float(gW0_ref.abs().max()) < 2e-5
    ref1 = t["d"][:M].double().cpu().t() @ h1[:M].double().cpu()
    assert float((gW1[:, :256].double().cpu() - ref1).abs().max()) / float(ref1.abs().max()) < 2e-5


def test_fp32x6_head_backward_matches_the_exact_path_end_to_end():
    """A four-layer (instance) and a five-layer (semantic, C = 2) head, forward + backward through engine.xyz_mlp_fwd / xyz_mlp_bwd in fp32x6
    mode (every fused end on the split kernels) against the same calls in exact fp32: outputs to 1e-5 of the output scale, every gradient
    within conftest.grad_close's band (2e-3 relative + 1e-3 of the scale, 0.1 % of the entries up to 1e-2 of the scale: one sample whose
    hidden unit lands on the other side of a ReLU kink between the two arithmetics moves whole rows of the downstream gradients by its
    contribution; the kernels themselves are held to 2e-5 against fp64 above)."""
    from contrastive_lift_amd import engine
    M = 30011
    g = torch.Generator().manual_seed(4)
    xa = torch.cat([torch.rand(M, 3, generator=g) * 2 - 1, torch.zeros(M, 1)], 1).to(DEV).contiguous()
    for nl, E in ((4, 3), (5, 2)):
        dims = [3] + [256] * (nl - 1) + [E]
        layers = [((torch.randn(dims[i + 1], dims[i], generator=g) / (dims[i] ** 0.5)).to(DEV), (0.2 * torch.randn(dims[i + 1], generator=g)).to(DEV))
                  for i in range(nl)]
        ldp = 4
        dpre = torch.zeros(M, ldp, device=DEV); dpre[:, :E] = torch.randn(M, E, generator=g).to(DEV)
        res = {}
        for mode in ("fp32", "fp32x6"):
            prev = engine.set_mlp_precision(mode)
            try:
                out = torch.zeros(M, E, device=DEV)
                acts = engine.xyz_mlp_fwd(layers, xa, M, out, E, keep_first=True)
                gl = [(torch.zeros_like(W), torch.zeros_like(b)) for W, b in layers]
                engine.xyz_mlp_bwd(layers, gl, xa, acts, dpre.clone(), M)
                torch.cuda.synchronize()
            finally:
                engine.set_mlp_precision(prev)
            res[mode] = (out, gl)
        o0, o6 = res["fp32"][0], res["fp32x6"][0]
        assert float((o0 - o6).abs().max()) <= 1e-5 * float(o0.abs().max())
        for (gW0, gb0), (gW6, gb6) in zip(res["fp32"][1], res["fp32x6"][1]):
            for a, b in ((gW0, gW6), (gb0, gb6)):
                grad_close(b, a, what=f"{nl}-layer head, {tuple(a.shape)}")


# ============================================================================ trainer: a pass that dies leaves nothing behind
def test_aborted_pass_leaves_no_stale_gradient_shards(monkeypatch):
    """A main pass that raises after its weight-gradient kernels have flushed into the XCD shards: the next pass's gradients must be those of a
    trainer that never saw the failure (the shards are switched off AND cleared on the way out, the CU reserve is reset)."""
    from contrastive_lift_amd import engine
    from test_gpu_trainer_modes import _setup
    ta, batch, jit = _setup(seed=5)
    tb, _, _ = _setup(seed=5)
    if ta._shards is None:
        pytest.skip("gradient shards are off")
    real = engine._density_backward

    def boom(*a, **k):
        raise RuntimeError("injected failure after the head chains")
    monkeypatch.setattr(engine, "_density_backward", boom)
    with pytest.raises(RuntimeError, match="injected"):
        ta.main_pass(batch[0], jitter=jit, white_bg=False)
    monkeypatch.setattr(engine, "_density_backward", real)
    torch.cuda.synchronize()
    assert float(ta._shards.abs().max()) == 0.0
    assert int(engine.grad_shard_record(ta.model.param_flat.device)[4]) == 0
    # the failed pass never reached Adam, so both trainers hold the same parameters: the next pass must give the same gradients
    ta.opt_main.step = lambda *a, **k: None
    tb.opt_main.step = lambda *a, **k: None
    ta.main_pass(batch[0], jitter=jit, white_bg=False)
    tb.main_pass(batch[0], jitter=jit, white_bg=False)
    torch.cuda.synchronize()
    a0, a1 = ta.model.arena.range_of("net_app", "net_sem")
    ga, gb = ta.model.grad_flat[a0:a1], tb.model.grad_flat[a0:a1]
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
