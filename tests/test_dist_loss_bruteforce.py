"""a8 (distortion loss): the package the reference calls (torch_efficient_distloss==0.1.3, renderer.py:30,101) is neither vendored nor
installed, so ``oracle.render.dist_loss`` cannot be pinned against it ("parity unpinned").  What CAN be verified is that the O(S)
prefix-sum form the oracle (and k_march_fwd/bwd) restates equals the published DEFINITION of the loss (Mip-NeRF 360 eq. 15 /
DVGOv2):   L(ray) = sum_i sum_j w_i w_j |m_i - m_j|  +  (1/3) sum_i w_i^2 delta_i ,   mean over rays,
evaluated here by brute force over all S^2 pairs in fp64.  The identity needs sorted midpoints; the reference's LAST midpoint
(renderer.py:84: cat(..., z[:, -2:-1])) is smaller than the one before it, so for that element the prefix-sum form carries the
signed term w_i w_j (m_i - m_j) instead of the absolute value -- both facts are asserted."""
import numpy as np
import torch

from oracle import render as orender


def brute_force(w, m, delta, signed=False):
    """O(S^2) definition in fp64.  signed=False: |m_i - m_j| (the published definition); signed=True: 2 sum_{i>j} w_i w_j (m_i - m_j),
    which is what ANY prefix-sum evaluation computes when the midpoints are not sorted."""
    w, m, delta = w.double(), m.double(), delta.double()
    dm = m[:, :, None] - m[:, None, :]                       # (N, S, S): m_i - m_j
    ww = w[:, :, None] * w[:, None, :]
    if signed:
        lower = torch.tril(torch.ones(w.shape[1], w.shape[1], dtype=torch.bool), diagonal=-1)          # i > j
        bi = 2.0 * (ww * dm * lower).sum((1, 2))
    else:
        bi = (ww * dm.abs()).sum((1, 2))
    uni = (w * w * delta).sum(1) / 3.0
    return (bi + uni).mean()


def _rays(seed, N=37, S=96, jitter=True):
    rng = np.random.default_rng(seed)
    t0 = torch.from_numpy(rng.uniform(0.1, 0.6, (N, 1)))
    step = 0.013
    k = torch.arange(S, dtype=torch.float64)[None] + (torch.from_numpy(rng.uniform(0, 1, (N, 1))) if jitter else 0.0)
    z = (t0 + step * k).float()
    sigma = torch.from_numpy(rng.gamma(0.6, 4.0, (N, S))).float() * (torch.from_numpy(rng.uniform(0, 1, (N, S))) > 0.5)
    dists, mid = orender._deltas_midpoints(z)
    alpha, w, bg = orender.sigma_to_weights(sigma, dists * 25.0)
    return z, dists, mid, w


def test_prefix_sum_form_equals_the_pairwise_definition_for_sorted_midpoints():
    for seed in range(4):
        z, dists, mid, w = _rays(seed)
        mid_sorted = mid.clone()
        mid_sorted[:, -1] = z[:, -1]                         # a sorted variant of the reference's midpoints
        assert bool((mid_sorted[:, 1:] >= mid_sorted[:, :-1]).all())
        got = orender.dist_loss(w.double(), mid_sorted.double(), dists.double())
        want = brute_force(w, mid_sorted, dists)
        assert abs(float(got) - float(want)) <= 1e-12 * max(1.0, abs(float(want))), (float(got), float(want))
        got32 = orender.dist_loss(w, mid_sorted, dists)
        assert abs(float(got32) - float(want)) <= 2e-5 * abs(float(want)) + 1e-9


def test_reference_midpoints_last_element_is_signed_not_absolute():
    """With the reference's own midpoints (last one out of order) the prefix-sum form equals the SIGNED pairwise sum exactly, and
    differs from the |.| definition only through the pairs that involve the last sample (whose weight is 0: its delta is 0,
    renderer.py:83 => alpha = 0), i.e. not at all in practice."""
    for seed in range(4):
        z, dists, mid, w = _rays(10 + seed)
        assert bool((mid[:, -1] < mid[:, -2]).all())
        got = orender.dist_loss(w.double(), mid.double(), dists.double())
        signed = brute_force(w, mid, dists, signed=True)
        absolute = brute_force(w, mid, dists, signed=False)
        assert abs(float(got) - float(signed)) <= 1e-12 * max(1.0, abs(float(signed)))
        assert float(w[:, -1].abs().max()) == 0.0            # delta_last = 0  =>  alpha_last = 0  =>  w_last = 0
        assert abs(float(signed) - float(absolute)) <= 1e-15


def test_gradient_of_prefix_sum_form_equals_gradient_of_the_definition():
    z, dists, mid, w = _rays(99, N=9, S=64)
    w1 = w.double().clone().requires_grad_(True)
    g1 = torch.autograd.grad(orender.dist_loss(w1, mid.double(), dists.double()), w1)[0]
    w2 = w.double().clone().requires_grad_(True)
    g2 = torch.autograd.grad(brute_force(w2, mid, dists), w2)[0]
    assert float((g1 - g2).abs().max()) <= 1e-12 * max(1.0, float(g2.abs().max()))
