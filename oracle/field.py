"""Oracle: TensoRF VM-decomposition lookups and MLP heads.

Test infrastructure only (see oracle/__init__.py).  Functional style over a ``P`` dict whose keys are
the reference state_dict names (oracle/params.py).  Two VM-lookup implementations are kept:
``*_fast`` bottoms out in the same ATen ops as the reference (used for the CPU baseline so it is the
same workload), ``*_explicit`` is index arithmetic written from the grid_sample definition
(align_corners=True, zero padding) and is what pins the axis/tap conventions in tests.
"""
import torch
import torch.nn.functional as F

from .params import MATRIX_MODE, VECTOR_MODE


# ----------------------------------------------------------------------------- explicit taps
def _bilinear_cl(plane, x, y):
    """plane (C,H,W); x->W, y->H in [-1,1]; align_corners=True; out-of-range taps contribute zero."""
    C, H, W = plane.shape
    fx = (x + 1) / 2 * (W - 1)
    fy = (y + 1) / 2 * (H - 1)
    x0 = torch.floor(fx)
    y0 = torch.floor(fy)
    wx1 = fx - x0
    wy1 = fy - y0
    out = torch.zeros((C, x.shape[0]), dtype=plane.dtype)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            v = plane[:, yi.clamp(0, H - 1), xi.clamp(0, W - 1)]
            out = out + v * (wx * wy * ok)[None]
    return out


def _linear_cl(line, t):
    """line (C,R); t in [-1,1]; align_corners=True; zero padding."""
    C, R = line.shape
    f = (t + 1) / 2 * (R - 1)
    i0 = torch.floor(f)
    w1 = f - i0
    out = torch.zeros((C, t.shape[0]), dtype=line.dtype)
    for di, w in ((0, 1 - w1), (1, w1)):
        ii = (i0 + di).long()
        ok = (ii >= 0) & (ii < R)
        out = out + line[:, ii.clamp(0, R - 1)] * (w * ok)[None]
    return out


def vm_products_explicit(P, prefix, xn):
    """(sum_i C_i, M): per component plane_i(x[a_i],x[b_i]) * line_i(x[v_i]), plane-major concat.
    Reference tensoRF.py:108-112,117-120,130-134."""
    out = []
    for i in range(3):
        a, b = MATRIX_MODE[i]
        v = VECTOR_MODE[i]
        pl = _bilinear_cl(P[f"{prefix}_plane.{i}"][0], xn[:, a], xn[:, b])
        ln = _linear_cl(P[f"{prefix}_line.{i}"][0, :, :, 0], xn[:, v])
        out.append(pl * ln)
    return torch.cat(out, 0)


# ----------------------------------------------------------------------------- ATen-op path
def vm_products_fast(P, prefix, xn):
    M = xn.shape[0]
    out = []
    for i in range(3):
        a, b = MATRIX_MODE[i]
        v = VECTOR_MODE[i]
        gp = torch.stack((xn[:, a], xn[:, b]), -1).detach().view(1, M, 1, 2)
        gl = torch.stack((torch.zeros_like(xn[:, v]), xn[:, v]), -1).detach().view(1, M, 1, 2)
        pl = F.grid_sample(P[f"{prefix}_plane.{i}"], gp, mode="bilinear", padding_mode="zeros", align_corners=True)
        ln = F.grid_sample(P[f"{prefix}_line.{i}"], gl, mode="bilinear", padding_mode="zeros", align_corners=True)
        out.append(pl.view(-1, M) * ln.view(-1, M))
    return torch.cat(out, 0)


def density_raw(P, xn, shift=-10.0, explicit=False):
    """tensoRF.py:114-122: sum over all density components + softplus shift."""
    prod = (vm_products_explicit if explicit else vm_products_fast)(P, "density", xn)
    return prod.sum(0) + shift


def density(P, xn, shift=-10.0, explicit=False):
    """tensoRF.py:124-125."""
    return F.softplus(density_raw(P, xn, shift, explicit))


def appearance_feature(P, xn, explicit=False):
    """tensoRF.py:127-137: basis Linear(144->27, no bias) of the plane-major product vector."""
    prod = (vm_products_explicit if explicit else vm_products_fast)(P, "appearance", xn)
    return prod.T @ P["appearance_basis_mat.weight"].T


def posenc(x, freqs):
    """tensoRF.py:413-418: [sin(x_0 f_0), sin(x_0 f_1), sin(x_1 f_0), ... , then all cos]."""
    fb = 2.0 ** torch.arange(freqs, dtype=x.dtype)
    pts = (x[..., None] * fb).reshape(x.shape[:-1] + (freqs * x.shape[-1],))
    return torch.cat([torch.sin(pts), torch.cos(pts)], -1)


def _mlp(P, prefix, x, n_layers):
    h = x
    for li in range(n_layers):
        h = F.linear(h, P[f"{prefix}.{2 * li}.weight"], P[f"{prefix}.{2 * li}.bias"])
        if li < n_layers - 1:
            h = torch.relu(h)
    return h


def _count_layers(P, prefix):
    n = 0
    while f"{prefix}.{2 * n}.weight" in P:
        n += 1
    return n


def appearance_mlp(P, viewdirs, feat, pe_view=2, pe_feat=2):
    """tensoRF.py:400-411: input [feat, dirs, PE(feat), PE(dirs)] -> 128 -> 128 -> 3 -> sigmoid."""
    x = torch.cat([feat, viewdirs, posenc(feat, pe_feat), posenc(viewdirs, pe_view)], -1)
    return torch.sigmoid(_mlp(P, "render_appearance_mlp.mlp", x, 3))


def semantic_mlp(P, xn, softmax=True):
    """tensoRF.py:584-594 on normalised xyz (use_semantic_mlp => feature is xyz, tensoRF.py:142-144)."""
    out = _mlp(P, "render_semantic_mlp.mlp", xn, _count_layers(P, "render_semantic_mlp.mlp"))
    return torch.softmax(out, -1) if softmax else out


def instance_mlp(P, xn):
    """tensoRF.py:497-511: cat[fast(xyz), slow(xyz)], identity output activation."""
    n = _count_layers(P, "render_instance_mlp.mlp")
    fast = _mlp(P, "render_instance_mlp.mlp", xn, n)
    if "render_instance_mlp.slow_mlp.0.weight" in P:
        return torch.cat([fast, _mlp(P, "render_instance_mlp.slow_mlp", xn, n)], -1)
    return fast


def grid_feature(P, prefix, xn, explicit=False):
    """tensoRF.py:127-134 (compute_feature) for the semantic / instance grids: basis Linear (no bias) of the plane-major product vector."""
    prod = (vm_products_explicit if explicit else vm_products_fast)(P, prefix, xn)
    return prod.T @ P[f"{prefix}_basis_mat.weight"].T


def semantic_head(P, xn, softmax=True, explicit=False):
    """compute_semantic_feature + render_semantic_mlp (tensoRF.py:142-145, 584-594): the MLP's input is the normalised position when
    use_semantic_mlp, else the 27 features of the semantic VM grid (tensoRF.py:78-83)."""
    if "semantic_plane.0" not in P:
        return semantic_mlp(P, xn, softmax)
    out = _mlp(P, "render_semantic_mlp.mlp", grid_feature(P, "semantic", xn, explicit), _count_layers(P, "render_semantic_mlp.mlp"))
    return torch.softmax(out, -1) if softmax else out


def instance_head(P, xn, explicit=False):
    """compute_instance_feature + render_instance_mlp (tensoRF.py:152-156, 497-511), likewise; fast and slow nets read the same features."""
    if "instance_plane.0" not in P:
        return instance_mlp(P, xn)
    return instance_mlp(P, grid_feature(P, "instance", xn, explicit))


def instance_width(P):
    n = _count_layers(P, "render_instance_mlp.mlp")
    e = P[f"render_instance_mlp.mlp.{2 * (n - 1)}.weight"].shape[0]
    return 2 * e if "render_instance_mlp.slow_mlp.0.weight" in P else e
