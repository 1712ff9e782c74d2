"""Deterministic parameter sets for the oracle, the golden-vector generator and the GPU tests.

Keys and shapes follow the reference ``state_dict`` layout of ``TensorVMSplit`` in the
MLP-heads configuration (``use_semantic_mlp=True, use_instance_mlp=True, slow_fast_mode=True``;
reference model/radiance_field/tensoRF.py:63-85,99-106,393-398,475-491,576-582).

numpy's ``default_rng`` is bit-reproducible across platforms, so the generator script (which runs
where the reference is importable) and the tests (which run anywhere) build identical tensors
without the fixture having to carry megabytes of weights.
"""
import numpy as np
import torch

MATRIX_MODE = ((0, 1), (0, 2), (1, 2))   # tensoRF.py:61  plane i spans xyz axes (a,b): a->W, b->H
VECTOR_MODE = (2, 1, 0)                  # tensoRF.py:62  line i runs along axis v


def _linear(rng, out_f, in_f, zero_bias=False):
    bound = 1.0 / np.sqrt(in_f)
    w = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
    b = np.zeros(out_f, np.float32) if zero_bias else rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)


def make_params(seed, res, num_classes, num_inst, n_dens=(16, 16, 16), n_app=(48, 48, 48), dim_app=27,
                pe_view=2, pe_feat=2, dim_mlp_color=128, dim_mlp_sem=256, n_sem_layers=5,
                dim_mlp_inst=256, n_inst_layers=4, grid_scale=0.1, slow_fast=True, sem_grid=False, inst_grid=False,
                n_sem=(32, 32, 32), n_inst=(32, 32, 32), dim_sem=27, dim_inst=27, dim_mlp_sem_grid=128):
    """res = (Rx, Ry, Rz).  Returns an ordered dict name -> float32 tensor (reference shapes).
    ``sem_grid`` / ``inst_grid``: the head on its own VM grid (use_semantic_mlp / use_instance_mlp False, tensoRF.py:70-83): 3 x n components ->
    basis Linear (no bias) -> 27 features -> 3-layer MLP (128 wide for semantics, dim_mlp_inst for instances)."""
    rng = np.random.default_rng(seed)
    P = {}

    def grids(prefix, comps):
        for i in range(3):
            a, b = MATRIX_MODE[i]
            v = VECTOR_MODE[i]
            P[f"{prefix}_plane.{i}"] = torch.from_numpy(
                (grid_scale * rng.standard_normal((1, comps[i], res[b], res[a]))).astype(np.float32))
            P[f"{prefix}_line.{i}"] = torch.from_numpy(
                (grid_scale * rng.standard_normal((1, comps[i], res[v], 1))).astype(np.float32))

    grids("density", n_dens)
    grids("appearance", n_app)
    w, _ = _linear(rng, dim_app, sum(n_app))
    P["appearance_basis_mat.weight"] = w
    in_app = 2 * pe_view * 3 + 2 * pe_feat * dim_app + dim_app + 3
    dims = [in_app, dim_mlp_color, dim_mlp_color, 3]
    for li in range(3):
        w, b = _linear(rng, dims[li + 1], dims[li], zero_bias=(li == 2))
        P[f"render_appearance_mlp.mlp.{2 * li}.weight"] = w
        P[f"render_appearance_mlp.mlp.{2 * li}.bias"] = b
    if inst_grid:               # (drawn in the order of the reference's __init__: instance head before the semantic one)
        grids("instance", n_inst)
        P["instance_basis_mat.weight"] = _linear(rng, dim_inst, sum(n_inst))[0]
    if sem_grid:
        grids("semantic", n_sem)
        P["semantic_basis_mat.weight"] = _linear(rng, dim_sem, sum(n_sem))[0]
        dim_mlp_sem, n_sem_layers = dim_mlp_sem_grid, 3
    dims = [dim_sem if sem_grid else 3] + [dim_mlp_sem] * (n_sem_layers - 1) + [num_classes]
    for li in range(n_sem_layers):
        w, b = _linear(rng, dims[li + 1], dims[li])
        P[f"render_semantic_mlp.mlp.{2 * li}.weight"] = w
        P[f"render_semantic_mlp.mlp.{2 * li}.bias"] = b
    if inst_grid:
        n_inst_layers = 3
    dims = [dim_inst if inst_grid else 3] + [dim_mlp_inst] * (n_inst_layers - 1) + [num_inst]
    for net in (("mlp", "slow_mlp") if slow_fast else ("mlp",)):
        for li in range(n_inst_layers):
            w, b = _linear(rng, dims[li + 1], dims[li])
            P[f"render_instance_mlp.{net}.{2 * li}.weight"] = w
            P[f"render_instance_mlp.{net}.{2 * li}.bias"] = b
    return P


def add_blob(P, res, amplitude=3.0, sigma_g=0.35):
    """overwrite density component 0 of every plane/line by a separable Gaussian
    bump so that (with splus_density_shift=-3) the field has an opaque blob at the origin."""
    for i in range(3):
        a, b = MATRIX_MODE[i]
        v = VECTOR_MODE[i]
        ga = torch.exp(-(torch.linspace(-1, 1, res[a]) / sigma_g) ** 2)
        gb = torch.exp(-(torch.linspace(-1, 1, res[b]) / sigma_g) ** 2)
        gv = torch.exp(-(torch.linspace(-1, 1, res[v]) / sigma_g) ** 2)
        P[f"density_plane.{i}"].zero_()
        P[f"density_line.{i}"].zero_()
        P[f"density_plane.{i}"][0, 0] = gb[:, None] * ga[None, :] * amplitude
        P[f"density_line.{i}"][0, 0, :, 0] = gv * amplitude
    return P


def clone_params(P, requires_grad=False):
    return {k: v.detach().clone().requires_grad_(requires_grad) for k, v in P.items()}
