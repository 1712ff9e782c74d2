"""Synthetic stand-in scenes with the shapes of BASELINE.json's configs (no datasets are available offline).

Recipe (scaled): reference initialisation of the field (torch seed), density component 0 of every
plane/line overwritten by a separable Gaussian bump (sigma_g 0.35, amplitude 3) with splus_density_shift = -3 => an
opaque blob at the origin (in-box fraction ~0.58, active fraction ~0.16 of the N*S nominal samples); cameras on the
radius-0.9 sphere looking at the origin so every ray intersects the unit sphere (util/ray.py:81-99).
"""
import numpy as np
import torch

from .field import TensorVMSplit, MATRIX_MODE, VECTOR_MODE
from .rays import generate_ray_table
from .renderer import TensoRFRenderer


def look_at(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    M = np.eye(4)
    M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = r, d, f, eye
    return M.astype(np.float32)


@torch.no_grad()
def add_blob(model, amplitude=3.0, sigma_g=0.35):
    dev = model.param_flat.device
    for i in range(3):
        a, b = MATRIX_MODE[i]
        v = VECTOR_MODE[i]
        pl, ln = model.density_plane[i], model.density_line[i]
        ga = torch.exp(-(torch.linspace(-1, 1, pl.shape[3]) / sigma_g) ** 2).to(dev)
        gb = torch.exp(-(torch.linspace(-1, 1, pl.shape[2]) / sigma_g) ** 2).to(dev)
        gv = torch.exp(-(torch.linspace(-1, 1, ln.shape[2]) / sigma_g) ** 2).to(dev)
        pl.zero_()
        ln.zero_()
        pl[0, 0] = gb[:, None] * ga[None, :] * amplitude
        ln[0, 0, :, 0] = gv * amplitude
    return model


def make_scene(grid=128, num_classes=22, max_instances=3, seed=0, device="cuda", image=512, n_cams=3, step_ratio=0.5,
               aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))):
    """Returns (model, renderer, ray_pool (P,8) on device)."""
    torch.manual_seed(seed)
    model = TensorVMSplit([grid] * 3, num_semantics_comps=(32, 32, 32), num_instance_comps=(32, 32, 32),
                          num_semantic_classes=num_classes, dim_feature_instance=2 * max_instances, splus_density_shift=-3.0,
                          use_semantic_mlp=True, use_instance_mlp=True, slow_fast_mode=True, device=device)
    add_blob(model)
    renderer = TensoRFRenderer(torch.tensor(aabb), [grid] * 3, semantic_weight_mode="softmax", step_ratio=step_ratio).to(device)
    K = np.array([[1.25 * image, 0, image / 2], [0, 1.25 * image, image / 2], [0, 0, 1]], np.float32)
    eyes = [(0.0, 0.0, -0.9), (0.7, -0.35, 0.45), (-0.55, 0.5, 0.5), (0.3, 0.8, -0.25)][:n_cams]
    pool = torch.cat([generate_ray_table(image, image, K, look_at(e), device=device) for e in eyes], 0)
    return model, renderer, pool


def make_batches(pool, n_rays, n_inst_rays, num_classes, n_labels, seed, device):
    """One training_step batch in the reference's CombinedLoader layout (SURVEY 8b 'Batch dict')."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randint(0, pool.shape[0], (n_rays,), generator=g).to(device)
    b0 = dict(rays=pool[idx].contiguous(),
              rgbs=torch.rand((n_rays, 3), generator=g).to(device),
              probabilities=torch.softmax(torch.randn((n_rays, num_classes), generator=g), -1).to(device),
              confidences=torch.rand((n_rays,), generator=g).to(device),
              mask=torch.ones((n_rays,), dtype=torch.bool).to(device))
    # one instance image: rays of a single camera, labels from a Zipf-like histogram (some ids in one half only)
    cam_rays = pool.shape[0] // max(1, (pool.shape[0] // (512 * 512)))
    idx2 = torch.randint(0, min(cam_rays, pool.shape[0]), (n_inst_rays,), generator=g).to(device)
    w = 1.0 / torch.arange(1, n_labels + 1, dtype=torch.float32) ** 1.2
    labels = torch.multinomial(w / w.sum(), n_inst_rays, replacement=True, generator=g) + 1
    b1 = [dict(rays=pool[idx2].contiguous(), instances=labels.to(device), confidences=torch.rand((n_inst_rays,), generator=g).to(device))]
    return {0: b0, 1: b1}
