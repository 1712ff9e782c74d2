#!/usr/bin/env python3
"""Prediction folders for evaluator tests: what inference/render_panopli.py would write for a nearly converged model."""
import os

import numpy as np


def write_fake_predictions(out_dir, names, sems, insts, rng, flip=0.08):
    """pred_semantics/<name>.png (uint8) and pred_surrogateid/<name>.png (uint16) = ground truth with a fraction of the pixels
    re-labelled and the instance ids pushed through a fixed permutation (what a converged model + clustering would write)."""
    from PIL import Image
    os.makedirs(os.path.join(out_dir, "pred_semantics"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "pred_surrogateid"), exist_ok=True)
    n_sem = int(max(s.max() for s in sems)) + 1
    n_inst = int(max(i.max() for i in insts)) + 1
    perm = np.concatenate([[0], 1 + rng.permutation(n_inst)])
    for name, sem, inst in zip(names, sems, insts):
        noise = rng.uniform(0, 1, sem.shape) < flip
        ps = np.where(noise, rng.integers(0, n_sem, sem.shape), sem)
        pi = np.where(rng.uniform(0, 1, sem.shape) < flip, rng.integers(0, n_inst + 1, sem.shape), perm[inst])
        Image.fromarray(ps.astype(np.uint8)).save(os.path.join(out_dir, "pred_semantics", f"{name}.png"))
        Image.fromarray(pi.astype(np.uint16)).save(os.path.join(out_dir, "pred_surrogateid", f"{name}.png"))


def fake_thing_features(seed=171, n_img=2, per=36000):
    """(n_img * per, 1 + 3) rendered "thing feature" rows as render_panopli.py collects them: column 0 is -inf for thing pixels
    and +inf for stuff pixels, columns 1..3 are instance embeddings drawn around four well separated centres."""
    rng = np.random.default_rng(seed)
    centers = np.array([[0.0, 0.0, 0.0], [1.0, 0.2, -0.3], [-0.4, 0.9, 0.5], [0.5, -0.8, 0.7]])
    feats = np.concatenate([c + 0.06 * rng.standard_normal((n_img * per // len(centers), 3)) for c in centers]).astype(np.float32)
    feats = feats[rng.permutation(feats.shape[0])]
    thing = rng.uniform(0, 1, feats.shape[0]) < 0.85                   # 61 k thing pixels >= the reference's 50 k subsample
    first = np.where(thing, -np.inf, np.inf).astype(np.float32)
    return np.concatenate([first[:, None], feats], 1), n_img


def fake_semantics_for(all_thing, n_img, seed=181, n_classes=5):
    """Per-image (P, n_classes) semantic score tensors consistent with ``all_thing``: stuff rows score highest on class 0 or 1,
    thing rows on class 2, 3 or 4 -- class 2 by the sign of the first embedding coordinate, class 4 for ~60 rows only (so that
    one class falls under the 100-point MeanShift minimum)."""
    import torch
    rng = np.random.default_rng(seed)
    thing = all_thing[:, 0] == -np.inf
    cls = np.where(thing, np.where(all_thing[:, 1] > 0.3, 2, 3), rng.integers(0, 2, all_thing.shape[0]))
    few = np.flatnonzero(thing)[rng.permutation(int(thing.sum()))[:60]]
    cls[few] = 4
    scores = rng.standard_normal((all_thing.shape[0], n_classes)).astype(np.float32)
    scores[np.arange(all_thing.shape[0]), cls] += 6.0
    per = all_thing.shape[0] // n_img
    return [torch.from_numpy(scores[i * per:(i + 1) * per].copy()) for i in range(n_img)]
