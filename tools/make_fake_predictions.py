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
