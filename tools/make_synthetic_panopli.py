#!/usr/bin/env python3
"""Write a small synthetic scene in the PanopLi on-disk layout (color/*.jpg, intrinsic/intrinsic_color.txt, pose/*.txt,
m2f_semantics/*.png, m2f_instance/*.png, m2f_segments/*.png, m2f_probabilities/*.npz, rs_semantics / rs_instance, splits.json,
segmentation_data.pkl, optional invalid/*.jpg): the analytic sphere scene of make_synthetic_mos.py with one semantic class
per sphere group.  Used by the reader tests; no dataset is available offline.
Usage: tools/make_synthetic_panopli.py <out_dir> [n_frames] [size]"""
import json
import os
import pickle
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_synthetic_mos import look_at_cv  # noqa: E402


def make_scene(out, n_frames=12, size=32, seed=0, invalid_frames=(), n_classes=5):
    rng = np.random.default_rng(seed)
    for d in ("color", "intrinsic", "pose", "m2f_semantics", "m2f_instance", "m2f_segments", "m2f_probabilities", "rs_semantics", "rs_instance"):
        os.makedirs(os.path.join(out, d), exist_ok=True)
    centers = np.array([[0.0, 0.0, 0.35], [0.75, 0.15, 0.3], [-0.6, 0.55, 0.28], [-0.2, -0.75, 0.3], [0.5, -0.6, 0.25]])
    radii = np.array([0.35, 0.3, 0.28, 0.3, 0.25])
    colors = np.array([[0.9, 0.2, 0.2], [0.2, 0.8, 0.3], [0.2, 0.3, 0.9], [0.9, 0.8, 0.2], [0.8, 0.3, 0.8]])
    sphere_class = np.array([2, 2, 3, 3, 4])              # things: classes 2..4; stuff: 0 (void/background), 1 (floor)
    fx = 1.1 * size
    K = np.array([[fx, 0, size / 2, 0], [0, fx, size / 2, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])
    np.savetxt(os.path.join(out, "intrinsic", "intrinsic_color.txt"), K)
    light = np.array([0.4, -0.3, 0.85]); light /= np.linalg.norm(light)
    jj, ii = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    dirs_cam = np.stack([(ii - K[0, 2]) / K[0, 0], (jj - K[1, 2]) / K[1, 1], np.ones_like(ii, float)], -1).reshape(-1, 3)
    names = []
    for f in range(n_frames):
        az = 2 * np.pi * rng.uniform()
        el = rng.uniform(0.35, 1.0)
        eye = 3.2 * np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
        Rcv = look_at_cv(eye, target=(0, 0, 0.25))
        d = dirs_cam @ Rcv.T
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tbest = np.full(d.shape[0], np.inf)
        inst = np.zeros(d.shape[0], np.int64)
        sem = np.zeros(d.shape[0], np.int64)
        rgb = np.tile(np.array([0.08, 0.08, 0.1]), (d.shape[0], 1))
        tg = -eye[2] / d[:, 2]
        pg = eye + tg[:, None] * d
        hit = (tg > 0) & (np.linalg.norm(pg[:, :2], axis=1) < 1.6)
        checker = ((np.floor(pg[:, 0] * 2) + np.floor(pg[:, 1] * 2)) % 2)
        rgb[hit] = (0.35 + 0.2 * checker[hit])[:, None] * np.array([1.0, 0.95, 0.9])
        tbest[hit] = tg[hit]
        sem[hit] = 1
        for s, (c, r, col) in enumerate(zip(centers, radii, colors)):
            oc = eye - c
            b = d @ oc
            disc = b * b - (oc @ oc - r * r)
            ok = disc > 0
            t = -b - np.sqrt(np.where(ok, disc, 0))
            ok &= (t > 0) & (t < tbest)
            n = (eye + t[:, None] * d - c) / r
            shade = 0.25 + 0.75 * np.clip(n @ light, 0, 1)
            rgb[ok] = shade[ok, None] * col
            tbest[ok] = t[ok]
            inst[ok] = s + 1
            sem[ok] = sphere_class[s]
        name = str(f * 10)                                  # numeric names, sorted numerically by the reader
        names.append(name)
        Image.fromarray((rgb.reshape(size, size, 3).clip(0, 1) * 255).astype(np.uint8)).save(os.path.join(out, "color", name + ".jpg"), quality=95)
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = Rcv, eye                # camera-to-world, OpenCV axes
        np.savetxt(os.path.join(out, "pose", name + ".txt"), pose)
        perm = np.concatenate([[0], 1 + rng.permutation(len(centers))])      # per-view inconsistent machine ids
        segs = np.where(sem == 1, len(centers) + 1, perm[inst])              # 2D segments: every sphere + the floor region
        for dname, arr in (("m2f_semantics", sem), ("rs_semantics", sem), ("rs_instance", inst), ("m2f_instance", perm[inst]), ("m2f_segments", segs)):
            Image.fromarray(arr.reshape(size, size).astype(np.uint8)).save(os.path.join(out, dname, name + ".png"))
        logits = rng.standard_normal((size, size, n_classes)) + 4.0 * np.eye(n_classes)[sem.reshape(size, size)]
        prob = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
        np.savez(os.path.join(out, "m2f_probabilities", name + ".npz"), probability=prob.astype(np.float32),
                 confidence=rng.uniform(0.5, 1.0, (size, size)).astype(np.float32))
    n_test = max(1, n_frames // 5)
    json.dump({"train": list(names[:-n_test]), "test": list(names[-n_test:])}, open(os.path.join(out, "splits.json"), "w"))
    i2s = {int(i + 1): int(c) for i, c in enumerate(sphere_class)}
    pickle.dump({"fg_classes": [4, 2, 3], "bg_classes": [1, 0], "m2f_instance_to_semantic": i2s, "rs_instance_to_semantic": i2s},
                open(os.path.join(out, "segmentation_data.pkl"), "wb"))
    if invalid_frames:
        os.makedirs(os.path.join(out, "invalid"), exist_ok=True)
        for f in invalid_frames:
            m = np.zeros((size, size), np.uint8)
            m[size // 4: size // 2, size // 3: (2 * size) // 3] = 255
            Image.fromarray(m).save(os.path.join(out, "invalid", f"{names[f]}.jpg"), quality=95)
    return out


if __name__ == "__main__":
    print(make_scene(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12, int(sys.argv[3]) if len(sys.argv) > 3 else 32))
