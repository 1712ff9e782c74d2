#!/usr/bin/env python3
"""Write a small synthetic scene in the Messy-Rooms ("MOS") on-disk layout (color/*.png, metadata.json,
detic_semantic/*.npy, detic_instance/*.npy, detic_probabilities/*.npy, semantic/, instance/): a handful of shaded
spheres over a ground disc, rendered analytically (ray-sphere intersection).  Used by the end-to-end tests; no dataset
is available offline.   Usage: tools/make_synthetic_mos.py <out_dir> [n_frames] [size]"""
import json
import os
import sys

import numpy as np
from PIL import Image


def rot_to_quat(R):
    """3x3 rotation -> (w,x,y,z)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


def look_at_cv(eye, target=(0, 0, 0), up=(0, 0, 1)):
    """OpenCV camera-to-world rotation (x right, y down, z forward)."""
    eye = np.asarray(eye, float)
    f = np.asarray(target, float) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, float))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    return np.stack([r, d, f], 1)


def make_scene(out, n_frames=30, size=64, seed=0, invalid_frames=(), trajectory_frames=0):
    rng = np.random.default_rng(seed)
    for d in ("color", "detic_semantic", "detic_instance", "detic_probabilities", "semantic", "instance"):
        os.makedirs(os.path.join(out, d), exist_ok=True)
    centers = np.array([[0.0, 0.0, 0.35], [0.75, 0.15, 0.3], [-0.6, 0.55, 0.28], [-0.2, -0.75, 0.3], [0.5, -0.6, 0.25]])
    radii = np.array([0.35, 0.3, 0.28, 0.3, 0.25])
    colors = np.array([[0.9, 0.2, 0.2], [0.2, 0.8, 0.3], [0.2, 0.3, 0.9], [0.9, 0.8, 0.2], [0.8, 0.3, 0.8]])
    fx = 1.1 * size
    K = np.array([[fx, 0, size / 2], [0, fx, size / 2], [0, 0, 1.0]])
    light = np.array([0.4, -0.3, 0.85]); light /= np.linalg.norm(light)
    positions, quats = [], []
    jj, ii = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    dirs_cam = np.stack([(ii - K[0, 2]) / K[0, 0], (jj - K[1, 2]) / K[1, 1], np.ones_like(ii, float)], -1).reshape(-1, 3)
    for f in range(n_frames):
        az = 2 * np.pi * rng.uniform()
        el = rng.uniform(0.35, 1.0)
        eye = 3.2 * np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
        Rcv = look_at_cv(eye, target=(0, 0, 0.25))
        d = dirs_cam @ Rcv.T
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tbest = np.full(d.shape[0], np.inf)
        inst = np.zeros(d.shape[0], np.int64)
        rgb = np.tile(np.array([0.08, 0.08, 0.1]), (d.shape[0], 1))
        # ground disc z = 0, radius 1.6
        tg = -eye[2] / d[:, 2]
        pg = eye + tg[:, None] * d
        hit = (tg > 0) & (np.linalg.norm(pg[:, :2], axis=1) < 1.6)
        checker = ((np.floor(pg[:, 0] * 2) + np.floor(pg[:, 1] * 2)) % 2)
        rgb[hit] = (0.35 + 0.2 * checker[hit])[:, None] * np.array([1.0, 0.95, 0.9])
        tbest[hit] = tg[hit]
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
        sem = (inst > 0).astype(np.int64)
        name = f"{f:04d}"
        Image.fromarray((rgb.reshape(size, size, 3).clip(0, 1) * 255).astype(np.uint8)).save(os.path.join(out, "color", name + ".png"))
        # per-view inconsistent instance ids (what a 2D segmenter gives): a random permutation per frame
        perm = np.concatenate([[0], 1 + rng.permutation(len(centers))])
        for dname, arr in (("detic_semantic", sem), ("semantic", sem), ("instance", inst), ("detic_instance", perm[inst])):
            np.save(os.path.join(out, dname, name + ".npy"), arr.reshape(size, size))
        np.save(os.path.join(out, "detic_probabilities", name + ".npy"), rng.uniform(0.7, 1.0, (size, size)).astype(np.float32))
        Rbl = Rcv @ np.diag([1.0, -1.0, -1.0])                       # OpenCV -> Blender camera axes
        positions.append(eye.tolist())
        quats.append(rot_to_quat(Rbl).tolist())
    meta = {"camera": {"K": [[K[0, 0] / size, 0, K[0, 2] / size], [0, K[1, 1] / size, K[1, 2] / size], [0, 0, 1]],
                       "positions": positions, "quaternions": quats}}
    json.dump(meta, open(os.path.join(out, "metadata.json"), "w"))
    # optional predefined camera path (MOS / PanopLi layout: trajectories/trajectory_blender.pkl = list of 4x4 camera-to-scene
    # matrices, OpenCV axes): an orbit at fixed elevation
    if trajectory_frames:
        import pickle
        os.makedirs(os.path.join(out, "trajectories"), exist_ok=True)
        traj = []
        for j in range(trajectory_frames):
            az = 2 * np.pi * j / trajectory_frames
            eye = 3.0 * np.array([np.cos(az) * np.cos(0.6), np.sin(az) * np.cos(0.6), np.sin(0.6)])
            P = np.eye(4)
            P[:3, :3], P[:3, 3] = look_at_cv(eye, target=(0, 0, 0.25)), eye
            traj.append(P)
        pickle.dump(traj, open(os.path.join(out, "trajectories", "trajectory_blender.pkl"), "wb"))
    # optional "invalid" room masks (MOS layout: invalid/<frame>.jpg, non-zero = pixel excluded): a rectangle per listed frame
    if invalid_frames:
        os.makedirs(os.path.join(out, "invalid"), exist_ok=True)
        for f in invalid_frames:
            m = np.zeros((size, size), np.uint8)
            m[size // 4: size // 2, size // 3: (2 * size) // 3] = 255
            Image.fromarray(m).save(os.path.join(out, "invalid", f"{f:04d}.jpg"), quality=95)
    return out


if __name__ == "__main__":
    print(make_scene(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30, int(sys.argv[3]) if len(sys.argv) > 3 else 64))
