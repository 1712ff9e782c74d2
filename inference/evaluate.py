#!/usr/bin/env python3
"""Drop-in for reference inference/evaluate.py: scene-level mIoU and PQ_scene from the PNG folders written by
render_panopli.py against the dataset's ground truth (reference dataset/preprocessing/preprocess_scannet.py:622-732).

    python inference/evaluate.py --root_path <scene dir> --exp_path runs/<scene>_test_<experiment> --MOS

Scene-level PQ: the predictions / targets of ALL test frames are concatenated before matching, so an instance id has to
be consistent across views to count (that is what "PQ_scene" measures).  Note: the reference script unpacks
``pq, rq, sq = f(...)`` from a function that returns ``(pq, sq, rq)`` and therefore prints SQ and RQ under swapped
labels (inference/evaluate.py:26,31); here the labels are correct and ``metrics.txt`` says so.
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from contrastive_lift_amd.inference import ConfusionMatrix                 # noqa: E402
from contrastive_lift_amd.metrics import panoptic_quality                   # noqa: E402


def read_png(path, size):
    # preprocess_scannet.py:594-596: ``size`` goes to PIL unchanged, i.e. it is read as (width, height); the reference always
    # evaluates at (512, 512).  int64 so that 16-bit id PNGs and 8-bit label PNGs mix whatever Pillow decodes them to.
    return np.array(Image.open(path).resize(tuple(size), Image.NEAREST)).astype(np.int64)


def read_npy(path, size):
    return np.array(Image.fromarray(np.load(path).astype(np.int16)).resize(tuple(size), Image.NEAREST)).astype(np.int64)


def _mos_val_names(target_dir):
    names = sorted([x.stem for x in Path(target_dir).iterdir() if x.name.endswith(".npy")], key=lambda y: int(y) if y.isnumeric() else y)
    return set(names[int(len(names) * 0.8):])            # last 20 % of the frames (many_object_scenes.py:72)


def _pred_paths(pred_dir, val_names):
    return [y for y in sorted(Path(pred_dir).iterdir(), key=lambda x: int(x.stem)) if y.stem in val_names]


def evaluate_mos(exp_path, root_path, image_dim):
    exp_path, root_path = Path(exp_path), Path(root_path)
    val = _mos_val_names(root_path / "semantic")
    cm = ConfusionMatrix(num_classes=2, ignore_class=[])
    pred, target = [], []
    for p in _pred_paths(exp_path / "pred_semantics", val):
        ps = read_png(p, image_dim)
        ts = read_npy(root_path / "semantic" / f"{p.stem}.npy", image_dim)
        cm.add_batch(ps, ts)                               # the reference passes (pred, target) (:654)
        pi = read_png(exp_path / "pred_surrogateid" / p.name, image_dim)
        ti = read_npy(root_path / "instance" / f"{p.stem}.npy", image_dim)
        pred.append(np.stack([ps.reshape(-1), pi.reshape(-1)], -1))
        target.append(np.stack([ts.reshape(-1), ti.reshape(-1)], -1))
    pq, sq, rq = panoptic_quality(torch.from_numpy(np.concatenate(pred).astype(np.int64)), torch.from_numpy(np.concatenate(target).astype(np.int64)),
                                  {1}, {0}, allow_unknown_preds_category=True)
    return cm.get_miou(), float(pq), float(sq), float(rq)


def evaluate_panopli(exp_path, root_path, image_dim, is_thing):
    """ScanNet-style layout: rs_semantics / rs_instance PNGs, splits.json['test'], void class 0 masked out."""
    exp_path, root_path = Path(exp_path), Path(root_path)
    val = set(json.loads((root_path / "splits.json").read_text())["test"])
    things = {i for i, t in enumerate(is_thing) if t}
    stuff = {i for i, t in enumerate(is_thing) if not t}
    cm = ConfusionMatrix(num_classes=len(is_thing), ignore_class=[])
    pred, target = [], []
    for p in _pred_paths(exp_path / "pred_semantics", val):
        ts = read_png(root_path / "rs_semantics" / p.name, image_dim)
        valid = ~np.isin(ts, [0])
        ps = read_png(p, image_dim)
        cm.add_batch(ps[valid], ts[valid])                 # (pred, target) as :633
        pi = read_png(exp_path / "pred_surrogateid" / p.name, image_dim)
        ti = read_png(root_path / "rs_instance" / p.name, image_dim)
        pred.append(np.stack([ps[valid], pi[valid]], -1))
        target.append(np.stack([ts[valid], ti[valid]], -1))
    pq, sq, rq = panoptic_quality(torch.from_numpy(np.concatenate(pred).astype(np.int64)), torch.from_numpy(np.concatenate(target).astype(np.int64)),
                                  things, stuff, allow_unknown_preds_category=True)
    return cm.get_miou(), float(pq), float(sq), float(rq)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="metrics")
    ap.add_argument("--root_path", required=False)
    ap.add_argument("--exp_path", required=False)
    ap.add_argument("--MOS", action="store_true")
    ap.add_argument("--image_dim", type=int, nargs=2, default=[512, 512])
    ap.add_argument("--things_csv", default="resources/scannet_reduced_things.csv", help="name,is_thing rows (non-MOS only)")
    a = ap.parse_args()
    if a.MOS:
        iou, pq, sq, rq = evaluate_mos(a.exp_path, a.root_path, tuple(a.image_dim))
    else:
        rows = [l.split(",") for l in Path(a.things_csv).read_text().strip().splitlines()]
        is_thing = [False] + [bool(int(r[1])) for r in rows]           # class 0 = void
        iou, pq, sq, rq = evaluate_panopli(a.exp_path, a.root_path, tuple(a.image_dim), is_thing)
    print(f"[dataset] iou, pq, sq, rq: {iou:.3f}, {pq:.3f}, {sq:.3f}, {rq:.3f}")
    with open(Path(a.exp_path, "metrics.txt"), "w") as f:
        f.write(f"iou, pq, sq, rq: {iou:.3f}, {pq:.3f}, {sq:.3f}, {rq:.3f}")
