"""PanopLi-layout scene reader (ScanNet / Replica / Hypersim scenes as prepared by Panoptic Lifting) -- the on-disk layout of
reference dataset/panopli.py:42-198 feeding the ray / label tables the hot path consumes (BASELINE
configs[1] names a ScanNet scene).

Layout: ``color/<frame>.jpg``, ``intrinsic/intrinsic_color.txt`` (3x3 or 4x4), ``pose/<frame>.txt`` (4x4 camera-to-world),
``<semantics_dir>/<frame>.png``, ``<instance_dir>/<frame>.png``, ``<prefix>_probabilities/<frame>.npz`` with ``probability``
(h, w, C) and ``confidence`` (h, w) (``confidence_notta`` for the *notta* directories), ``splits.json`` (train / test or val),
``segmentation_data.pkl`` (fg_classes, bg_classes, <prefix>_instance_to_semantic), optional ``invalid/<frame>.jpg``.
Defaults follow dataset/__init__.py:14 (``m2f_semantics`` / ``m2f_instance`` / ``m2f_instance_to_semantic``).

Like the MOS reader, rays are generated on the device per frame (clift_gen_rays) instead of being stored 8 floats per pixel.
"""
import json
import os
import pickle

import numpy as np
import torch
from PIL import Image

from .mos import SceneTables


def _read_matrix(path):
    return np.array([[float(y) for y in x.split()] for x in open(path).read().splitlines() if x.strip() != ""], dtype=np.float64)


class PanopLiScene(SceneTables):
    def __init__(self, root_dir, split, image_dim, max_depth, subsample_frames=1, device="cuda", semantics_dir="m2f_semantics",
                 instance_dir="m2f_instance", instance_to_semantic_key="m2f_instance_to_semantic", num_val_samples=8, seed=None):
        self.root = str(root_dir)
        self.split = split
        self.image_dim = (int(image_dim[0]), int(image_dim[1]))
        self.device = torch.device(device)
        self.semantics_dir, self.instance_dir = semantics_dir, instance_dir
        names = [os.path.splitext(f)[0] for f in os.listdir(os.path.join(self.root, "color")) if f.endswith(".jpg")]
        self.all_frame_names = sorted(names, key=lambda y: int(y) if y.isnumeric() else y)
        idx = list(range(len(self.all_frame_names)))
        sp = os.path.join(self.root, "splits.json")
        if os.path.exists(sp):                                    # :52-66 with full_train_set_mode = True (:37)
            js = json.load(open(sp))
            self.train_indices = [self.all_frame_names.index(f"{x}") for x in js["train"]]
            key = "test" if (split != "test" or "test" in js) else "val"
            self.val_indices = [self.all_frame_names.index(f"{x}") for x in js[key]]
        else:                                                     # :67-69 (the reference draws this split from np.random)
            rng = np.random.default_rng(seed)
            self.val_indices = list(rng.choice(idx, min(len(idx), num_val_samples)))
            self.train_indices = [i for i in idx if i not in self.val_indices]
        self.train_indices = self.train_indices[::subsample_frames]
        self.val_indices = self.val_indices[::subsample_frames]
        img_h, img_w = np.array(Image.open(os.path.join(self.root, "color", f"{self.all_frame_names[0]}.jpg"))).shape[:2]
        K = _read_matrix(os.path.join(self.root, "intrinsic", "intrinsic_color.txt"))[:3, :3]
        poses = [_read_matrix(os.path.join(self.root, "pose", f"{n}.txt")) for n in self.all_frame_names]
        self._finish_cameras(K, poses, img_h, img_w, max_depth)
        seg = pickle.load(open(os.path.join(self.root, "segmentation_data.pkl"), "rb"))      # panopli.py:327-335
        fg, bg = sorted(seg["fg_classes"]), sorted(seg["bg_classes"])
        self.segmentation_data = type("Seg", (), dict(fg_classes=fg, bg_classes=bg, instance_to_semantics=seg.get(instance_to_semantic_key),
                                                      num_semantic_classes=len(fg) + len(bg), num_instances=len(fg)))()
        self.num_semantics = len(fg) + len(bg)

    def load_segments(self, sample_index):
        """:387-388: m2f_segments/<frame>.png, NEAREST-resized."""
        H, W = self.image_dim
        seg = Image.open(os.path.join(self.root, "m2f_segments", f"{self.all_frame_names[sample_index]}.png"))
        return torch.from_numpy(np.array(seg.resize((W, H), Image.NEAREST))).long().reshape(-1)

    def load_rs_targets(self, sample_index):
        """:215-225: ground-truth labels of a validation view (``rs_semantics`` / ``rs_instance`` folders), NEAREST-resized."""
        H, W = self.image_dim
        name = self.all_frame_names[sample_index]
        sem = Image.open(os.path.join(self.root, "rs_semantics", f"{name}.png"))
        inst = Image.open(os.path.join(self.root, "rs_instance", f"{name}.png"))
        return dict(rs_semantics=torch.from_numpy(np.array(sem.resize((W, H), Image.NEAREST))).long().reshape(-1),
                    rs_instances=torch.from_numpy(np.array(inst.resize((W, H), Image.NEAREST))).long().reshape(-1))

    def load_targets(self, sample_index):
        """:129-198 minus the rays: rgb (HW,3), semantics (HW,), instances (HW,), probabilities (HW,C), confidences (HW,), mask."""
        H, W = self.image_dim
        name = self.all_frame_names[sample_index]
        image = Image.open(os.path.join(self.root, "color", f"{name}.jpg"))
        rgb = torch.from_numpy(np.array(image.resize((W, H), Image.LANCZOS)) / 255).float()
        sem = Image.open(os.path.join(self.root, self.semantics_dir, f"{name}.png"))
        inst = Image.open(os.path.join(self.root, self.instance_dir, f"{name}.png"))
        sem_t = torch.from_numpy(np.array(sem.resize((W, H), Image.NEAREST))).long()
        prefix = self.semantics_dir.split("_")[0]
        if prefix != "rs":
            npz = np.load(os.path.join(self.root, f"{prefix}_probabilities", f"{name}.npz"))
            probs, conf = torch.from_numpy(npz["probability"]), torch.from_numpy(npz["confidence"])
            if "notta" in self.semantics_dir:
                conf = torch.from_numpy(npz["confidence_notta"]) if "confidence_notta" in npz else torch.ones_like(conf)
        else:      # ground-truth labels: one-hot probabilities, unit confidences (the reference sizes the one-hot by its class csv)
            probs = torch.nn.functional.one_hot(sem_t, num_classes=self.num_semantics).float()
            conf = torch.ones_like(probs)[..., 0]
        # joint bilinear resize to image_dim[::-1] = (W, H), exactly as :148 (same quirk as the MOS reader for non-square dims)
        both = torch.cat([probs.permute(2, 0, 1).float(), conf.float()[None]], 0)[None]
        both = torch.nn.functional.interpolate(both, size=(W, H), mode="bilinear", align_corners=False)[0]
        probs_t, conf_t = both[:-1].permute(1, 2, 0), both[-1]
        inst_t = torch.from_numpy(np.array(inst.resize((W, H), Image.NEAREST))).long()
        return dict(rgbs=rgb.reshape(-1, 3), semantics=sem_t.reshape(-1), instances=inst_t.reshape(-1),
                    probabilities=probs_t.reshape(-1, probs_t.shape[-1]).contiguous(), confidences=conf_t.reshape(-1).contiguous(),
                    mask=self._room_mask(name))
