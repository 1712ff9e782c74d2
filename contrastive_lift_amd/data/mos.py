"""Messy-Rooms ("MOS") scene reader -- the on-disk layout of reference dataset/many_object_scenes.py:22-207 feeding the
ray / label tables the hot path consumes (SURVEY 8b 'Batch dict', 8f rank 4).

Layout: ``color/<frame>.png``, ``metadata.json`` (camera K normalised by image size, positions, quaternions in Blender
convention), ``detic_semantic/<frame>.npy``, ``detic_instance/<frame>.npy``, ``detic_probabilities/<frame>.npy``
(optional GT ``semantic/``, ``instance/``).  The last 20 % of the sorted frames are the val/test split (:72).

MI355X-first difference to the reference: rays are NOT precomputed on the host and shipped as 8 floats per pixel;
``rays_for`` generates a frame's ray table on the device (clift_gen_rays) from 16 + 9 floats per camera, and the training
tables live in HBM.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from ..rays import generate_ray_table


def quat_to_rot(q):
    """(w, x, y, z) unit quaternion -> 3x3 rotation (what pyquaternion.Quaternion(*q).rotation_matrix returns, :33)."""
    w, x, y, z = [float(v) for v in q]
    n = (w * w + x * x + y * y + z * z) ** 0.5
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def read_cameras(meta, H, W):
    """:22-40: K rows scaled by (W, H), absolute value; pose = [R|t] @ diag(1,-1,-1,1) (Blender -> OpenCV)."""
    K = np.array(meta["camera"]["K"], dtype=np.float64)
    K[0] *= W
    K[1] *= H
    K = np.abs(K)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    poses = []
    for t, q in zip(meta["camera"]["positions"], meta["camera"]["quaternions"]):
        P = np.eye(4)
        P[:3, :3] = quat_to_rot(q)
        P[:3, 3] = np.array(t)
        poses.append(P @ flip)
    return K, poses


def world_to_normscene(dims, intrinsics, cam2worlds, max_depth, rescale_factor=1.0):
    """util/camera.py:10-73: bounding sphere of three frustum corners per camera at depths max_depth and 0.01
    (centre = mean of the corner cloud, radius = farthest corner) mapped to the unit sphere."""
    corners_hw = np.array([[0, 1, 1], [1, 0, 1], [1, 1, 1]], dtype=np.float64)
    pts = []
    for (h, w), K, c2w in zip(dims, intrinsics, cam2worlds):
        Kinv = np.linalg.inv(np.asarray(K, np.float64)[[1, 0, 2]])       # K in (h, w) order
        for depth in (max_depth, 0.01):
            for c in corners_hw:
                cam = Kinv @ (c * np.array([h, w, 1.0])) * depth
                pts.append((np.asarray(c2w, np.float64) @ np.append(cam, 1.0))[:3])
    pts = np.stack(pts)
    center = pts.mean(0)
    radius = np.linalg.norm(pts - center, axis=1).max()
    s = 1.0 / (rescale_factor * radius)
    M = np.eye(4)
    M[:3, :3] *= s
    M[:3, 3] = -center * s
    return M


# Thing / stuff table both dataset classes of the reference read (many_object_scenes.py:51, panopli.py:29: ``get_thing_semantics()`` =
# [False] + column 2 of resources/scannet_reduced_things.csv, whatever the scene's own label set is): void, wall, floor, cabinet, bed, chair,
# sofa, table, door, window, counter, shelves, curtain, ceiling, refridgerator, television, person, toilet, sink, lamp, bag, otherprop.
# Data, pinned by golden G16 (``is_thing``).
SCANNET_REDUCED_IS_THING = (False, False, False, False, True, True, True, False, False, False, False, False, False, False, True, True, True, True,
                            True, False, True, False)


class SceneTables:
    """What the readers share once cameras and per-frame targets exist: device-side ray tables, HBM-resident training
    tables, pixel batches and per-image instance batches (reference: BaseDataset + Inconsistent*SingleDataset)."""

    faulty_classes = (0,)                                   # many_object_scenes.py:50, panopli.py:28
    is_thing = SCANNET_REDUCED_IS_THING

    @property
    def things_filtered(self):                              # many_object_scenes.py:212-214
        return {i for i, t in enumerate(self.is_thing) if t} - set(self.faulty_classes)

    @property
    def stuff_filtered(self):                               # many_object_scenes.py:216-218
        return {i for i, t in enumerate(self.is_thing) if not t} - set(self.faulty_classes)

    def __len__(self):
        return len(self.train_indices if self.split == "train" else self.val_indices)

    def frame_index(self, i):
        return (self.train_indices if self.split == "train" else self.val_indices)[i]

    def rays_for(self, sample_index):
        H, W = self.image_dim
        return generate_ray_table(H, W, self.intrinsics[sample_index], self.cam2normscene[sample_index], near=0.01, device=self.device)

    def trajectory_set(self, trajectory_name):
        """dataset/base.py:320-365 (get_trajectory_set(name, norm_scene=True) -> MainerTrajectoryDataset): the camera-to-scene
        matrices pickled in ``trajectories/<name>.pkl`` mapped into the normalised scene; every frame uses the intrinsics of
        frame 0 and is named by its index.  Returns [(name, rays (H*W, 8))] lazily generated on the device."""
        import pickle
        with open(os.path.join(self.root, "trajectories", f"{trajectory_name}.pkl"), "rb") as f:
            poses = pickle.load(f)
        H, W = self.image_dim
        s2n = torch.from_numpy(np.asarray(self.scene2normscene)).float()
        for idx, pose in enumerate(poses):
            c2n = s2n @ torch.from_numpy(np.asarray(pose)).float()
            yield f"{idx:04d}", generate_ray_table(H, W, self.intrinsics[0], c2n, near=0.01, device=self.device)

    def build_train_tables(self, instance_images=True):
        """All training pixels as HBM-resident tables (reference keeps them on the host and feeds 8 loader workers).
        ``instance_images``: also cut the per-image instance ray sets from these tables (the reference builds those from a
        SEPARATE dataset object at a fixed (128, 128) size, dataset/__init__.py:57,66 -- see ``build_instance_tables``)."""
        rays, tg = [], []
        for i in self.train_indices:
            rays.append(self.rays_for(i))
            tg.append({k: v.to(self.device) for k, v in self.load_targets(i).items()})
        self.tables = dict(rays=torch.cat(rays, 0), **{k: torch.cat([t[k] for t in tg], 0) for k in tg[0]})
        if instance_images:
            self._cut_instance_images(self.tables)
        return self.tables

    def _cut_instance_images(self, tables):
        hw = self.image_dim[0] * self.image_dim[1]
        self.instance_images = []
        for j in range(len(self.train_indices)):
            sl = slice(j * hw, (j + 1) * hw)
            m = tables["instances"][sl] != 0
            if bool(m.any()):
                # many_object_scenes.py:242-258 / panopli.py:211-238: pixels with a label, confidences of room-masked pixels forced to 0
                conf = tables["confidences"][sl] * tables["mask"][sl].to(torch.float32)
                self.instance_images.append(dict(rays=tables["rays"][sl][m], instances=tables["instances"][sl][m], confidences=conf[m]))

    def build_instance_tables(self):
        """Inconsistent*SingleDataset (many_object_scenes.py:209-286, panopli.py:200-238): one ray set per training frame that has
        labelled pixels.  The reference constructs it at a FIXED (128, 128) image size whatever ``image_dim`` is
        (dataset/__init__.py:57,66): call this on a scene constructed with image_dim=(128, 128)."""
        rays, tg = [], []
        for i in self.train_indices:
            rays.append(self.rays_for(i))
            tg.append({k: v.to(self.device) for k, v in self.load_targets(i).items() if k in ("instances", "confidences", "mask")})
        self._cut_instance_images(dict(rays=torch.cat(rays, 0), **{k: torch.cat([t[k] for t in tg], 0) for k in tg[0]}))
        return self.instance_images

    def build_segment_tables(self):
        """Segment*Dataset (many_object_scenes.py:334-395, panopli.py:372-432): for every training frame and every non-zero id
        of its 2D segment map, the rays of that segment and their confidences (room-masked pixels at 0).  The reference
        builds this at a FIXED (128, 128) image size (dataset/__init__.py:70,78); call it on a scene constructed that way."""
        H, W = self.image_dim
        self.segment_items = []
        for i in self.train_indices:
            rays = self.rays_for(i)
            t = self.load_targets(i)
            conf = (t["confidences"] * t["mask"].to(torch.float32)).to(self.device)
            seg = self.load_segments(i).to(self.device)
            for sid in torch.unique(seg).tolist():
                if sid != 0:
                    m = seg == sid
                    self.segment_items.append(dict(rays=rays[m], confidences=conf[m]))
        return self.segment_items

    def segment_batch(self, batch_size_segments, max_rays, batch_index):
        """One collated batch[2] (DataLoader(shuffle=False, drop_last=True) + collate_fn): ``batch_size_segments`` consecutive
        segments, each subsampled to ``max_rays`` rays, group = position of the segment in the batch."""
        n_batches = len(self.segment_items) // batch_size_segments
        if n_batches == 0:
            return None
        b0 = (batch_index % n_batches) * batch_size_segments
        rays, conf, group = [], [], []
        for j in range(batch_size_segments):
            it = self.segment_items[b0 + j]
            n = it["rays"].shape[0]
            if n > max_rays:
                sel = torch.randperm(n, device=self.device)[:max_rays]
                it = {k: v[sel] for k, v in it.items()}
            rays.append(it["rays"]); conf.append(it["confidences"])
            group.append(torch.full((it["rays"].shape[0],), j, dtype=torch.int32, device=self.device))
        return dict(rays=torch.cat(rays).contiguous(), confidences=torch.cat(conf).contiguous(), group=torch.cat(group), n_groups=batch_size_segments)

    def pixel_batch(self, batch_size, generator=None):
        """A batch drawn WITH replacement (benchmarks, tests).  Training uses ``epoch_order`` + ``pixel_batch_at``: the reference's
        DataLoader(shuffle=True, drop_last=True) visits every training pixel exactly once per epoch, and that is not a detail -- on the
        G22 schedule, batches drawn with replacement end 5 - 7 dB lower (tests/golden/make_schedule_golden.py, round 6)."""
        n = self.tables["rays"].shape[0]
        idx = torch.randint(0, n, (batch_size,), device=self.device, generator=generator)
        return {k: v[idx] for k, v in self.tables.items()}

    def epoch_order(self, seed, epoch, rank=0, world=1):
        """This rank's share of one epoch's pixel order: DataLoader(train_set, shuffle=True) under Lightning's DistributedSampler (T:434,
        trainer/__init__.py:97) -- ONE permutation of all training pixels per epoch, the same on every rank (seeded by seed and epoch),
        of which rank r takes the entries r, r + world, ...  Every pixel is visited exactly once per epoch across the ranks."""
        n = self.tables["rays"].shape[0]
        g = torch.Generator()                             # a CPU generator, like the DataLoader's RandomSampler
        g.manual_seed((int(seed) * 1000003 + int(epoch)) * 7919 + 17)
        return torch.randperm(n, generator=g)[rank::world].to(self.device)

    def pixel_batch_at(self, order, it, batch_size):
        """Batch ``it`` of an epoch order (drop_last: the CLI runs len(order) // batch_size steps; a longer ``steps_per_epoch`` wraps around)."""
        pos = (it * batch_size + torch.arange(batch_size, device=self.device)) % order.shape[0]
        idx = order[pos]
        return {k: v[idx] for k, v in self.tables.items()}

    def instance_batch(self, max_rays, image_index):
        img = self.instance_images[image_index % len(self.instance_images)]
        n = img["rays"].shape[0]
        if n > max_rays:
            sel = torch.randperm(n, device=self.device)[:max_rays]
            img = {k: v[sel] for k, v in img.items()}
        return [dict(rays=img["rays"].contiguous(), instances=img["instances"], confidences=img["confidences"].contiguous())]

    def _finish_cameras(self, K, poses, img_h, img_w, max_depth):
        """Shared tail of setup_data: frustum-sphere normalisation, per-frame scaled intrinsics and normalised cameras."""
        n = len(poses)
        idx = list(range(n))
        self.scene2normscene = world_to_normscene([[img_h, img_w]] * n, [K] * n, poses, max_depth, 1.0)
        self.normscene_scale = float(self.scene2normscene[0, 0])
        scale = np.diag([self.image_dim[1] / img_w, self.image_dim[0] / img_h, 1.0])
        self.intrinsics = {i: torch.from_numpy(scale @ K).float() for i in idx}
        self.cam2normscene = {i: torch.from_numpy(self.scene2normscene @ poses[i]).float() for i in idx}
        self.scene_bounds = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
        self.white_bg = False

    def _room_mask(self, name):
        """invalid/<frame>.jpg, non-zero = pixel excluded (True in the returned mask = pixel is used)."""
        H, W = self.image_dim
        mpath = os.path.join(self.root, "invalid", f"{name}.jpg")
        if os.path.exists(mpath):
            return ~torch.from_numpy(np.array(Image.open(mpath).resize((W, H), Image.NEAREST)) > 0).bool().reshape(-1)
        return torch.ones(H * W, dtype=torch.bool)


class MOSScene(SceneTables):
    def __init__(self, root_dir, split, image_dim, max_depth, subsample_frames=1, device="cuda",
                 semantics_dir="detic_semantic", instance_dir="detic_instance"):
        self.root = str(root_dir)
        self.split = split
        self.image_dim = (int(image_dim[0]), int(image_dim[1]))
        self.device = torch.device(device)
        self.semantics_dir, self.instance_dir = semantics_dir, instance_dir
        names = [os.path.splitext(f)[0] for f in os.listdir(os.path.join(self.root, "color")) if f.endswith(".png")]
        self.all_frame_names = sorted(names, key=lambda y: int(y) if y.isnumeric() else y)
        n = len(self.all_frame_names)
        idx = list(range(n))
        self.val_indices = idx[int(n * 0.8):][::subsample_frames]
        vs = set(idx[int(n * 0.8):])
        self.train_indices = [i for i in idx if i not in vs][::subsample_frames]
        img = np.array(Image.open(os.path.join(self.root, "color", f"{self.all_frame_names[0]}.png")))
        img_h, img_w = img.shape[:2]
        meta = json.load(open(os.path.join(self.root, "metadata.json")))
        K, poses = read_cameras(meta, img_h, img_w)
        self._finish_cameras(K, poses, img_h, img_w, max_depth)
        self.segmentation_data = type("Seg", (), dict(fg_classes=[1], bg_classes=[0], num_semantic_classes=2, num_instances=1))()
        self.num_semantics = 2

    def load_segments(self, sample_index):
        """:349-351: the 2D segment map of a frame = the machine-generated instance map (detic_instance), NEAREST-resized."""
        H, W = self.image_dim
        seg = np.load(os.path.join(self.root, "detic_instance", f"{self.all_frame_names[sample_index]}.npy"))
        return torch.from_numpy(np.array(Image.fromarray(seg.astype(np.int16)).resize((W, H), Image.NEAREST))).long().reshape(-1)

    def load_rs_targets(self, sample_index):
        """:220-231: ground-truth labels of a validation view (``semantic`` / ``instance`` folders), NEAREST-resized."""
        H, W = self.image_dim
        name = self.all_frame_names[sample_index]
        sem = Image.fromarray(np.load(os.path.join(self.root, "semantic", f"{name}.npy")).astype(np.uint8))
        inst = Image.fromarray(np.load(os.path.join(self.root, "instance", f"{name}.npy")).astype(np.int16))
        return dict(rs_semantics=torch.from_numpy(np.array(sem.resize((W, H), Image.NEAREST))).long().reshape(-1),
                    rs_instances=torch.from_numpy(np.array(inst.resize((W, H), Image.NEAREST))).long().reshape(-1))

    def load_targets(self, sample_index):
        """:146-207 minus the rays: rgb (HW,3), semantics (HW,), instances (HW,), probabilities (HW,2), confidences (HW,)."""
        H, W = self.image_dim
        name = self.all_frame_names[sample_index]
        image = Image.open(os.path.join(self.root, "color", f"{name}.png"))
        rgb = torch.from_numpy(np.array(image.resize((W, H), Image.LANCZOS)) / 255).float()[..., :3]
        sem = np.load(os.path.join(self.root, self.semantics_dir, f"{name}.npy"))
        inst = np.load(os.path.join(self.root, self.instance_dir, f"{name}.npy"))
        prefix = self.semantics_dir.split("_")[0]
        if prefix != "semantic":
            conf = np.load(os.path.join(self.root, f"{prefix}_probabilities", f"{name}.npy")).astype(np.float32)
            conf[sem == 0] = 1.0
        else:
            conf = np.ones_like(sem).astype(np.float32)
        sem_t = torch.from_numpy(np.array(Image.fromarray(sem.astype(np.uint8)).resize((W, H), Image.NEAREST))).long()
        inst_t = torch.from_numpy(np.array(Image.fromarray(inst.astype(np.int16)).resize((W, H), Image.NEAREST))).long()
        # size = image_dim[::-1] = (W, H) exactly as the reference (:171): a no-op distinction for the square training images
        # (TI:71); for non-square image_dim the reference's confidence table is the (W, H)-shaped resize flattened (golden G14)
        conf_t = torch.nn.functional.interpolate(torch.from_numpy(conf).float()[None, None], size=(W, H), mode="bilinear",
                                                 align_corners=False)[0, 0]
        probs = torch.nn.functional.one_hot(sem_t, num_classes=self.num_semantics).float()
        mask = self._room_mask(name)
        return dict(rgbs=rgb.reshape(-1, 3), semantics=sem_t.reshape(-1), instances=inst_t.reshape(-1),
                    probabilities=probs.reshape(-1, self.num_semantics), confidences=conf_t.reshape(-1), mask=mask)
