"""CPU checks of the config-tree loader and the MOS reader's camera conventions (no kernels)."""
import json
import os
import sys

import numpy as np
import torch

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))


def test_config_tree_matches_reference_interface():
    from contrastive_lift_amd.config import load_config, save_config, load_run_config
    c = load_config(os.path.join(REPO, "config"), overrides=["+experiment=contrastive_lift_MOS", "template.lr=1e-3", "dataset_root=/x/scene_1"])
    assert c.dataset_class == "mos" and c.instance_loss_mode == "slow_fast" and c.use_DINO_style is True and c.max_instances == 3
    assert isinstance(c.lr, float) and c.lr == 1e-3 and isinstance(c.weight_decay, float) and c.weight_decay == 1e-8
    assert c.batch_size == 2048 and c.chunk == 2048 and c.max_rays_instances == 1024 and c.min_grid_dim == 128 and c.max_grid_dim == 192
    assert c.bbox_aabb_reset_epochs == [1, 2, 3] and c.grid_upscale_epochs == [1, 2, 3, 4] and c.late_semantic_optimization == 2
    d = load_config(os.path.join(REPO, "config"))
    assert d.instance_loss_mode == "linear_assignment" and d.max_instances == 25 and d.dataset_class == "panopli"
    p = os.path.join("/tmp", "clift_cfg_test", "config.yaml")
    save_config(c, p)
    assert load_run_config(p).dataset_root == "/x/scene_1"


def test_mos_camera_conventions(tmp_path):
    import make_synthetic_mos as gen
    from contrastive_lift_amd.data.mos import read_cameras, quat_to_rot, world_to_normscene
    out = gen.make_scene(str(tmp_path / "scene"), n_frames=6, size=16)
    meta = json.load(open(os.path.join(out, "metadata.json")))
    K, poses = read_cameras(meta, 16, 16)
    assert abs(K[0, 0] - 1.1 * 16) < 1e-9 and abs(K[0, 2] - 8) < 1e-9
    for P, pos in zip(poses, meta["camera"]["positions"]):
        R = P[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
        fwd = R[:, 2]                                   # OpenCV: +z looks at the scene centre
        to_c = np.array([0, 0, 0.25]) - np.array(pos)
        assert np.dot(fwd, to_c / np.linalg.norm(to_c)) > 0.999
    q = np.array([0.5, 0.5, 0.5, 0.5])
    assert np.allclose(quat_to_rot(q), np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]]), atol=1e-12)
    M = world_to_normscene([[16, 16]] * 6, [K] * 6, poses, max_depth=5.0)
    for P in poses:                                     # every camera ends up inside the unit sphere
        assert np.linalg.norm((M @ P)[:3, 3]) < 1.0


def _g14_scene(tmp_path, g):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_mos as gen
    return gen.make_scene(str(tmp_path / "scene"), n_frames=int(g["n_frames"]), size=int(g["size"]), seed=int(g["seed"]),
                          invalid_frames=(int(g["invalid_frame"]),), trajectory_frames=4)


def test_g14_mos_reader_vs_reference_dataset(tmp_path):
    """MOSScene against the REFERENCE's MOSDataset run on the same files (golden G14): split, scene normalisation, per-frame
    intrinsics and camera matrices, and every per-pixel target table (image LANCZOS resize, NEAREST labels, bilinear
    confidences with the background override, one-hot probabilities, room mask) -- native and resized image_dim.
    The ray tables (device kernel) are compared in tests/test_gpu_parity.py."""
    import numpy as np
    import torch
    from conftest import load_golden, rel_close
    from contrastive_lift_amd.data import MOSScene
    g = load_golden("g14_mos_dataset")
    root = _g14_scene(tmp_path, g)
    for tag in ("native", "resized"):
        dim = tuple(int(x) for x in g[f"{tag}.dim"])
        sc = MOSScene(root, "train", dim, float(g["max_depth"]), device="cpu")
        assert sc.train_indices == list(g[f"{tag}.train_indices"]) and sc.val_indices == list(g[f"{tag}.val_indices"])
        rel_close(torch.from_numpy(sc.scene2normscene).float(), g[f"{tag}.scene2normscene"], 1e-5, atol=1e-6, what="scene2normscene")
        assert torch.equal(sc.scene_bounds, torch.from_numpy(g[f"{tag}.scene_bounds"]))
        for f in (int(x) for x in g["frames"]):
            rel_close(sc.intrinsics[f], g[f"{tag}.f{f}.K"], 1e-6, what="K")
            rel_close(sc.cam2normscene[f], g[f"{tag}.f{f}.cam2normscene"], 1e-5, atol=1e-6, what="cam2normscene")
            t = sc.load_targets(f)
            rel_close(t["rgbs"], g[f"{tag}.f{f}.rgbs"], 1e-6, what="rgbs")
            assert torch.equal(t["semantics"], torch.from_numpy(g[f"{tag}.f{f}.semantics"]))
            assert torch.equal(t["instances"], torch.from_numpy(g[f"{tag}.f{f}.instances"]))
            assert torch.equal(t["probabilities"], torch.from_numpy(g[f"{tag}.f{f}.probabilities"]))
            rel_close(t["confidences"], g[f"{tag}.f{f}.confidences"], 1e-6, what="confidences")
            assert torch.equal(t["mask"], torch.from_numpy(g[f"{tag}.f{f}.mask"]))
        assert int((~sc.load_targets(int(g["invalid_frame"]))["mask"]).sum()) > 0


def test_g15_panopli_reader_vs_reference_dataset(tmp_path):
    """PanopLiScene against the REFERENCE's PanopLiDataset run on the same files (golden G15): splits.json handling, text
    intrinsics / poses, scene normalisation, segmentation data, and every per-pixel target table (jpg LANCZOS resize, png NEAREST
    labels, jointly resized probabilities + confidences, room mask) -- native and resized image_dim."""
    import torch
    from conftest import load_golden, rel_close
    import make_synthetic_panopli as gen
    from contrastive_lift_amd.data import PanopLiScene
    g = load_golden("g15_panopli_dataset")
    root = gen.make_scene(str(tmp_path / "scene"), n_frames=int(g["n_frames"]), size=int(g["size"]), seed=int(g["seed"]),
                          invalid_frames=(int(g["invalid_frame"]),))
    for tag in ("native", "resized"):
        dim = tuple(int(x) for x in g[f"{tag}.dim"])
        sc = PanopLiScene(root, "train", dim, float(g["max_depth"]), device="cpu")
        st = PanopLiScene(root, "test", dim, float(g["max_depth"]), device="cpu")
        assert sc.train_indices == list(g[f"{tag}.train_indices"]) and sc.val_indices == list(g[f"{tag}.val_indices"])
        assert st.val_indices == list(g[f"{tag}.test_val_indices"])
        sd = sc.segmentation_data
        assert sd.fg_classes == list(g[f"{tag}.fg"]) and sd.bg_classes == list(g[f"{tag}.bg"])
        assert sd.num_semantic_classes == int(g[f"{tag}.num_classes"]) and sd.num_instances == int(g[f"{tag}.num_instances"])
        assert sorted(sd.instance_to_semantics.items()) == [tuple(x) for x in g[f"{tag}.i2s"].tolist()]
        rel_close(torch.from_numpy(sc.scene2normscene).float(), g[f"{tag}.scene2normscene"], 1e-5, atol=1e-6, what="scene2normscene")
        for f in (int(x) for x in g["frames"]):
            rel_close(sc.intrinsics[f], g[f"{tag}.f{f}.K"], 1e-6, what="K")
            rel_close(sc.cam2normscene[f], g[f"{tag}.f{f}.cam2normscene"], 1e-5, atol=1e-6, what="cam2normscene")
            t = sc.load_targets(f)
            rel_close(t["rgbs"], g[f"{tag}.f{f}.rgbs"], 1e-6, what="rgbs")
            assert torch.equal(t["semantics"], torch.from_numpy(g[f"{tag}.f{f}.semantics"]))
            assert torch.equal(t["instances"], torch.from_numpy(g[f"{tag}.f{f}.instances"]))
            rel_close(t["probabilities"], g[f"{tag}.f{f}.probabilities"], 1e-6, atol=1e-8, what="probabilities")
            rel_close(t["confidences"], g[f"{tag}.f{f}.confidences"], 1e-6, what="confidences")
            assert torch.equal(t["mask"], torch.from_numpy(g[f"{tag}.f{f}.mask"]))


def test_g16_scene_evaluators_vs_reference(tmp_path):
    """inference/evaluate.py (mIoU + PQ_scene over prediction folders) against the REFERENCE's evaluators
    (preprocess_scannet.py:622-732) on the same synthetic scenes and prediction folders: MOS and ScanNet-style layouts,
    square and non-square evaluation sizes."""
    import importlib.util
    import numpy as np
    from PIL import Image
    from conftest import load_golden
    import make_synthetic_mos as gen_m
    import make_synthetic_panopli as gen_p
    from make_fake_predictions import write_fake_predictions
    spec = importlib.util.spec_from_file_location("clift_eval", os.path.join(REPO, "inference", "evaluate.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    g = load_golden("g16_scene_evaluators")
    root = gen_m.make_scene(str(tmp_path / "mos"), n_frames=10, size=24, seed=7)
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(root, "semantic")))
    val = names[int(len(names) * 0.8):]
    rng = np.random.default_rng(161)
    write_fake_predictions(str(tmp_path / "mos_exp"), val, [np.load(os.path.join(root, "semantic", n + ".npy")) for n in val],
                           [np.load(os.path.join(root, "instance", n + ".npy")) for n in val], rng)
    for tag in ("sq", "ns"):
        got = ev.evaluate_mos(str(tmp_path / "mos_exp"), root, tuple(int(x) for x in g[f"mos.{tag}.dim"]))
        np.testing.assert_allclose(np.array(got), g[f"mos.{tag}.metrics"], rtol=1e-6, atol=1e-9)
    rootp = gen_p.make_scene(str(tmp_path / "pan"), n_frames=10, size=24, seed=5)
    test = [str(x) for x in json.load(open(os.path.join(rootp, "splits.json")))["test"]]
    rd = lambda d, n: np.array(Image.open(os.path.join(rootp, d, n + ".png")))
    write_fake_predictions(str(tmp_path / "pan_exp"), test, [rd("rs_semantics", n) for n in test], [rd("rs_instance", n) for n in test], rng)
    is_thing = [bool(x) for x in g["is_thing"]]
    assert len(is_thing) == int(g["num_classes_iou"])
    for tag in ("sq", "ns"):
        got = ev.evaluate_panopli(str(tmp_path / "pan_exp"), rootp, tuple(int(x) for x in g[f"pan.{tag}.dim"]), is_thing)
        np.testing.assert_allclose(np.array(got), g[f"pan.{tag}.metrics"], rtol=1e-6, atol=1e-9)


def test_g17_meanshift_clustering_vs_reference():
    """inference.cluster against the REFERENCE's cluster() (RP:196-263) on the same 72 k synthetic thing features with numpy's
    global generator seeded identically: identical per-pixel cluster labels and one-hot width, fixed bandwidth and Silverman."""
    import numpy as np
    import torch
    from conftest import load_golden
    from make_fake_predictions import fake_thing_features
    from contrastive_lift_amd.inference import cluster
    g = load_golden("g17_meanshift_clustering")
    all_thing, n_img = fake_thing_features(int(g["seed"]))
    assert all_thing.shape[0] == int(g["n_rows"]) and abs(float(np.where(np.isfinite(all_thing), all_thing, 0).sum()) - float(g["checksum"])) < 1e-6
    for tag, silver in (("bw", False), ("silverman", True)):
        np.random.seed(1234)
        onehot, cents = cluster(all_thing.copy(), 0.15, torch.device("cpu"), num_images=n_img, use_silverman=silver)
        assert onehot.shape == (n_img, all_thing.shape[0] // n_img, int(g[f"{tag}.width"])) and cents.shape[0] + 1 == onehot.shape[-1]
        assert np.array_equal(onehot.argmax(-1).reshape(-1).numpy().astype(np.int16), g[f"{tag}.labels"])


def test_g17_segmentwise_clustering_vs_reference():
    """inference.cluster_segmentwise against the REFERENCE's cluster_segmentwise (RP:265-368): per-class MeanShift with disjoint
    label offsets, a class below the 100-point minimum, the returned centroid list (incl. the reference's re-appended ones)."""
    import numpy as np
    import torch
    from conftest import load_golden
    from make_fake_predictions import fake_thing_features, fake_semantics_for
    from contrastive_lift_amd.inference import cluster_segmentwise
    g = load_golden("g17_meanshift_clustering")
    all_thing, n_img = fake_thing_features(int(g["seed"]))
    sems = fake_semantics_for(all_thing, n_img)
    np.random.seed(4321)
    onehot, cents = cluster_segmentwise(all_thing.copy(), sems, 0.15, torch.device("cpu"), num_images=n_img)
    assert onehot.shape[-1] == int(g["seg.width"])
    assert np.array_equal(onehot.argmax(-1).reshape(-1).numpy().astype(np.int16), g["seg.labels"])
    np.testing.assert_allclose(cents, g["seg.centroids"], rtol=1e-6, atol=1e-7)


def test_hdbscan_clustering_branch():
    """--use_dbscan (RP:236-255, 320-342).  The hdbscan package the reference imports is not in this image (PARITY UNPINNED against it): the fit
    comes from sklearn.cluster.HDBSCAN, the same algorithm.  Checked here on five well-separated 3-D blobs + stuff rows: every blob becomes
    exactly one cluster, every thing pixel gets its blob's label (nearest probability-weighted centroid), stuff pixels get label 0, the one-hot
    width is clusters + 1, the centroids (scaled back to feature units) sit on the blob means; segment-wise: the same inside every class with
    disjoint label ranges; a class of 30 scattered points (fewer than min_cluster_size) comes out as ONE cluster (allow_single_cluster: the root
    of a tree without a split is selected), as the algorithm prescribes."""
    import numpy as np
    import torch
    from contrastive_lift_amd.inference import cluster, cluster_segmentwise, _hdbscan_fit
    rng = np.random.default_rng(5)
    n_img, per_img, n_blob = 4, 3000, 5
    means = np.array([[0.0, 0.0, 0.0], [1.0, 0.2, -0.3], [-0.8, 0.9, 0.4], [0.3, -1.1, 0.8], [-0.5, -0.6, -1.0]])
    N = n_img * per_img
    blob = rng.integers(0, n_blob, N)
    feats = means[blob] + 0.03 * rng.standard_normal((N, 3))
    thing = rng.uniform(size=N) < 0.7
    all_thing = np.concatenate([np.where(thing, -np.inf, 0.0)[:, None], feats], 1).astype(np.float32)
    np.random.seed(1)
    onehot, cents = cluster(all_thing.copy(), 0.15, torch.device("cpu"), num_images=n_img, use_dbscan=True, cluster_size=50)
    lab = onehot.argmax(-1).reshape(-1).numpy()
    assert onehot.shape == (n_img, per_img, n_blob + 1) and cents.shape == (n_blob, 3)
    assert np.all(lab[~thing] == 0) and np.all(lab[thing] >= 1)
    for b in range(n_blob):                                           # one label per blob, and the labels of different blobs differ
        assert len(np.unique(lab[thing & (blob == b)])) == 1
    assert len({int(lab[thing & (blob == b)][0]) for b in range(n_blob)}) == n_blob
    for b in range(n_blob):
        k = int(lab[thing & (blob == b)][0]) - 1
        assert np.abs(cents[k] - means[b]).max() < 0.02
    # the centroid is the membership-probability-weighted mean of the cluster's points (hdbscan's weighted_cluster_centroid)
    pts = rng.standard_normal((400, 2)) * 0.05 + np.repeat(np.array([[0.0, 0.0], [1.0, 1.0]]), 200, 0)
    labels, c = _hdbscan_fit(pts, 20)
    from sklearn.cluster import HDBSCAN
    ref = HDBSCAN(min_cluster_size=20, min_samples=1, allow_single_cluster=True).fit(pts)
    assert np.array_equal(labels, ref.labels_) and c.shape == (2, 2)
    for k in range(2):
        m = ref.labels_ == k
        np.testing.assert_allclose(c[k], np.average(pts[m], weights=ref.probabilities_[m], axis=0), rtol=1e-12)
    # segment-wise: two thing classes holding blobs {0, 1, 2} and {3, 4}; a third class of 30 scattered points
    sem_cls = np.where(blob <= 2, 1, 2)
    scat = rng.choice(np.nonzero(thing)[0], 30, replace=False)
    sem_cls[scat] = 3
    all_thing[scat, 1:] = rng.uniform(-3, 3, (30, 3)).astype(np.float32)
    sems = [torch.nn.functional.one_hot(torch.from_numpy(sem_cls[i * per_img:(i + 1) * per_img]), 4).float() for i in range(n_img)]
    np.random.seed(2)
    onehot_s, cents_s = cluster_segmentwise(all_thing.copy(), sems, 0.15, torch.device("cpu"), num_images=n_img, use_dbscan=True, cluster_size=50)
    lab_s = onehot_s.argmax(-1).reshape(-1).numpy()
    keep = thing & (sem_cls != 3)
    assert cents_s.shape == (n_blob + 1, 3) and onehot_s.shape[-1] == n_blob + 2
    assert np.all(lab_s[~thing] == 0) and np.all(lab_s[scat] == n_blob + 1)
    ids = [np.unique(lab_s[keep & (blob == b)]) for b in range(n_blob)]
    assert all(len(i) == 1 for i in ids) and len({int(i[0]) for i in ids}) == n_blob
    assert {int(ids[b][0]) for b in (0, 1, 2)} == {1, 2, 3} and {int(ids[b][0]) for b in (3, 4)} == {4, 5}


def test_trainer_rejects_unbuilt_config_variants():
    """Options of the reference's config tree that the hot-path trainer does not implement raise instead of being ignored."""
    import pytest
    import contrastive_lift_amd as cl
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    m = cl.TensorVMSplit([6, 7, 8], num_semantic_classes=3, dim_feature_instance=6, use_semantic_mlp=True, use_instance_mlp=True,
                         slow_fast_mode=True, device="cpu")
    r = cl.TensoRFRenderer(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), [6, 7, 8], semantic_weight_mode="softmax")
    HotPathTrainer(m, r, default_config())                                   # the shipped settings construct fine
    HotPathTrainer(m, r, default_config(use_symmetric_ce=True))             # SCELoss is built (round 2)
    HotPathTrainer(m, r, default_config(probabilistic_ce_mode="NoTTAConf"))  # the label map as the target (round 6)
    HotPathTrainer(m, r, default_config(probabilistic_ce_mode="NoConf"))     # any other string: label map, no confidences (T:181-182; golden G12p)
    HotPathTrainer(m, r, default_config(optimize_instance_only=True))        # T:151: the main pass is skipped
    for k, v in (("use_proj", True), ("use_distilled_features_semantic", True)):
        with pytest.raises(NotImplementedError):
            HotPathTrainer(m, r, default_config(**{k: v}))


def test_every_experiment_overlay_of_the_reference_is_kept():
    """config/experiment/ holds all twelve overlays of the reference's tree and `+experiment=<name>` resolves to the same values as the
    reference's own files (golden G20: the reference's YAML files resolved over its template); `mlp_dtype` is this build's extension key."""
    from contrastive_lift_amd.config import load_config
    want = json.load(open(os.path.join(REPO, "tests", "golden", "g20_config_overlays.json")))
    assert len(want) == 13
    for name, ref in want.items():
        ours = dict(load_config(os.path.join(REPO, "config"), None if name == "<template only>" else name))
        assert ours.pop("mlp_dtype") == "fp32x6"
        assert ours == ref, (name, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)})


def test_linear_assignment_matching_on_the_host():
    """loss.create_virtual_gt_with_linear_assignment (device index_add + host Hungarian step) against the oracle's restatement of T:332-344 --
    which golden G12l pins against the reference trainer -- incl. more 2-D ids than output slots and ids without any matched slot."""
    from contrastive_lift_amd.loss import create_virtual_gt_with_linear_assignment as prod
    from oracle.losses import virtual_labels_linear_assignment as orc
    g = torch.Generator().manual_seed(0)
    for E, L, n in ((6, 8, 64), (3, 2, 50), (25, 40, 300), (4, 4, 10), (500, 30, 256)):
        for _ in range(4):
            y = torch.randint(1, L + 1, (n,), generator=g)
            f = 3.0 * torch.randn(n, E, generator=g)
            assert torch.equal(prod(y, f), orc(y, f)), (E, L, n)


def test_epoch_order_is_one_permutation_shared_by_the_ranks():
    """SceneTables.epoch_order (the reference's DataLoader(shuffle=True, drop_last=True) under Lightning's DistributedSampler, T:434): one
    permutation of all training pixels per epoch, identical on every rank, of which rank r takes entries r, r + world, ...: every pixel exactly
    once per epoch across the ranks, another order the next epoch; pixel_batch_at slices it (and wraps when asked for more steps)."""
    from contrastive_lift_amd.data.mos import SceneTables
    sc = SceneTables.__new__(SceneTables)
    sc.device = torch.device("cpu")
    n = 1003
    sc.tables = {"rays": torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 8), "rgbs": torch.zeros(n, 3)}
    shards = [sc.epoch_order(5, 2, r, 4) for r in range(4)]
    assert sorted(torch.cat(shards).tolist()) == list(range(n)) and max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert all(torch.equal(a, b) for a, b in zip(shards, [sc.epoch_order(5, 2, r, 4) for r in range(4)]))
    assert not torch.equal(sc.epoch_order(5, 2), sc.epoch_order(5, 3)) and not torch.equal(sc.epoch_order(5, 2), sc.epoch_order(6, 2))
    full = sc.epoch_order(5, 2)
    seen = torch.cat([sc.pixel_batch_at(full, it, 100)["rays"][:, 0] for it in range(n // 100)])
    assert len(set(seen.tolist())) == 100 * (n // 100)                       # drop_last: no pixel twice within an epoch
    b = sc.pixel_batch_at(full, n // 100 + 3, 100)                           # a longer steps_per_epoch wraps around the same order
    assert b["rays"].shape == (100, 8) and torch.equal(b["rays"][:, 0].long(), full[((n // 100 + 3) * 100 + torch.arange(100)) % n])


def test_checkpoint_optimizer_state_is_torch_adams_own_layout():
    """HotPathTrainer.checkpoint_dict writes both Adam states in torch.optim.Adam's own ``state_dict`` layout with the REFERENCE's parameter-group
    order (tensoRF.py:199-246; the group sizes are pinned by golden G12's ``opt_groups``), loadable into a torch.optim.Adam built over the
    reference's groups, and ``load_torch_state_dict`` restores exactly what was written; a shrink without an optimizer rebuild carries the MLPs'
    moments to their new arena offsets and freezes the cropped tables (ArenaAdam.carry_from: the reference's behaviour, golden G21 scenario B)."""
    import numpy as np
    import contrastive_lift_amd as cl
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    g = np.load(os.path.join(REPO, "tests", "golden", "g12_training_steps.npz"))
    res, C_, E = [int(x) for x in g["res"]], int(g["C"]), int(g["E"])
    m = cl.TensorVMSplit(res, num_semantics_comps=(32, 32, 32), num_instance_comps=(32, 32, 32), num_semantic_classes=C_, dim_feature_instance=2 * E,
                         use_semantic_mlp=True, use_instance_mlp=True, slow_fast_mode=True, device="cpu")
    r = cl.TensoRFRenderer(torch.tensor(g["aabb"]), res, semantic_weight_mode="softmax")
    tr = HotPathTrainer(m, r, default_config(max_instances=E), current_epoch=4)
    main, inst = tr.torch_param_groups()
    og = g["opt_groups"]                                            # rows: lr, weight_decay, beta1, beta2, numel -- main groups first
    numel = lambda names: sum(m.arena.by_name[n].shape[0] * int(np.prod(m.arena.by_name[n].shape[1:])) for n in names)
    assert [numel(ns) for _, ns in main + inst] == [int(x) for x in og[:, 4]]
    assert [lr for lr, _ in main + inst] == [float(x) for x in og[:, 0]]
    # pretend two steps happened: fill the moments, set the counts, write, load into a fresh optimizer, compare
    gen = torch.Generator().manual_seed(3)
    for opt in (tr.opt_main, tr.opt_inst):
        opt.m.copy_(torch.randn(opt.m.shape, generator=gen)); opt.v.copy_(torch.rand(opt.v.shape, generator=gen))
        opt.t = {k: 2 for k in opt.t}
    tr.opt_main.t["net_sem"] = 0                                    # the semantic MLP has not been stepped yet: no state entries, like torch's lazy state
    ck = tr.checkpoint_dict(global_step=2, epoch_complete=False)
    sd = ck["optimizer_states"][0]
    n_sem = len(main[-1][1])
    assert len(sd["state"]) == sum(len(ns) for _, ns in main) - n_sem and sd["param_groups"][0]["betas"] == (0.9, 0.99)
    ref_params = [[torch.nn.Parameter(torch.zeros(m.arena.by_name[n].shape)) for n in ns] for _, ns in main]
    topt = torch.optim.Adam([{"params": ps, "lr": lr} for ps, (lr, _) in zip(ref_params, main)], lr=5e-4, betas=(0.9, 0.99))
    topt.load_state_dict(sd)                                        # torch accepts it: same group structure, same per-parameter shapes
    assert float(topt.state[ref_params[2][0]]["step"]) == 2.0 and topt.state[ref_params[2][0]]["exp_avg"].shape == m.arena.by_name["density_plane.0"].shape
    tr2 = HotPathTrainer(m, r, default_config(max_instances=E), current_epoch=4)
    tr2.opt_main.load_torch_state_dict(sd, main)
    tr2.opt_inst.load_torch_state_dict(ck["optimizer_states"][1], inst)
    mv, mv2 = m.arena.views(tr.opt_main.m), m.arena.views(tr2.opt_main.m)
    assert all(torch.equal(mv[n], mv2[n]) for _, ns in main[:-1] for n in ns) and tr2.opt_main.t == {"grids": 2, "net_app": 2, "net_sem": 0}
    assert float(m.arena.views(tr2.opt_main.m)["render_semantic_mlp.mlp.0.weight"].abs().max()) == 0.0
    assert tr2.opt_inst.t["inst_fast"] == 2 and ck["lr_schedulers"][0]["last_epoch"] == 0
    # a shrink without a rebuild: the arena is re-packed, the MLPs' moments follow, the cropped tables leave the optimizer
    before = mv["render_appearance_mlp.mlp.0.weight"].clone()
    m.shrink([1, 1, 1], [r_ - 1 for r_ in res])
    tr.setup_optimizers(carry=True)
    assert "grids" in tr.opt_main.frozen and "net_app" not in tr.opt_main.frozen and tr.opt_main.t["net_app"] == 2
    assert torch.equal(m.arena.views(tr.opt_main.m)["render_appearance_mlp.mlp.0.weight"], before)
    tr.setup_optimizers()
    assert not tr.opt_main.frozen and tr.opt_main.t["net_app"] == 0 and float(tr.opt_main.m.abs().max()) == 0.0
