#!/usr/bin/env python3
"""Drop-in for reference inference/render_panopli.py (same flags, same inputs and outputs; RP:430-458 -> 31-188).

    python inference/render_panopli.py --ckpt_path runs/<experiment>/checkpoints/<x>.ckpt [--cached_centroids_path P]
                                       [--bandwidth 0.15] [--subsample 1] [--render_trajectory] ...

Reads ``runs/<experiment>/config.yaml`` next to the checkpoint (RP:445), rebuilds the field at ``min_grid_dim``, restores the
renderer buffers, upsamples to the checkpoint's grid if it was saved after an upscale epoch (RP:91-98), loads the weights,
halves the step ratio (RP:104) and renders every test frame in chunks on the GPU.  Writes ``instance_features.npy``,
``thing_features.npy``, ``slow_features.npy``, ``pred_semantics/*.png`` (uint8), ``pred_surrogateid/*.png`` (uint16) and
``vis_semantics_and_surrogate/*.png`` under ``runs/<scene>_<test|trajectory>_<experiment>/`` (RP:142-193).
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import contrastive_lift_amd as cl                                   # noqa: E402
from contrastive_lift_amd import inference as inf                    # noqa: E402
from contrastive_lift_amd.config import load_run_config              # noqa: E402
from contrastive_lift_amd.data import get_scene                       # noqa: E402


def strip_prefix(state_dict, key):
    """util/misc.py:159-164."""
    return {k[len(key) + 1:]: v for k, v in state_dict.items() if k.startswith(key + ".")}


def build_from_checkpoint(config, scene, device):
    from contrastive_lift_amd import engine
    engine.set_mlp_precision(getattr(config, "mlp_dtype", None) or engine.DEFAULT_MLP_DTYPE)
    ckpt = torch.load(config.resume, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"]
    total_classes = len(scene.segmentation_data.bg_classes) + len(scene.segmentation_data.fg_classes)
    slow_fast = config.instance_loss_mode == "slow_fast"
    g = int(config.min_grid_dim)
    model = cl.TensorVMSplit([g, g, g], num_semantics_comps=(32, 32, 32), num_instance_comps=(32, 32, 32),
                             num_semantic_classes=total_classes,
                             dim_feature_instance=2 * config.max_instances if slow_fast else config.max_instances,
                             output_mlp_semantics=torch.nn.Identity() if config.semantic_weight_mode != "softmax" else torch.nn.Softmax(dim=-1),
                             use_semantic_mlp=config.use_mlp_for_semantics, use_instance_mlp=config.use_mlp_for_instances,
                             use_distilled_features_semantic=config.use_distilled_features_semantic,
                             use_distilled_features_instance=config.use_distilled_features_instance,
                             pe_sem=config.pe_sem, pe_ins=config.pe_ins, slow_fast_mode=slow_fast, use_proj=config.use_proj, device=device)
    renderer = cl.TensoRFRenderer(scene.scene_bounds, [g, g, g], semantic_weight_mode=config.semantic_weight_mode).to(device)
    renderer.load_state_dict(strip_prefix(sd, "renderer"))
    for epoch in list(config.grid_upscale_epochs)[::-1]:
        if ckpt["epoch"] >= epoch:
            model.upsample_volume_grid(renderer.grid_dim.tolist())
            break
    renderer.update_step_size(renderer.grid_dim)            # refresh host scalars from the restored buffers
    msd = strip_prefix(sd, "model")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    if tuple(msd["density_plane.0"].shape) != shapes["density_plane.0"]:
        # checkpoint written after a bbox shrink: adopt the checkpoint's table shapes (RP:91-98 relies on
        # renderer.grid_dim carrying them)
        p0, l0 = msd["density_plane.0"], msd["density_line.0"]
        model.upsample_volume_grid([p0.shape[3], p0.shape[2], l0.shape[2]])
    model.load_state_dict(msd)
    return model, renderer, ckpt


def output_dirname(config, trajectory_name, test_only, use_dbscan, segmentwise):
    return Path("runs") / (f"{Path(config.dataset_root).stem}_{trajectory_name if not test_only else 'test'}_{Path(config.experiment)}"
                           f"{'_dbscan' if use_dbscan else ''}{'_seg' if segmentwise else ''}")


def glasbey(n):
    rng = np.random.default_rng(7)
    c = rng.uniform(0.15, 1.0, size=(n, 3))
    c[0] = 0
    return c


def render_panopli_checkpoint(config, trajectory_name, test_only=True, bandwidth=0.15, use_dbscan=False, segmentwise=False,
                              cached_centroids_path=None, device="cuda:0", use_silverman=False, cluster_size=500):
    out = output_dirname(config, trajectory_name, test_only, use_dbscan, segmentwise)
    out.mkdir(exist_ok=True, parents=True)
    # launched under torch.distributed.run: one process per GPU, every frame rendered as row-tiles (one per rank) and
    # assembled with one all-gather (inference.render_rays_sharded); clustering and file output happen on rank 0
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        device = f"cuda:{local}"
        dist.init_process_group(os.environ.get("CLIFT_DIST_BACKEND", "nccl"))
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = torch.device(device)
    scene = get_scene(config, "test", device)
    H, W = scene.image_dim
    model, renderer, _ = build_from_checkpoint(config, scene, device)
    renderer.update_step_ratio(renderer.step_ratio * 0.5)                                    # RP:104
    fg = scene.segmentation_data.fg_classes
    rgbs, sems, depths, inst_feats, thing_feats, slow_feats = [], [], [], [], [], []
    # RP:67-72: the test split, or the predefined trajectory trajectories/<trajectory_name>.pkl (frames named by index, all
    # with the intrinsics of frame 0)
    if test_only:
        frames = ((scene.all_frame_names[i], scene.rays_for(i), scene.intrinsics[i]) for i in scene.val_indices)
    else:
        frames = ((n, r, scene.intrinsics[0]) for n, r in scene.trajectory_set(trajectory_name))
    names = []
    with torch.no_grad():
        for name, rays, K_frame in frames:
            names.append(name)
            p_rgb, p_sem, p_inst, p_dist = inf.render_rays_sharded(model, renderer, rays, int(config.chunk), scene.white_bg)
            depths.append(inf.distance_to_depth(K_frame, p_dist.view(H, W)))
            if config.use_delta:
                p_inst = p_inst + (rays[:, 0:3] + p_dist[:, None] * rays[:, 3:6])
            if model.slow_fast_mode:
                slow_feats.append(p_inst[:, config.max_instances:])
                p_inst = p_inst[:, :config.max_instances]
            inst_feats.append(p_inst)
            rgbs.append(p_rgb)
            sems.append(p_sem)
            thing_feats.append(inf.create_instances_from_semantics(p_inst, p_sem, fg))
    if rank != 0:
        return out
    np.save(out / "instance_features.npy", torch.cat(inst_feats, 0).cpu().numpy())
    all_thing = torch.cat(thing_feats, 0).cpu().numpy()
    np.save(out / "thing_features.npy", all_thing)
    if model.slow_fast_mode:
        np.save(out / "slow_features.npy", torch.cat(slow_feats, 0).cpu().numpy())
    if cached_centroids_path is not None:
        with open(cached_centroids_path, "rb") as f:
            cents = pickle.load(f)
        insts = inf.assign_clusters(all_thing, sems, cents, device, num_images=len(rgbs))
    else:
        if not segmentwise:
            insts, _ = inf.cluster(all_thing, bandwidth, device, num_images=len(rgbs), use_silverman=use_silverman, use_dbscan=use_dbscan,
                                   cluster_size=cluster_size)
        else:
            insts, _ = inf.cluster_segmentwise(all_thing, sems, bandwidth, device, num_images=len(rgbs), use_silverman=use_silverman,
                                               use_dbscan=use_dbscan, cluster_size=cluster_size)
    for d in ("vis_semantics_and_surrogate", "pred_semantics", "pred_surrogateid"):
        (out / d).mkdir(exist_ok=True)
    for j, frame_name in enumerate(names):
        name = f"{frame_name}.png"
        sem_id = sems[j].argmax(dim=1).reshape(H, W).cpu().numpy()
        sur_id = insts[j].argmax(dim=1).reshape(H, W).cpu().numpy()
        Image.fromarray(sem_id.astype(np.uint8)).save(out / "pred_semantics" / name)
        Image.fromarray(sur_id.astype(np.uint16)).save(out / "pred_surrogateid" / name)
        pal = glasbey(int(max(sur_id.max(), sem_id.max())) + 2)
        d = depths[j].reshape(H, W).cpu().numpy()
        d = (d - d.min()) / max(float(np.ptp(d)), 1e-8)
        vis = np.concatenate([rgbs[j].reshape(H, W, 3).cpu().numpy(), pal[sem_id], pal[sur_id], np.repeat(d[..., None], 3, -1)], 1)
        Image.fromarray((vis.clip(0, 1) * 255).astype(np.uint8)).save(out / "vis_semantics_and_surrogate" / name)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt_path", type=str, required=True)
    ap.add_argument("--render_trajectory", action="store_true")
    ap.add_argument("--bandwidth", type=float, default=0.15, required=False)
    ap.add_argument("--cluster_size", type=int, default=500, required=False, help="min_cluster_size for HDBSCAN")
    ap.add_argument("--use_dbscan", action="store_true",
                    help="HDBSCAN clustering (reference RP:236-255): the hdbscan package when it is installed, otherwise "
                         "sklearn.cluster.HDBSCAN (the same algorithm; scikit-learn >= 1.3)")
    ap.add_argument("--segmentwise", action="store_true")
    ap.add_argument("--subsample", type=int, default=1, required=False)
    ap.add_argument("--use_silverman", action="store_true")
    ap.add_argument("--cached_centroids_path", type=str, required=False)
    ap.add_argument("--image_dim", type=int, nargs=2, default=[256, 384], help="reference hard-codes [256, 384] (RP:450)")
    args = ap.parse_args()
    cfg = load_run_config(Path(args.ckpt_path).parents[1] / "config.yaml")
    cfg.resume = args.ckpt_path
    cfg.subsample_frames = args.subsample
    cfg.image_dim = list(args.image_dim)
    print(render_panopli_checkpoint(cfg, "trajectory_blender", test_only=not args.render_trajectory, bandwidth=args.bandwidth,
                                    use_dbscan=args.use_dbscan, segmentwise=args.segmentwise,
                                    cached_centroids_path=args.cached_centroids_path, use_silverman=args.use_silverman,
                                    cluster_size=args.cluster_size))
