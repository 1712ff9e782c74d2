#!/usr/bin/env python3
"""Drop-in for reference trainer/train_panopli_tensorf.py (T:473-489): same config tree and Hydra-style overrides, same
run directory (runs/<experiment>/config.yaml, runs/<experiment>/checkpoints/*.ckpt in the Lightning layout).

    python trainer/train_panopli_tensorf.py +experiment=contrastive_lift_MOS dataset_root=data/mos/<scene> [key=value ...]
    python -m torch.distributed.run --nproc-per-node 8 trainer/train_panopli_tensorf.py ...     # one process per GPU (RCCL)

Epoch schedule of the reference (T:446-459): dist-reg ramp, bbox shrink at ``bbox_aabb_reset_epochs``, log-spaced grid
upsampling at ``grid_upscale_epochs`` (after which the optimizers are rebuilt and weight decay drops to 0), LR decay at
``decay_step``; instance pass from ``instance_optimization_epoch + late_semantic_optimization`` on (T:46).
"""
import datetime
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import contrastive_lift_amd as cl                                        # noqa: E402
from contrastive_lift_amd.config import load_config, save_config          # noqa: E402
from contrastive_lift_amd.data import get_scene                            # noqa: E402
from contrastive_lift_amd.trainer import HotPathTrainer                   # noqa: E402


def experiment_name(config):
    """trainer/__init__.py:48-58 without the random-name dependency."""
    if config.get("resume"):
        return Path(config.resume).parents[1].name
    if os.environ.get("experiment"):
        return os.environ["experiment"]
    return f"{datetime.datetime.now().strftime('%m%d%H%M')}_MOS_{Path(config.dataset_root).stem}_{config.experiment}"


def resume_from(path, cfg, tr, model, renderer, dev):
    """Continue a run from one of its checkpoints -- or from a checkpoint the REFERENCE wrote (same layout, torch-layout optimizer
    states) -- reference: trainer.fit(ckpt_path=config.resume) + on_load_checkpoint, T:461-470.  ``HotPathTrainer.on_load_checkpoint``
    restores tables at the checkpoint's resolution, every weight, the renderer buffers, both Adam states and the scheduler position.
    Returns (first epoch to run, global step, whether that epoch was already in progress)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    tr.config = cfg
    extra = tr.on_load_checkpoint(ckpt)
    epoch = int(ckpt["epoch"])
    # a Lightning checkpoint carries no "epoch finished" flag of its own: ModelCheckpoint(every_n_train_steps) writes in the middle of an
    # epoch, so a checkpoint without this repo's record is taken as mid-epoch (the epoch's hook already ran before it was written)
    complete = bool(extra.get("epoch_complete", False))
    first = epoch + 1 if complete else epoch
    # Only rank 0 writes checkpoints, so the RNG record (CPU / device generators, pixel-batch generator) is ITS streams: restoring it on
    # every rank would make all ranks draw the same pixel batches and jitter from here on (the all-reduced gradient would be that of one
    # batch).  The other ranks re-seed from (seed, rank, global step): decorrelated from rank 0 and from each other, reproducible.
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        tr.load_rng_state(extra.get("rng"))
    else:
        seed = int(cfg.seed if cfg.seed is not None else 0)
        mix = (seed * 9973 + rank) * 1000003 + int(ckpt.get("global_step", 0)) + 1
        torch.manual_seed(mix)
        if dev.type == "cuda":
            torch.cuda.manual_seed(mix)
        g = getattr(tr, "pixel_generator", None)
        if g is not None:
            g.manual_seed(mix)
    return first, int(ckpt.get("global_step", 0)), (not complete)


VAL_KEYS = ("loss_rgb", "loss_sem", "psnr", "iou", "pq", "sq", "rq", "rs_iou", "rs_pq", "rs_sq", "rs_rq")


def validation_epoch(tr, val, limit=None):
    """validation_step over the validation views + the table of on_validation_epoch_end (T:356-408): the mean of every metric over the
    views, printed with tabulate like the reference.  Returns {metric: mean}."""
    rows = []
    idxs = list(val.val_indices)
    if limit:
        idxs = idxs[:max(1, int(round(len(idxs) * float(limit))))] if float(limit) <= 1 else idxs[:int(limit)]
    for i in idxs:
        batch = dict(rays=val.rays_for(i), **val.load_targets(i), **val.load_rs_targets(i))
        rows.append(tr.validation_step(batch, val.things_filtered, val.stuff_filtered, val.faulty_classes))
    if not rows:
        return {}
    means = {k: float(np.mean([r[k] for r in rows])) for k in VAL_KEYS}
    try:
        from tabulate import tabulate
        print(tabulate([VAL_KEYS, tuple(means[k] for k in VAL_KEYS)], headers="firstrow", tablefmt="fancy_grid"), flush=True)
    except ImportError:
        print(" ".join(f"{k} {means[k]:.4f}" for k in VAL_KEYS), flush=True)
    return means


def main(argv):
    config_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config")
    for a in list(argv):
        if a.startswith("--config-dir="):
            config_dir = a.split("=", 1)[1]
            argv.remove(a)
    cfg = load_config(config_dir, overrides=argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise ValueError("No GPU found")                                   # trainer/__init__.py:121-122
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    if isinstance(cfg.image_dim, int):
        cfg.image_dim = [cfg.image_dim, cfg.image_dim]
    cfg.experiment = experiment_name(cfg)
    if getattr(cfg, "skip_discarded_instance_heads", None) is None:
        # the reference's main pass evaluates the instance heads and throws the result away (T:155: `_`); same training either way
        cfg.skip_discarded_instance_heads = True
    if getattr(cfg, "host_rng", None) is None:
        cfg.host_rng = True                                                # jitter from torch's CPU generator, like the reference's renderer (R:808-810)
    seed = cfg.seed if cfg.seed is not None else 0
    torch.manual_seed(seed)                                                # identical initial weights on every rank
    cfg.instance_optimization_epoch = cfg.instance_optimization_epoch + cfg.late_semantic_optimization     # T:46
    cfg.segment_optimization_epoch = cfg.segment_optimization_epoch + cfg.late_semantic_optimization       # T:47
    scene = get_scene(cfg, "train", dev)
    scene.build_train_tables(instance_images=False)
    # the per-image instance ray sets come from their own dataset object, always at (128, 128) (dataset/__init__.py:57,66)
    inst_scene = get_scene(cfg, "train", dev, image_dim=(128, 128))
    inst_scene.build_instance_tables()
    seg_scene = None
    if cfg.segment_grouping_mode != "none" and int(cfg.segment_optimization_epoch) < int(cfg.max_epoch):
        seg_scene = get_scene(cfg, "train", dev, image_dim=(128, 128))         # get_segment_dataset: always (128, 128) (dataset/__init__.py:70,78)
        seg_scene.build_segment_tables()
    val = get_scene(cfg, "val", dev)
    total_classes = len(scene.segmentation_data.bg_classes) + len(scene.segmentation_data.fg_classes)      # T:51
    slow_fast = cfg.instance_loss_mode == "slow_fast"
    g = int(cfg.min_grid_dim)
    model = cl.TensorVMSplit([g, g, g], num_semantics_comps=(32, 32, 32), num_instance_comps=(32, 32, 32), num_semantic_classes=total_classes,
                             dim_feature_instance=2 * cfg.max_instances if slow_fast else cfg.max_instances,
                             output_mlp_semantics=torch.nn.Identity() if cfg.semantic_weight_mode != "softmax" else torch.nn.Softmax(dim=-1),
                             use_semantic_mlp=cfg.use_mlp_for_semantics, use_instance_mlp=cfg.use_mlp_for_instances,
                             pe_sem=cfg.pe_sem, pe_ins=cfg.pe_ins, slow_fast_mode=slow_fast, use_proj=cfg.use_proj, device=dev)
    renderer = cl.TensoRFRenderer(scene.scene_bounds, [g, g, g], semantic_weight_mode=cfg.semantic_weight_mode,
                                  stop_semantic_grad=cfg.stop_semantic_grad).to(dev)
    cw = cl.get_semantic_weights(getattr(cfg, "reweight_fg", False), list(scene.segmentation_data.fg_classes), total_classes)   # T:69
    cw[0] = cfg.weight_class_0                                             # T:70
    tr = HotPathTrainer(model, renderer, cfg, class_weights=cw, current_epoch=0)
    run_dir = Path("runs") / cfg.experiment
    if rank == 0:
        (run_dir / "checkpoints").mkdir(parents=True, exist_ok=True)
        save_config(cfg, str(run_dir / "config.yaml"))
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 9973 + rank)                                    # every rank draws its own pixels (DistributedSampler role)
    tr.pixel_generator = gen                                               # (its state travels in the checkpoint)
    # Lightning DDP (trainer/__init__.py:97): every rank takes a FULL batch_size batch and the DistributedSampler divides the steps
    # of an epoch by the world size -- the LR / EMA / epoch schedules of the configs are tuned for that
    per_rank = int(cfg.batch_size)
    steps_per_epoch = int(cfg.get("steps_per_epoch") or max(1, scene.tables["rays"].shape[0] // (int(cfg.batch_size) * world)))
    gstep, start_epoch, resumed_mid_epoch = 0, 0, False
    if cfg.get("resume"):
        start_epoch, gstep, resumed_mid_epoch = resume_from(str(cfg.resume), cfg, tr, model, renderer, dev)
        if rank == 0:
            print(f"resumed {cfg.resume}: continuing at epoch {start_epoch} (global step {gstep}), grid {renderer.grid_dim.tolist()}", flush=True)
    # the interpreter's first full garbage collection walks everything the imports and the set-up created (35 - 60 ms, i.e. ten training steps);
    # it is taken here, once, and the set-up's objects are frozen out of later collections (bench.py does the same before its warm-up)
    import gc
    gc.collect()
    gc.freeze()
    val_every = max(1, int(cfg.get("val_check_interval") or 1))              # trainer/__init__.py:104-105: every n-th epoch (fractions: every epoch)
    last_val = {}
    for epoch in range(start_epoch, int(cfg.max_epoch)):
        tr.current_epoch = epoch
        # T:446-457 (golden G21): ramp, shrink, upsample + weight_decay 0 + optimizer / scheduler rebuild.  A checkpoint written in the
        # middle of an epoch already holds that epoch's shrunk / upsampled tables: only the ramp then
        tr.on_train_epoch_start(maintenance=not (resumed_mid_epoch and epoch == start_epoch))
        order = torch.randperm(max(1, len(inst_scene.instance_images)), generator=torch.Generator().manual_seed(seed * 31 + epoch)).tolist()   # DataLoader(shuffle=True), T:436
        pixels = scene.epoch_order(seed, epoch, rank, world)               # DataLoader(train_set, shuffle=True, drop_last=True), T:434: every pixel once per epoch
        # a checkpoint written in the middle of an epoch continues at ITS batch: the epoch's pixel order is a function of (seed, epoch), so the
        # remaining batches are exactly the ones the interrupted run would have drawn (Lightning's fit loop restores its batch progress likewise)
        first_it = (gstep % steps_per_epoch) if (resumed_mid_epoch and epoch == start_epoch) else 0
        for it in range(first_it, steps_per_epoch):
            batch = {0: scene.pixel_batch_at(pixels, it, per_rank)}
            if epoch >= cfg.instance_optimization_epoch and inst_scene.instance_images:
                batch[1] = inst_scene.instance_batch(int(cfg.max_rays_instances), order[(it * world + rank) % len(order)])
            if seg_scene is not None and epoch >= cfg.segment_optimization_epoch:                                   # T:458-459
                sb = seg_scene.segment_batch(int(cfg.batch_size_segments), int(cfg.max_rays_segments), gstep * world + rank)
                if sb is not None:
                    batch[2] = sb
            tr.training_step(batch)
            gstep += 1
            if rank == 0 and gstep % int(cfg.save_every_n_train_steps) == 0:
                tr.save_checkpoint(str(run_dir / "checkpoints" / f"epoch={epoch}-step={gstep}.ckpt"), gstep, epoch_complete=False)
            if rank == 0 and (it % 50 == 0 or it == steps_per_epoch - 1):
                l = tr.losses.tolist()
                print(f"epoch {epoch} it {it}/{steps_per_epoch} loss_rgb {l[0]:.5f} (psnr {-10 * math.log10(max(l[0], 1e-12)):.2f}) "
                      f"loss_sem {l[1]:.4f} tv {l[2]:.5f} clustering {l[3]:.4f}"
                      + (f" segment {float(tr.loss_segment[0]):.4f}" if 2 in batch else "")
                      + (f" overflow_steps {tr.overflow_steps}" if tr.nosync else "")
                      + f" lr x{tr.opt_main.lr_scale:g} S={renderer.n_samples} grid={renderer.grid_dim.tolist()}", flush=True)
        tr.scheduler_step()                                                # T:226-228: both MultiStepLR schedulers, at the epoch's last batch
        if rank == 0:
            tr.save_checkpoint(str(run_dir / "checkpoints" / f"epoch={epoch}-step={gstep}.ckpt"), gstep, epoch_complete=True)
            if (epoch + 1) % val_every == 0:                               # the reference's validation table (T:356-408)
                last_val = validation_epoch(tr, val, cfg.get("val_check_percent"))
    main.last_validation = last_val
    if world > 1:
        dist.destroy_process_group()
    return str(run_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
