#!/usr/bin/env python3
"""Golden G22: one full SHORT TRAINING SCHEDULE run by the REFERENCE itself (build container only, CPU, several minutes per seed).

    python tests/golden/make_schedule_golden.py [seed ...]          # default seeds 0 1 2; writes g22_short_schedule.json

For every seed: the reference's ``TensoRFTrainer`` (trainer/train_panopli_tensorf.py:38-470, constructed by its own ``__init__`` on the
synthetic Messy-Rooms-layout scene of tools/make_synthetic_mos.py, datasets = the reference's MOSDataset / InconsistentMOSSingleDataset)
is driven through the loop Lightning's ``trainer.fit`` would run -- ``on_train_epoch_start``, ``training_step`` over the combined loaders
(pixel loader exhausted once per epoch, the instance loader cycling), schedulers at the last batch, ``validation_step`` over the validation
views after every epoch -- then its checkpoint is rendered by the reference's own ``inference/render_panopli.py`` (test views, MeanShift
clustering) and scored by the reference's scene evaluators (``calculate_panoptic_quality_folders_MOS``: PQ_scene).  Recorded per seed: the
last validation table (PSNR, mIoU, PQ ...), mIoU_scene / PQ_scene / SQ / RQ; over the seeds: mean and spread.  The product's train / render /
evaluate CLIs run the same schedule on the GPU in tests/test_gpu_round6.py and must land within max(0.1, the reference's own spread).

Third-party packages absent here are replaced as in make_golden.py (inert stand-ins; ``eff_distloss`` restated: the dist-reg term is UNPINNED);
what Lightning would supply is a minimal base class.  Visualisation calls are stubbed (no arithmetic of the path)."""
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as MG                       # noqa: E402  (puts REPO and the reference on sys.path)

REF = MG.REF
SCHEDULE = dict(experiment="contrastive_lift_MOS", image_dim=64, min_grid_dim=32, max_grid_dim=64, max_epoch=6, batch_size=256, chunk=2048,
                max_depth=3, max_rays_instances=512, decay_step=[4, 5], n_frames=40, scene_seed=0, bandwidth=0.15,
                infer_dim=240)      # the test views are rendered at 240 x 240: the reference's clustering draws 50 000 thing pixels without replacement (RP:213-214)


class FakeLightningModule(torch.nn.Module):
    """What TensoRFTrainer uses of pl.LightningModule."""
    device = torch.device("cpu")
    current_epoch = 0
    global_step = 0

    def save_hyperparameters(self, *a, **k):
        pass

    def optimizers(self):
        return self._opts

    def lr_schedulers(self):
        return self._scheds

    def manual_backward(self, loss):
        loss.backward()

    def log(self, name, value, **k):
        self._logged[name] = float(value)


def _infer(seed, cfg, ckpt, scene_dir, table, gstep, t0, final):
    """The reference's inference on its own checkpoint: test views, MeanShift clustering of the rendered instance features, scene evaluators."""
    if True:
        import inference.render_panopli as RP
        import dataset.preprocessing.preprocess_scannet as PS
        import pathlib
        H = W = SCHEDULE["infer_dim"]
        cfg.image_dim = [H, W]
        class _TorchOnCpu:                          # environment shim: render_panopli.py:48 hard-codes torch.device("cuda:0"); no GPU in this container
            def __getattr__(self, n):
                return getattr(torch, n)

            def device(self, *a, **k):
                return torch.device("cpu")
        RP.torch = _TorchOnCpu()
        RP.visualize_panoptic_outputs = lambda *a, **k: torch.zeros(5, 3, H, W)
        RP.make_grid = lambda stack, **k: torch.zeros(3, H, W)
        cfg.resume = ckpt
        cfg.subsample_frames = 1
        np.random.seed(seed)
        real_cuda, real_to = torch.Tensor.cuda, None
        torch.Tensor.cuda = lambda self, *a, **k: self
        _read = PS.read_and_resize_labels
        PS.read_and_resize_labels = lambda path, size: _read(path, size).astype(np.int32)      # (environment shim as in make_golden.g16: Pillow's uint16)
        real_dev = torch.cuda.is_available
        try:
            scene = None
            try:
                with (contextlib.nullcontext() if os.environ.get("G22_VERBOSE") else MG.quiet()):
                    RP.render_panopli_checkpoint(cfg, "trajectory_blender", test_only=True, bandwidth=SCHEDULE["bandwidth"])
            except ValueError as e:
                # RP:213-214 draws 50 000 thing pixels WITHOUT replacement: a field that has not separated the objects yet predicts fewer and the
                # reference's own script stops here -- such a run has no PQ_scene on the reference's side
                if "larger sample than population" not in str(e):
                    raise
                print(f"seed {seed}: the reference's render_panopli.py raised ({e}): no scene metrics for this run", flush=True)
                return dict(seed=seed, steps=gstep, seconds=time.time() - t0, val=table, scene=None, scene_note="render_panopli.py:214 raised: fewer than 50 000 predicted thing pixels", **final)
            out_dir = RP.output_dirname(cfg, "trajectory_blender", True, False, False)
            with MG.quiet():
                iou = PS.calculate_iou_folders_MOS(pathlib.Path(out_dir, "pred_semantics"), pathlib.Path(scene_dir) / "semantic", (H, W))
                pq, sq, rq = PS.calculate_panoptic_quality_folders_MOS(pathlib.Path(out_dir, "pred_semantics"), pathlib.Path(out_dir, "pred_surrogateid"),
                                                                       pathlib.Path(scene_dir) / "semantic", pathlib.Path(scene_dir) / "instance", (H, W))
        finally:
            torch.Tensor.cuda = real_cuda
            PS.read_and_resize_labels = _read
        res = dict(seed=seed, steps=gstep, seconds=time.time() - t0, val=table, scene=dict(iou=float(iou), pq=float(pq), sq=float(sq), rq=float(rq)), **final)
        print(f"seed {seed}: scene mIoU {iou:.4f} PQ_scene {pq:.4f} SQ {sq:.4f} RQ {rq:.4f}", flush=True)
        return res


def run_seed(seed, threads, reuse=None):
    """``reuse``: a work directory of an earlier call whose training finished (train.json + checkpoint present): only the inference stage runs."""
    import yaml
    from contrastive_lift_amd.config import load_config, save_config
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_mos as gen
    tmp = reuse or os.path.join(os.environ.get("G22_WORK", "/tmp/g22_work"), f"seed{seed}")
    os.makedirs(tmp, exist_ok=True)
    if not os.path.exists(os.path.join(tmp, "resources")):
        os.symlink(os.path.join(REF, "resources"), os.path.join(tmp, "resources"))   # the datasets read resources/*.csv relative to the cwd
    scene_dir = gen.make_scene(os.path.join(tmp, "data", "synth_scene"), n_frames=SCHEDULE["n_frames"], size=SCHEDULE["image_dim"], seed=SCHEDULE["scene_seed"])
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        cfg = load_config(os.path.join(REF, "config"), overrides=[f"+experiment={SCHEDULE['experiment']}", f"dataset_root={scene_dir}",
                          f"image_dim={SCHEDULE['image_dim']}", f"min_grid_dim={SCHEDULE['min_grid_dim']}", f"max_grid_dim={SCHEDULE['max_grid_dim']}",
                          f"max_epoch={SCHEDULE['max_epoch']}", f"batch_size={SCHEDULE['batch_size']}", f"chunk={SCHEDULE['chunk']}",
                          f"max_depth={SCHEDULE['max_depth']}", f"max_rays_instances={SCHEDULE['max_rays_instances']}",
                          f"decay_step={SCHEDULE['decay_step']}", f"seed={seed}", "num_workers=0", "logger=none"])
        cfg.image_dim = [cfg.image_dim, cfg.image_dim]
        cfg.experiment = f"g22_seed{seed}"
        os.makedirs(f"runs/{cfg.experiment}/checkpoints", exist_ok=True)
        done = os.path.join(tmp, "train.json")
        if os.path.exists(done):
            st = json.load(open(done))
            table, gstep, ckpt, t0 = st["table"], st["gstep"], st["ckpt"], time.time() - st["seconds"]
            state = torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"]
            final = dict(grid=[int(x) for x in state["renderer.grid_dim"].tolist()], aabb=[[float(x) for x in r] for r in state["renderer.bbox_aabb"].tolist()])
            return _infer(seed, cfg, ckpt, scene_dir, table, gstep, t0, final)
        torch.manual_seed(seed)                    # seed_everything(config.seed), trainer/__init__.py:73
        np.random.seed(seed)
        __import__("random").seed(seed)
        import trainer.train_panopli_tensorf as T
        t0 = time.time()
        with MG.quiet():
            model = T.TensoRFTrainer(cfg)
        model._logged = {}
        model._opts, model._scheds = model.configure_optimizers()

        def setup_optimizers(_tr):                  # Lightning: Strategy.setup_optimizers -> _init_optimizers_and_lr_schedulers
            model._opts, model._scheds = model.configure_optimizers()
        model.trainer = types.SimpleNamespace(is_last_batch=False, current_epoch=0, strategy=types.SimpleNamespace(setup_optimizers=setup_optimizers))
        with MG.quiet():
            loaders = model.train_dataloader()
        val_loader = model.val_dataloader()
        n_steps = max(len(l) for l in loaders.values())
        table, gstep, first_epoch = {}, 0, 0
        state_file = os.path.join(tmp, "epoch_state.pt")
        if os.path.exists(state_file):          # an interrupted generation continues at the last finished epoch (states of everything that carries over)
            st = torch.load(state_file, map_location="cpu", weights_only=False)
            grid = [int(x) for x in st["renderer"]["grid_dim"].tolist()]
            with MG.quiet():
                if grid != [int(x) for x in model.renderer.grid_dim.tolist()]:
                    model.model.upsample_volume_grid(grid)
                model.renderer.bbox_aabb.data = st["renderer"]["bbox_aabb"].clone()
                model.renderer.update_step_size(torch.tensor(grid))
            model.model.load_state_dict(st["model"], strict=True)
            cfg.weight_decay = st["weight_decay"]
            model._opts, model._scheds = model.configure_optimizers()
            for o, s_ in zip(model._opts, st["opts"]):
                o.load_state_dict(s_)
            for o, s_ in zip(model._scheds, st["scheds"]):
                o.load_state_dict(s_)
            torch.set_rng_state(st["rng_torch"]); np.random.set_state(st["rng_numpy"]); __import__("random").setstate(st["rng_python"])
            table, gstep, first_epoch, t0 = st["table"], st["gstep"], st["epoch"] + 1, time.time() - st["seconds"]
            print(f"seed {seed}: resuming after epoch {st['epoch']}", flush=True)
        for epoch in range(first_epoch, int(cfg.max_epoch)):
            model.current_epoch = model.trainer.current_epoch = epoch
            with MG.quiet():
                model.on_train_epoch_start()
            its = {k: iter(l) for k, l in loaders.items()}
            for i in range(n_steps):                # CombinedLoader "max_size_cycle": the shorter loaders restart
                batch = {}
                for k in its:
                    try:
                        batch[k] = next(its[k])
                    except StopIteration:
                        its[k] = iter(loaders[k])
                        batch[k] = next(its[k])
                model.trainer.is_last_batch = i == n_steps - 1
                with MG.quiet():
                    model.training_step(batch, i)
                gstep += 1
                model.global_step = gstep
            model.validation_step_outputs = []
            with torch.no_grad(), MG.quiet():
                for j, vb in enumerate(val_loader):
                    model.validation_step(vb, j)
            rows = model.validation_step_outputs
            table = {k: float(np.mean([r[k] for r in rows])) for k in rows[0]}
            print(f"seed {seed} epoch {epoch}: {n_steps} steps, train psnr {model._logged.get('train/psnr', float('nan')):.2f}, val psnr {table['psnr']:.3f} "
                  f"iou {table['iou']:.3f} pq {table['pq']:.3f} grid {model.renderer.grid_dim.tolist()} S {model.renderer.n_samples} "
                  f"({time.time() - t0:.0f} s)", flush=True)
            torch.save(dict(epoch=epoch, gstep=gstep, table=table, seconds=time.time() - t0, weight_decay=cfg.weight_decay,
                            model={k: v.detach().clone() for k, v in model.model.state_dict().items()},
                            renderer={k: v.detach().clone() for k, v in model.renderer.state_dict().items()},
                            opts=[o.state_dict() for o in model._opts], scheds=[s_.state_dict() for s_ in model._scheds],
                            rng_torch=torch.get_rng_state(), rng_numpy=np.random.get_state(), rng_python=__import__("random").getstate()), state_file + ".tmp")
            os.replace(state_file + ".tmp", state_file)
        ckpt = f"runs/{cfg.experiment}/checkpoints/epoch={int(cfg.max_epoch) - 1}-step={gstep}.ckpt"
        torch.save({"state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()}, "epoch": int(cfg.max_epoch) - 1, "global_step": gstep}, ckpt)
        save_config(cfg, f"runs/{cfg.experiment}/config.yaml")
        json.dump(dict(table=table, gstep=gstep, ckpt=os.path.abspath(ckpt), seconds=time.time() - t0), open(done, "w"))
        final = dict(grid=[int(x) for x in model.renderer.grid_dim.tolist()], aabb=[[float(x) for x in r] for r in model.renderer.bbox_aabb.tolist()])
        return _infer(seed, cfg, ckpt, scene_dir, table, gstep, t0, final)
    finally:
        os.chdir(cwd)


def main():
    seeds = [int(x) for x in sys.argv[1:]] or [0, 1, 2]
    if not os.path.isdir(REF):
        sys.exit(f"reference not found at {REF}")
    threads = int(os.environ.get("G22_THREADS", "4"))
    torch.set_num_threads(threads)
    MG.install_stand_ins()
    MG.install_quaternion()
    torch.cuda.device_count = lambda: 1          # environment shim: dataset/base.py:88 divides by the device count (no GPU in this container)
    sys.modules["pytorch_lightning"].LightningModule = FakeLightningModule
    sys.modules["hdbscan"].HDBSCAN = MG._Inert
    reuse = dict(kv.split("=") for kv in os.environ.get("G22_REUSE", "").split(",") if kv)      # seed=workdir: skip the training of that seed
    work = os.environ.get("G22_WORK", "/tmp/g22_work")
    os.makedirs(work, exist_ok=True)
    runs = []
    for s_ in seeds:                                  # one result file per seed: an interrupted generation resumes where it stopped
        f = os.path.join(work, f"seed{s_}.json")
        if not os.path.exists(f):
            json.dump(run_seed(s_, threads, reuse.get(str(s_))), open(f, "w"))
        runs.append(json.load(open(f)))
    if os.environ.get("G22_NO_AGGREGATE"):
        return
    agg = {}
    for key, get in (("val_psnr", lambda r: r["val"]["psnr"]), ("val_iou", lambda r: r["val"]["iou"]), ("val_pq", lambda r: r["val"]["pq"]),
                     ("scene_iou", lambda r: r["scene"]["iou"]), ("pq_scene", lambda r: r["scene"]["pq"])):
        v = np.array([get(r) for r in runs if not (key in ("scene_iou", "pq_scene") and r["scene"] is None)])      # (scene metrics: the runs the reference's script finished)
        agg[key] = dict(mean=float(v.mean()), median=float(np.median(v)), min=float(v.min()), max=float(v.max()), spread=float(v.max() - v.min()),
                        values=[float(x) for x in v])
    path = os.path.join(HERE, "g22_short_schedule.json")
    json.dump(dict(schedule=SCHEDULE, unpinned_note="dist-reg term via the restated eff_distloss; Lightning's fit loop emulated (see the docstring)",
                   runs=runs, summary=agg), open(path, "w"), indent=1)
    print("wrote", path, json.dumps(agg))


if __name__ == "__main__":
    main()
