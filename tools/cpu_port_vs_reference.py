#!/usr/bin/env python3
"""Build container only: time the oracle's CpuTrainer (what bench.py reports as cpu_baseline, kind "port") and the REFERENCE's own
trainer class (imported from /root/reference with the stand-in modules of tests/golden/make_golden.py) side by side on the same
training_step -- the equivalence evidence BASELINE.md section 3 promises for the "port" baseline.

    python tools/cpu_port_vs_reference.py [--rays 4096] [--inst-rays 1024] [--grid 128] [--classes 22] [--steps 3] [--threads N]

Both run: main pass (chunk 2048, MSE + TV + confidence-weighted CE + dist-reg, backward, Adam) + slow-fast instance pass, fp32, the
bench workload's shapes (C = 22, E = 3, grid 128^3 => S = 440) on a blob scene, same weights, same rays.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--inst-rays", type=int, default=1024)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--classes", type=int, default=22)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    import make_golden as mg                      # inserts /root/reference into sys.path
    if not os.path.isdir(mg.REF):
        sys.exit("needs the reference tree (build container only)")
    mg.install_stand_ins()
    torch.set_num_threads(a.threads)
    from oracle import params as op, render as orender, rays as orays
    from oracle.train_step import CpuTrainer
    import trainer.train_panopli_tensorf as T
    from model.loss.loss import TVLoss

    res, C, E = (a.grid,) * 3, a.classes, 3
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    P = op.add_blob(op.make_params(0, res, C, E), res, amplitude=3.0, sigma_g=0.35)
    rng = np.random.default_rng(1)
    img = 128
    K = torch.tensor([[img * 1.25, 0, img / 2], [0, img * 1.25, img / 2], [0, 0, 1]])
    pool = torch.cat([orays.ray_table(img, img, K, mg.look_at(e)) for e in ((0.0, 0.0, -0.9), (0.7, -0.35, 0.45), (-0.55, 0.5, 0.5))], 0)
    B, Bi = a.rays, a.inst_rays
    rays = pool[torch.from_numpy(rng.choice(pool.shape[0], B, replace=False))].contiguous()
    irays = pool[torch.from_numpy(rng.choice(img * img, Bi, replace=False))].contiguous()
    rgbs = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32))
    probs = torch.softmax(torch.from_numpy(rng.standard_normal((B, C)).astype(np.float32)), -1)
    conf = torch.from_numpy(rng.uniform(0, 1, B).astype(np.float32))
    labels = torch.from_numpy(rng.integers(1, 26, Bi))
    iconf = torch.from_numpy(rng.uniform(0, 1, Bi).astype(np.float32))

    # ---- the port
    ct = CpuTrainer(P, orender.RenderCfg(aabb, res, density_shift=-3.0), chunk=2048, epoch=4)
    g = torch.Generator().manual_seed(2)

    def port_step():
        jit = torch.rand(B, generator=g)
        ct.main_pass(rays, rgbs, probs, conf, jit, [False] * ((B + 2047) // 2048))
        ct.instance_pass(irays, labels, iconf, torch.rand(Bi, generator=g))

    # ---- the reference's own trainer class on a shim (as tests/golden/make_golden.py:g12 does)
    cfg = types.SimpleNamespace(
        lr=5e-4, weight_decay=1e-8, decay_step=[9, 10], decay_gamma=0.5, warmup_epochs=0, chunk=2048, perturb=1.0,
        optimize_instance_only=False, lambda_rgb=1.0, lambda_semantics=0.1, lambda_feat=0.0, lambda_segment=0.0,
        lambda_tv_density=0.1, lambda_tv_appearance=0.01, lambda_tv_semantics=0.02, lambda_tv_instances=0.02,
        use_distilled_features_semantic=False, use_distilled_features_instance=False, feature_optimization_end_epoch=0,
        late_semantic_optimization=1, instance_optimization_epoch=3, segment_optimization_epoch=100, segment_grouping_mode="none",
        batch_size_segments=6, chunk_segment=16384, probabilistic_ce_mode="TTAConf", use_proj=False, max_instances=E)
    m = mg.build_reference_model(P, res, C, E, shift=-3.0)
    rr = mg.build_reference_renderer(aabb, res, "softmax")
    cw = torch.ones(C); cw[0] = 0.0

    class Shim:
        configure_optimizers = T.TensoRFTrainer.configure_optimizers
        forward = T.TensoRFTrainer.forward
        forward_instance = T.TensoRFTrainer.forward_instance
        training_step = T.TensoRFTrainer.training_step
        calculate_instance_clustering_loss = T.TensoRFTrainer.calculate_instance_clustering_loss
        ema_update_slownet = T.TensoRFTrainer.ema_update_slownet

        def __call__(self, *x): return self.forward(*x)
        def optimizers(self): return self._opts
        def lr_schedulers(self): return self._scheds
        def manual_backward(self, loss): loss.backward()
        def log(self, *x, **k): pass
    sh = Shim()
    sh.config, sh.model, sh.renderer = cfg, m, rr
    sh.train_set = types.SimpleNamespace(white_bg=False)
    sh.loss, sh.loss_feat, sh.tv_regularizer = torch.nn.MSELoss(reduction="mean"), torch.nn.L1Loss(reduction="mean"), TVLoss()
    sh.loss_semantics = torch.nn.CrossEntropyLoss(reduction="none", weight=cw)
    sh.instance_loss_mode, sh.use_DINO_style, sh.temperature, sh.use_delta = "slow_fast", True, 100.0, False
    sh.device, sh.current_epoch = torch.device("cpu"), 4
    sh.current_lambda_dist_reg = 0.005 * (1 - np.exp(-1.0))
    sh.trainer = types.SimpleNamespace(is_last_batch=False, current_epoch=4)
    sh._opts, sh._scheds = sh.configure_optimizers()
    sem = probs.argmax(-1)
    mask = torch.ones(B, dtype=torch.bool)

    def ref_step():
        batch = {0: dict(rays=rays, rgbs=rgbs.clone(), semantics=sem, probabilities=probs, confidences=conf.clone(), mask=mask, feats=torch.zeros(B, 1)),
                 1: dict(rays=[irays], instances=[labels], confidences=[iconf])}
        with mg.quiet():
            sh.training_step(batch, 0)

    def timed(fn):
        fn()                                       # warm-up
        ts = []
        for _ in range(a.steps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        ts.sort()
        return ts[len(ts) // 2]
    t_port, t_ref = timed(port_step), timed(ref_step)
    S = int(orender.RenderCfg(aabb, res).n_samples)
    n = (B + Bi) * S
    print(json.dumps({"threads": a.threads, "rays": B, "inst_rays": Bi, "grid": a.grid, "classes": C, "samples_per_ray": S,
                      "port_s_per_step": round(t_port, 3), "reference_s_per_step": round(t_ref, 3),
                      "port_ray_samples_per_s": round(n / t_port), "reference_ray_samples_per_s": round(n / t_ref),
                      "port_over_reference_time": round(t_port / t_ref, 3), "torch": torch.__version__}))


if __name__ == "__main__":
    main()
