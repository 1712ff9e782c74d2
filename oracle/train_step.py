"""Oracle: one training_step of the hot path on the CPU (reference trainer/train_panopli_tensorf.py:148-228):
main pass (chunked forward, MSE + TV + confidence-weighted CE + dist-reg, backward, Adam betas (0.9,0.99)) and
instance pass (forward_instance_feature, slow-fast loss after the EMA step, backward, Adam betas (0.9,0.999)).

Test infrastructure only (see oracle/__init__.py).  It is what ``bench.py`` times as ``cpu_baseline`` (kind
"port": same ATen ops, same shapes and chunking as the reference's CPU-PyTorch path) and what the GPU training
step is checked against in tests/.
"""
import math

import torch

from . import losses as olosses
from . import render as orender

GRID_KEYS = ("density_plane", "density_line", "appearance_plane", "appearance_line", "semantic_plane", "semantic_line")
INST_GRID_KEYS = ("instance_plane", "instance_line")


class CpuTrainer:
    def __init__(self, P, cfg, lr=5e-4, weight_decay=1e-8, lambda_rgb=1.0, lambda_semantics=0.1, lambda_dist_reg=0.005,
                 lambda_tv_density=0.1, lambda_tv_appearance=0.01, chunk=2048, epoch=4, class_weights=None, dino=True,
                 instance_loss_mode="slow_fast", temperature=100.0, use_delta=False, lambda_segment=1.2,
                 late_semantic_optimization=0, sce=None, lambda_tv_semantics=0.02, lambda_tv_instances=0.02, instance_optimization_epoch=0):
        self.P = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        self.cfg, self.chunk = cfg, chunk
        self.inst_mode, self.temperature, self.use_delta = instance_loss_mode, temperature, use_delta
        self.l_seg = lambda_segment
        self.sce = sce                                             # (ce_alpha, ce_beta) when config.use_symmetric_ce (T:74-77)
        self.l_rgb, self.l_sem, self.l_tvd, self.l_tva = lambda_rgb, lambda_semantics, lambda_tv_density, lambda_tv_appearance
        self.l_tvs, self.l_tvi = lambda_tv_semantics, lambda_tv_instances
        self.lambda_dist_reg, self.late_sem, self.inst_epoch = lambda_dist_reg, late_semantic_optimization, instance_optimization_epoch
        self.lr, self.weight_decay, self.dino = lr, weight_decay, dino
        self.decay_step, self.decay_gamma = (9, 10), 0.5
        self.set_epoch(epoch)
        self.configure_optimizers()
        C = self.P[[k for k in self.P if k.startswith("render_semantic_mlp.mlp.") and k.endswith(".bias")][-1]].shape[0]
        self.cw = torch.ones(C) if class_weights is None else class_weights
        if class_weights is None:
            self.cw[0] = 0.0

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.sem_on = epoch >= self.late_sem                       # T:175,198: no semantic term before that epoch
        self.inst_on = epoch >= self.inst_epoch
        self.l_dist = self.lambda_dist_reg * (1 - math.exp(-0.25 * epoch))                 # T:447

    def configure_optimizers(self):
        """T:98-103 + trainer/__init__.py:134-139: two Adams over the CURRENT parameter tensors, each with a MultiStepLR."""
        lr, weight_decay = self.lr, self.weight_decay
        grids = [v for k, v in self.P.items() if k.startswith(GRID_KEYS)]
        nets = [v for k, v in self.P.items() if not k.startswith(GRID_KEYS + INST_GRID_KEYS) and not k.startswith(("render_instance_mlp", "instance_basis_mat"))]
        self.main_params = grids + nets
        self.opt_main = torch.optim.Adam([{"params": grids, "lr": lr * 20}, {"params": nets, "lr": lr}], lr=lr,
                                         weight_decay=weight_decay, betas=(0.9, 0.99))     # T:99-100
        self.fast = [v for k, v in self.P.items() if k.startswith("render_instance_mlp.mlp.")]
        self.slow = [v for k, v in self.P.items() if k.startswith("render_instance_mlp.slow_mlp.")]
        # grid instance head (tensoRF.py:232-236): its planes / lines at the grid rate, basis matrix + fast MLP at the net rate
        self.inst_grid = [v for k, v in self.P.items() if k.startswith(INST_GRID_KEYS)]
        self.fast_mlp = list(self.fast)                                   # (the EMA pairs the two MLPs' parameters, T:325-329)
        self.fast = [v for k, v in self.P.items() if k.startswith("instance_basis_mat")] + self.fast
        groups = ([{"params": self.inst_grid, "lr": lr * 20}] if self.inst_grid else []) + [{"params": self.fast + ([] if self.dino else self.slow), "lr": lr}]
        self.opt_inst = torch.optim.Adam(groups, lr=lr, weight_decay=weight_decay, betas=(0.9, 0.999))    # T:101-102
        self.scheds = [torch.optim.lr_scheduler.MultiStepLR(o, milestones=list(self.decay_step), gamma=self.decay_gamma) for o in (self.opt_main, self.opt_inst)]

    def on_train_epoch_start(self, epoch, bbox_aabb_reset_epochs=(), grid_upscale_epochs=(), min_grid_dim=128, max_grid_dim=192):
        """T:446-457.  The parameter dict gets NEW tensors from shrink / upsample (the reference assigns new nn.Parameters, F:162-196); the
        optimizers are rebuilt only in the upsample branch -- after a shrink alone they keep stepping the tensors that were replaced."""
        from . import grid_ops
        self.set_epoch(epoch)
        if epoch in bbox_aabb_reset_epochs:
            grid_ops.update_bbox_aabb_and_shrink(self.P, self.cfg)
        if epoch in grid_upscale_epochs:
            n_vox = grid_ops.voxel_schedule(min_grid_dim, max_grid_dim, len(grid_upscale_epochs))[list(grid_upscale_epochs).index(epoch)]
            target = grid_ops.target_resolution(self.cfg, n_vox)
            self.weight_decay = 0                                  # T:454
            grid_ops.upsample_volume_grid(self.P, target)
            self.cfg.grid_dim = tuple(target)
            self.cfg.refresh()                                     # T:456 update_step_size
            self.configure_optimizers()                            # T:457

    def end_of_epoch(self):
        """T:226-228: both schedulers step at the last batch of an epoch."""
        for s_ in self.scheds:
            s_.step()

    def main_pass(self, rays, rgbs, probs, conf, jitter, white_flags, mask=None, segments=None, ce_mode="TTAConf", semantics=None):
        """``segments`` = dict(rays, group, conf, jitter, n_groups): the segment-consistency term of T:185-197 (active from
        segment_optimization_epoch on in the shipped configs)."""
        self.opt_main.zero_grad(set_to_none=True)
        for p in self.fast + self.slow + self.inst_grid:
            p.grad = None
        outs = []
        for ci, i in enumerate(range(0, rays.shape[0], self.chunk)):
            jit = None if jitter is None else jitter[i:i + self.chunk]
            outs.append(orender.render_forward(self.P, rays[i:i + self.chunk], self.cfg, jit, bool(white_flags[ci])))
        rgb = torch.cat([o[0] for o in outs])
        sem = torch.cat([o[1] for o in outs])
        if mask is not None:                      # T:156-158: masked pixels contribute neither colour nor semantics
            keep = mask.to(rgb.dtype)
            rgb, rgbs, conf = rgb * keep[:, None], rgbs * keep[:, None], conf * keep
        if ce_mode != "TTAConf":                  # T:179-182: the label map is the target (a one-hot row for the same row formula) ...
            probs = torch.nn.functional.one_hot(semantics.long(), sem.shape[1]).to(sem.dtype)
            if ce_mode != "NoTTAConf":            # ... and without the confidences every pixel counts, the masked ones too (the mask only zeroes confs, T:158)
                conf = torch.ones_like(conf)
        dreg = torch.stack([o[5] for o in outs]).mean()
        l_rgb = torch.nn.functional.mse_loss(rgb, rgbs)
        l_tv = olosses.total_tv(self.P, self.l_tvd, self.l_tva, self.l_tvs, self.l_tvi, self.sem_on, self.inst_on)
        if not self.sem_on:
            l_sem = torch.zeros(())
        elif self.sce is not None:
            l_sem = (olosses.sce_rows(sem, probs, self.cw, *self.sce) * conf).mean()
        else:
            l_sem = olosses.semantic_ce(sem, probs, conf, self.cw)
        loss = self.l_rgb * (l_rgb + l_tv + dreg * self.l_dist)
        if self.sem_on:                           # before it the semantic MLP's .grad stays None and Adam skips it (T:198)
            loss = loss + self.l_sem * l_sem
        l_seg = torch.zeros(())
        if segments is not None and self.sem_on:
            feats = orender.render_segment_feature(self.P, segments["rays"], self.cfg, segments["jitter"])
            l_seg = olosses.segment_consistency(feats, segments["group"], segments["conf"], self.cw, segments["n_groups"])
            loss = loss + self.l_sem * self.l_seg * l_seg
        loss.backward()
        self.opt_main.step()
        return dict(rgb=rgb.detach(), sem=sem.detach(), loss_rgb=l_rgb.detach(), loss_sem=l_sem.detach(), loss_tv=l_tv.detach(),
                    loss_segment=l_seg.detach())

    def instance_pass(self, rays, labels, conf, jitter):
        self.opt_inst.zero_grad(set_to_none=True)
        # order of the reference: the features (fast AND slow halves) are rendered first (T:214), the EMA step of the slow
        # net happens at the top of the loss (T:258-259) -- so the slow features of step t come from the pre-update weights
        inst, xyz = orender.render_instance_feature(self.P, rays, self.cfg, jitter)
        if self.inst_mode == "slow_fast":
            olosses.ema_(self.slow, self.fast_mlp, 0.9)
            loss = olosses.slow_fast(inst, labels, conf)
        elif self.inst_mode == "linear_assignment":                # T:237-241: Hungarian-matched slots, confidence-weighted CE
            loss, active = olosses.linear_assignment(inst, labels, conf)
            if not active:                                         # a constant: no parameter receives a gradient, Adam skips them all
                return dict(loss=loss.detach(), inst=inst.detach())
        else:                                                      # T:243-250 contrastive (optionally on points + features)
            feats = xyz + inst if self.use_delta else inst
            loss = olosses.contrastive(feats, labels, self.temperature)
            if self.use_delta:
                loss = loss + 0.1 * torch.norm(feats - xyz, dim=-1).mean()
        loss.backward()
        self.opt_inst.step()
        return dict(loss=loss.detach(), inst=inst.detach())
