"""Losses of the path -- mirrors of reference model/loss/loss.py:9-26,62-82 and the slow-fast branch of
trainer/train_panopli_tensorf.py:256-310,325-329, running as HIP kernels (clift_tv_fwd_bwd, clift_contrastive,
clift_slow_fast, clift_ema).  Each returns a differentiable 0-dim tensor through a tiny autograd.Function whose
backward only scales the gradient the kernel already produced in the forward launch.
"""
import torch
from torch import nn

from . import _lib


class _Scaled(torch.autograd.Function):
    """loss value + precomputed d loss / d x: backward = upstream * grad."""

    @staticmethod
    def forward(ctx, x, loss, grad):
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None


def contrastive_loss(features, instance_labels, temperature, return_grad=False):
    """loss.py:62-82.  features (B,E) float32 cuda, instance_labels (B,) integer."""
    f = _lib.f32(features, "features").contiguous()
    B, E = f.shape
    y = instance_labels.to(device=f.device, dtype=torch.int32).contiguous()
    loss = torch.empty((1,), dtype=torch.float32, device=f.device)
    grad = torch.empty_like(f)
    work = torch.empty((max(4 * B, 4),), dtype=torch.float32, device=f.device)
    _lib.call("clift_contrastive", _lib.ptr(f), _lib.ptr(y), B, E, float(temperature), _lib.ptr(loss), _lib.ptr(grad),
              _lib.ptr(work), _lib.stream())
    if return_grad:
        return loss[0], grad
    return _Scaled.apply(features, loss[0], grad)


def slow_fast_loss(instance_features, labels_gt, confidences, return_grad=False):
    """Slow-fast branch of calculate_instance_clustering_loss (T:261-309, use_proj=False); the EMA step that the
    reference performs first (T:258-259) is ``ema_update``.  instance_features (B,2E) = [fast | slow]."""
    f = _lib.f32(instance_features, "instance_features").contiguous()
    B, D = f.shape
    E = D // 2
    y = labels_gt.to(device=f.device, dtype=torch.int32).contiguous()
    conf = _lib.f32(confidences.to(f.device), "confidences").contiguous()
    loss = torch.empty((1,), dtype=torch.float32, device=f.device)
    grad = torch.empty_like(f)
    work = torch.empty((8 * B + 2 * B * E + 8,), dtype=torch.float32, device=f.device)
    _lib.call("clift_slow_fast", _lib.ptr(f), _lib.ptr(y), _lib.ptr(conf), B, E, _lib.ptr(loss), _lib.ptr(grad),
              _lib.ptr(work), _lib.stream())
    if return_grad:
        return loss[0], grad
    return _Scaled.apply(instance_features, loss[0], grad)


@torch.no_grad()
def create_virtual_gt_with_linear_assignment(labels_gt, predicted_scores):
    """T:332-344: match the (sorted, first E) 2-D instance ids of an image to the E output slots: cost[id][slot] = -(sum of the slot's softmax
    probability over the id's rays / (count + 1e-4)), Hungarian method on the HOST like the reference (scipy; an L x E matrix, L <= E), every
    ray of a matched id gets that slot as its class, every other ray class 0.  The per-id sums are one one-hot product on the device; what crosses
    the bus is the L x E cost matrix."""
    import numpy as np
    import scipy.optimize
    E = predicted_scores.shape[-1]
    ids = torch.unique(labels_gt)[:E]                              # torch.unique sorts
    prob = torch.softmax(predicted_scores.detach().to(torch.float32), dim=-1)
    slot_of = torch.searchsorted(ids, labels_gt.contiguous())
    slot_of = torch.where((slot_of < ids.numel()) & (ids[slot_of.clamp_max(ids.numel() - 1)] == labels_gt), slot_of, torch.full_like(slot_of, ids.numel()))
    # per-id sums as a float64 one-hot product: a fixed summation order (an index_add_ is a race of atomics on the device, and a near-tie in
    # the cost matrix could flip the assignment from run to run).  This mode synchronises with the host (the .cpu() below, like the
    # reference's own scipy call): ``nosync`` steps do not apply to it.
    member = torch.zeros((ids.numel() + 1, slot_of.numel()), dtype=torch.float64, device=prob.device).scatter_(0, slot_of.reshape(1, -1), 1.0)
    sums, cnt = (member @ prob.to(torch.float64)).to(torch.float32), member.sum(1).to(torch.float32)
    cost = (-(sums[:-1] / (cnt[:-1, None] + 1e-4))).cpu().numpy().astype(np.float64)
    rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))
    table = torch.zeros(ids.numel() + 1, dtype=labels_gt.dtype)
    table[torch.as_tensor(rows, dtype=torch.long)] = torch.as_tensor(cols, dtype=labels_gt.dtype)
    return table.to(labels_gt.device)[slot_of]


def linear_assignment_loss(instance_features, labels_gt, confidences, return_grad=False):
    """The "linear_assignment" branch of calculate_instance_clustering_loss (T:237-241; the template's default instance_loss_mode, the
    Panoptic-Lifting baseline): mean over rays of CrossEntropyLoss(reduction='none')(scores, matched slot) * confidence -- unless every ray's
    argmax already is its slot: then the term is the constant 0 and NO gradient flows ("should never reinforce correct labels").
    Returns (loss, grad or None) with ``return_grad``; grad None means the inactive case."""
    f = _lib.f32(instance_features, "instance_features").contiguous()
    n, E = f.shape
    y = labels_gt.to(f.device)
    conf = _lib.f32(confidences.to(f.device), "confidences").contiguous()
    target = create_virtual_gt_with_linear_assignment(y, f)
    if not bool(torch.any(target != f.argmax(dim=-1))):
        zero = torch.zeros((), dtype=torch.float32, device=f.device)
        return (zero, None) if return_grad else zero.requires_grad_(True)
    onehot = torch.zeros_like(f).scatter_(1, target.reshape(-1, 1).to(torch.int64), 1.0)
    rows = torch.empty((n,), dtype=torch.float32, device=f.device)
    grows = torch.empty_like(f)
    _lib.call("clift_semantic_loss_rows", _lib.ptr(f), _lib.ptr(onehot), None, n, E, 0, 1.0, 0.0, _lib.ptr(rows), _lib.ptr(grows), _lib.stream())
    loss = (rows * conf).mean()
    grad = grows * (conf / n)[:, None]
    if return_grad:
        return loss, grad
    return _Scaled.apply(instance_features, loss, grad)


def get_semantic_weights(reweight_classes, fg_classes, num_semantic_classes):
    """loss.py:29-33: per-class weights of the semantic losses -- ones, foreground ("thing") classes doubled when
    ``reweight_classes`` (config reweight_fg).  The trainer then overwrites entry 0 with config.weight_class_0 (T:69-70)."""
    weights = torch.ones([num_semantic_classes]).float()
    if reweight_classes:
        weights[fg_classes] = 2
    return weights


class _SemanticRows(torch.autograd.Function):
    """Per-pixel semantic loss (reduction='none') with the gradient produced by the same kernel launch."""

    @staticmethod
    def forward(ctx, pred, target, class_weights, sce, alpha, beta):
        x = _lib.f32(pred, "pred").contiguous()
        p = _lib.f32(target.to(x.device), "labels_probabilities").contiguous()
        if p.shape != x.shape:
            raise ValueError(f"soft targets must have the prediction's shape {tuple(x.shape)}, got {tuple(p.shape)}")
        n, c = x.shape
        cw = None if class_weights is None else _lib.f32(torch.as_tensor(class_weights).to(x.device), "class_weights").contiguous()
        rows = torch.empty((n,), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        _lib.call("clift_semantic_loss_rows", _lib.ptr(x), _lib.ptr(p), _lib.ptr(cw), n, c, int(sce), float(alpha), float(beta),
                  _lib.ptr(rows), _lib.ptr(grad), _lib.stream())
        ctx.save_for_backward(grad)
        return rows

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g[:, None] * grad, None, None, None, None, None


class SCELoss(nn.Module):
    """loss.py:36-59: symmetric cross entropy on soft targets, one value per pixel:
    alpha * CrossEntropyLoss(weight, reduction='none')(pred, p) + beta * RCE, where the reverse term re-softmaxes the
    class-weighted logits: RCE = -sum_c clamp(softmax(pred * w), 1e-8, 1)_c * log(clamp(p_c, 1e-8, 1)) * w_c."""

    def __init__(self, alpha, beta, class_weights):
        super().__init__()
        self.alpha, self.beta, self.class_weights = alpha, beta, class_weights

    def forward(self, pred, labels_probabilities):
        return _SemanticRows.apply(pred, labels_probabilities, self.class_weights, 1, self.alpha, self.beta)


class SoftTargetCrossEntropy(nn.Module):
    """torch.nn.CrossEntropyLoss(reduction='none', weight=w) on probability targets (the reference's ``loss_semantics`` when
    use_symmetric_ce is off, T:75), through the same kernel."""

    def __init__(self, class_weights=None):
        super().__init__()
        self.class_weights = class_weights

    def forward(self, pred, labels_probabilities):
        return _SemanticRows.apply(pred, labels_probabilities, self.class_weights, 0, 1.0, 0.0)


@torch.no_grad()
def ema_update(slownet, fastnet, momentum):
    """trainer T:325-329: slow <- momentum*slow + (1-momentum)*fast, parameter by parameter."""
    for pf, ps in zip(fastnet.parameters(), slownet.parameters()):
        if ps.is_contiguous() and pf.is_contiguous():
            _lib.call("clift_ema", _lib.ptr(ps), _lib.ptr(pf), ps.numel(), float(momentum), _lib.stream())
        else:
            ps.mul_(momentum).add_((1 - momentum) * pf)


class TVLoss(nn.Module):
    """loss.py:9-26 on a (1,C,H,W) tensor (any strides; channels-last is the native layout)."""

    def forward(self, x):
        return _TVFn.apply(x)


class _TVFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        _lib.f32(x, "x")
        b, c, h, w = x.shape
        if b != 1:
            raise NotImplementedError("clift TVLoss: batch must be 1")
        xc = x.permute(0, 2, 3, 1).contiguous()          # (1,H,W,C): a no-op view for channels-last parameters
        loss = torch.zeros((1,), dtype=torch.float32, device=x.device)
        grad = torch.zeros_like(xc)
        _lib.call("clift_tv_fwd_bwd", _lib.ptr(xc), h, w, c, 1.0, _lib.ptr(grad), _lib.ptr(loss), _lib.stream())
        ctx.save_for_backward(grad)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (g * grad).permute(0, 3, 1, 2)
