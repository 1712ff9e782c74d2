"""Oracle: losses on the path.  Test infrastructure only (see oracle/__init__.py)."""
import torch
import torch.nn.functional as F


def tv_plane(x):
    """Reference model/loss/loss.py:14-22 for one (1,C,H,W) tensor."""
    b, c, h, w = x.shape
    cnt_h = c * (h - 1) * w + 1e-4
    cnt_w = c * h * (w - 1) + 1e-4
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    return 2 * (h_tv / cnt_h + w_tv / cnt_w) / b


def total_tv(P, lambda_density=0.1, lambda_appearance=0.01, lambda_semantics=0.02, lambda_instances=0.02, sem_on=True, inst_on=True):
    """tensoRF.py:248-290 (total_tv_loss): density and appearance PLANES x 1e-2; semantic / instance grids, where the head has one, planes x 1e-2
    + LINES x 1e-3, switched on with their loss terms (``sem_on``: epoch >= late_semantic_optimization, ``inst_on``: epoch >=
    instance_optimization_epoch)."""
    d = sum(tv_plane(P[f"density_plane.{i}"]) * 1e-2 for i in range(3))
    a = sum(tv_plane(P[f"appearance_plane.{i}"]) * 1e-2 for i in range(3))
    out = d * lambda_density + a * lambda_appearance
    for pre, lam, on in (("semantic", lambda_semantics, sem_on), ("instance", lambda_instances, inst_on)):
        if on and f"{pre}_plane.0" in P:
            out = out + lam * sum(tv_plane(P[f"{pre}_plane.{i}"]) * 1e-2 + tv_plane(P[f"{pre}_line.{i}"]) * 1e-3 for i in range(3))
    return out


def contrastive(features, labels, temperature):
    """model/loss/loss.py:62-82.  mask excludes the diagonal; temperature applies to *positive* pairs,
    1 to negatives; denominator includes the diagonal; rows with p == 0 are dropped, sum / B."""
    B = features.shape[0]
    mask = (labels.view(-1, 1) == labels.view(1, -1))
    mask = mask & ~torch.eye(B, dtype=torch.bool)
    d2 = ((features[:, None, :] - features[None, :, :]) ** 2).sum(-1)
    tau = torch.where(mask, torch.full_like(d2, float(temperature)), torch.ones_like(d2))
    logits = torch.exp(torch.exp(-d2 / tau))
    p = (logits * mask).sum(-1)
    Z = logits.sum(-1)
    prob = p / Z
    return -(torch.log(prob[prob != 0])).sum() / B


def ema_(slow_params, fast_params, momentum=0.9):
    """trainer/train_panopli_tensorf.py:325-329."""
    with torch.no_grad():
        for s, f in zip(slow_params, fast_params):
            # .data like the reference: the update happens between the forward and the backward of the same step (T:214 vs
            # T:258) and must not invalidate the autograd graph (the slow half only ever receives zero gradients)
            s.data.mul_(momentum).add_((1 - momentum) * f.detach().data)


def slow_fast(inst_feats, labels, conf):
    """trainer/train_panopli_tensorf.py:261-309 (use_proj=False); the EMA step (T:258-259) is done by the
    caller.  inst_feats (B, 2E) = [fast | slow]; first half of the rays is the fast set, second the slow set."""
    E = inst_feats.shape[-1] // 2
    fast, slow = inst_feats[:, :E], inst_feats[:, E:].detach()
    B = labels.shape[0]
    half = B // 2
    fm = torch.zeros(B, dtype=torch.bool)
    fm[:half] = True
    sm = ~fm
    fast_labels = torch.unique(labels[fm])
    slow_labels = torch.unique(labels[sm])
    if len(fast_labels) == 0 or len(slow_labels) == 0:
        return torch.tensor(0.0)
    cents = torch.stack([slow[sm & (labels == l)].mean(0) for l in slow_labels])
    inter = fast_labels[torch.isin(fast_labels, slow_labels)]
    loss = 0
    for l in inter:
        m_ = fm & (labels == l)
        c_ = cents[slow_labels == l]
        d2 = ((fast[m_] - c_) ** 2).sum(-1)
        loss = loss + -1.0 * (torch.exp(-d2) * conf[m_]).mean()
    if inter.shape[0] > 0:
        loss = loss / inter.shape[0]
    lab = labels[fm][:, None] == labels[sm][None, :]
    sim = torch.exp(-torch.cdist(fast[fm], slow[sm], p=2))
    logits = torch.exp(sim)
    prob = (logits * lab).sum(-1) / logits.sum(-1)
    return loss + -(torch.log(prob[prob != 0])).mean()


def semantic_ce(log_probs, target_probs, conf, class_weight):
    """CrossEntropyLoss(reduction='none', weight=w) applied to the renderer's log-probabilities with soft
    targets, times per-pixel confidence, mean (trainer/train_panopli_tensorf.py:75,177-178)."""
    return (F.cross_entropy(log_probs, target_probs, weight=class_weight, reduction="none") * conf).mean()


def semantic_weights(reweight_classes, fg_classes, num_semantic_classes):
    """model/loss/loss.py:29-33."""
    w = torch.ones(num_semantic_classes)
    if reweight_classes:
        w[torch.as_tensor(fg_classes, dtype=torch.long)] = 2.0
    return w


def sce_rows(pred, target_probs, class_weight, alpha, beta):
    """model/loss/loss.py:45-59 (SCELoss.forward): alpha * weighted soft-target CE + beta * reverse CE, where the reverse term uses
    softmax of the class-weighted predictions clamped to [1e-8, 1] and log of the clamped targets, again class-weighted."""
    ce = F.cross_entropy(pred, target_probs, weight=class_weight, reduction="none")
    w = class_weight[None, :]
    q = torch.softmax(pred * w, dim=1).clamp(min=1e-8, max=1.0)
    rce = -(q * torch.log(target_probs.clamp(min=1e-8, max=1.0)) * w).sum(1)
    return alpha * ce + beta * rce


def segment_consistency(seg_features, group, conf, class_weight, n_groups):
    """trainer/train_panopli_tensorf.py:189-194: the per-segment mean of the rendered semantic features (torch_scatter.scatter_mean
    restated: sum / max(count, 1)) picks ONE class per 2D segment (argmax); every ray of the segment is then pulled towards it:
    mean over rays of CrossEntropyLoss(reduction='none', weight=w)(features, class) * confidence."""
    mean = torch.zeros(n_groups, seg_features.shape[1]).index_add_(0, group, seg_features)
    cnt = torch.zeros(n_groups).index_add_(0, group, torch.ones(group.shape[0])).clamp_(min=1)
    target = (mean / cnt[:, None])[group].argmax(-1)
    return (F.cross_entropy(seg_features, target, weight=class_weight, reduction="none") * conf).mean()


def virtual_labels_linear_assignment(labels_gt, scores):
    """trainer/train_panopli_tensorf.py:332-344 (create_virtual_gt_with_linear_assignment): the (sorted, first E) 2-D instance ids of an image
    are matched to the E output slots by the Hungarian method on cost[id][slot] = -(mean softmax probability of slot over the id's rays, count
    + 1e-4 in the denominator); every ray of a matched id takes that slot as its class, everything else class 0."""
    import numpy as np
    import scipy.optimize
    ids = sorted(torch.unique(labels_gt).cpu().tolist())[:scores.shape[-1]]
    prob = torch.softmax(scores.detach(), dim=-1)
    cost = np.zeros([len(ids), prob.shape[-1]])
    for i, l in enumerate(ids):
        sel = labels_gt == l
        cost[i, :] = -(prob[sel, :].sum(dim=0) / (sel.sum() + 1e-4)).cpu().numpy()
    rows, cols = scipy.optimize.linear_sum_assignment(np.nan_to_num(cost))
    new = torch.zeros_like(labels_gt)
    for a, i in enumerate(rows):
        new[labels_gt == ids[i]] = int(cols[a])
    return new


def linear_assignment(scores, labels_gt, conf):
    """trainer/train_panopli_tensorf.py:237-241: confidence-weighted cross entropy of the instance scores against the matched slots -- unless
    every ray's argmax already is its slot, in which case the term is a constant 0 without a gradient ("should never reinforce correct
    labels").  Returns (loss, active)."""
    target = virtual_labels_linear_assignment(labels_gt, scores)
    if bool(torch.any(target != scores.argmax(dim=-1))):
        return (F.cross_entropy(scores, target, reduction="none") * conf).mean(), True
    return torch.zeros(()), False
