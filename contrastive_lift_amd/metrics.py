"""Panoptic quality -- restatement of reference util/panoptic_quality.py:205-247 (a torchmetrics-derived PQ with the
reference's "non-robust class" filtering), vectorised over packed (category, instance) ids instead of Python dicts of
colour tuples.  Host-side evaluation code (not on the per-ray path); pinned by tests/golden/g11_metrics.npz.

Semantics kept: classes absent from both maps and classes covering < ``robust`` of either map are dropped from
things/stuff; stuff instance ids are zeroed; unknown categories become void = (1 + max id, 0); a (pred, target) segment
pair of equal category matches when IoU > 0.5 with the void overlaps removed from the union; unmatched segments that
are more than half void on the other side are ignored; PQ/SQ/RQ are averaged over the remaining categories.
"""
import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _non_robust(sem_pred, sem_tgt, thr):
    out = set()
    for s in (sem_pred, sem_tgt):
        u, c = np.unique(s, return_counts=True)
        out |= set(u[(c / c.sum()) < thr].tolist())
    return out


def panoptic_quality(preds, target, things, stuff, allow_unknown_preds_category=False, robust=0.005):
    """preds, target: (..., 2) integer [category, instance].  Returns (pq, sq, rq) as 0-dim float64 tensors."""
    p = _np(preds).reshape(-1, 2).astype(np.int64).copy()
    t = _np(target).reshape(-1, 2).astype(np.int64).copy()
    if p.shape != t.shape:
        raise ValueError("Expected argument `preds` and `target` to have the same shape")
    things, stuff = set(int(x) for x in things), set(int(x) for x in stuff)
    present = set(np.unique(p[:, 0]).tolist()) | set(np.unique(t[:, 0]).tolist())
    drop = ((things | stuff) - present) | _non_robust(p[:, 0], t[:, 0], robust)
    things, stuff = things - drop, stuff - drop
    if things & stuff:
        raise ValueError("Expected arguments `things` and `stuffs` to have distinct keys.")
    void_cat = 1 + max([0] + list(things) + list(stuff))
    cats = list(things) + list(stuff)                      # things first, like the reference's continuous ids
    cid = {c: i for i, c in enumerate(cats)}

    def prep(img, allow_unknown):
        is_stuff = np.isin(img[:, 0], list(stuff))
        is_thing = np.isin(img[:, 0], list(things))
        img[is_stuff, 1] = 0
        if not allow_unknown and not np.all(is_stuff | is_thing):
            raise ValueError("Unknown categories found in preds")
        unk = ~(is_stuff | is_thing)
        img[unk, 0], img[unk, 1] = void_cat, 0
        return img
    p, t = prep(p, allow_unknown_preds_category), prep(t, True)
    base = int(max(p[:, 1].max(initial=0), t[:, 1].max(initial=0))) + 1
    kp, kt = p[:, 0] * base + p[:, 1], t[:, 0] * base + t[:, 1]
    void_key = void_cat * base
    up, ip, ap = np.unique(kp, return_inverse=True, return_counts=True)
    ut, it, at = np.unique(kt, return_inverse=True, return_counts=True)
    inter = np.zeros((len(up), len(ut)), np.int64)
    np.add.at(inter, (ip, it), 1)
    vp = int(np.searchsorted(up, void_key)) if void_key in up else -1          # void row / column, if present
    vt = int(np.searchsorted(ut, void_key)) if void_key in ut else -1
    p_void_t = inter[:, vt] if vt >= 0 else np.zeros(len(up), np.int64)         # pred segment ∩ void target
    void_p_t = inter[vp, :] if vp >= 0 else np.zeros(len(ut), np.int64)         # void pred ∩ target segment
    n = len(cats)
    iou_sum, tp, fp, fn = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    matched_p, matched_t = np.zeros(len(up), bool), np.zeros(len(ut), bool)
    pi, ti = np.nonzero(inter)
    for a, b in zip(pi.tolist(), ti.tolist()):
        if ut[b] == void_key or up[a] // base != ut[b] // base:
            continue
        cat = int(up[a] // base)
        if cat not in cid:
            continue
        i_ab = inter[a, b]
        union = ap[a] - p_void_t[a] + at[b] - void_p_t[b] - i_ab
        iou = i_ab / union
        if iou > 0.5:
            matched_p[a] = matched_t[b] = True
            iou_sum[cid[cat]] += iou
            tp[cid[cat]] += 1
    for b in np.nonzero(~matched_t)[0].tolist():
        if ut[b] == void_key or void_p_t[b] / at[b] > 0.5:
            continue
        fn[cid[int(ut[b] // base)]] += 1
    for a in np.nonzero(~matched_p)[0].tolist():
        if up[a] == void_key or p_void_t[a] / ap[a] > 0.5:
            continue
        fp[cid[int(up[a] // base)]] += 1
    den = tp + 0.5 * fp + 0.5 * fn
    with np.errstate(divide="ignore", invalid="ignore"):
        pq = np.where(den > 0, iou_sum / den, 0.0)
        sq = np.where(tp > 0, iou_sum / tp, 0.0)
        rq = np.where(den > 0, tp / den, 0.0)
    f = lambda v: torch.tensor(float(np.mean(v)) if len(v) else float("nan"), dtype=torch.float64)
    return f(pq), f(sq), f(rq)
