"""Inference render loop and post-processing -- mirror of reference inference/render_panopli.py:108-140 (per-frame
chunked rendering with is_train=False and the halved step ratio), util/camera.py:86-104 (distance_to_depth),
RP:422-427 (create_instances_from_semantics), RP:371-419 (assign_clusters) and RP:196-263 (MeanShift clustering, which
stays sklearn on the CPU exactly like the reference).
"""
import numpy as np
import torch

from . import _lib, engine


@torch.no_grad()
def render_rays(model, renderer, rays, chunk, white_bg=False):
    """Chunked renderer(model, rays[i:i+chunk], perturb, white_bg, is_train=False) (RP:114-120): returns
    rgb (P,3), semantics (P,C), instances (P,D), distance (P,)."""
    outs = [[], [], [], []]
    chunk = int(chunk) if chunk and int(chunk) > 0 else rays.shape[0]
    for i in range(0, rays.shape[0], chunk):
        o, ctx = engine.render_forward(model, renderer, rays[i:i + chunk], None, bool(white_bg), grad_heads=())
        outs[0].append(o["rgb"]); outs[1].append(o["semantics"]); outs[2].append(o["instances"]); outs[3].append(o["depth"].clone())
        del ctx
    return tuple(torch.cat(x, 0) for x in outs)


def tile_bounds(n, world):
    """Contiguous row-tile boundaries of n rays over `world` ranks (sizes differ by at most one)."""
    return [(n * r) // world for r in range(world + 1)]


@torch.no_grad()
def render_rays_sharded(model, renderer, rays, chunk, white_bg=False, render_fn=None):
    """Multi-GPU frame render (BASELINE configs[4]): the rays of a frame are cut into contiguous row-tiles, one
    per rank; every rank renders its tile with no communication, then ONE all-gather (RCCL over xGMI; 4*(4+C+D) bytes per
    ray) assembles the per-ray outputs on every rank.  With no process group it is ``render_rays``.
    ``render_fn(model, renderer, rays, chunk, white_bg)`` defaults to ``render_rays`` (injectable for CPU tests)."""
    import torch.distributed as dist
    fn = render_fn or render_rays
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return fn(model, renderer, rays, chunk, white_bg)
    world, rank = dist.get_world_size(), dist.get_rank()
    P = rays.shape[0]
    b = tile_bounds(P, world)
    n_mine = b[rank + 1] - b[rank]
    # a rank whose tile is empty (fewer rays than ranks) renders one ray so that it knows the output widths, and keeps none
    mine = fn(model, renderer, rays[b[rank]:b[rank + 1]] if n_mine else rays[:1], chunk, white_bg)
    mine = tuple(x[:n_mine] for x in mine)
    widths = [int(x.shape[1]) if x.dim() == 2 else 1 for x in mine]
    packed = torch.cat([x.reshape(n_mine, w) for x, w in zip(mine, widths)], 1)
    cap = max(b[r + 1] - b[r] for r in range(world))                      # equal-size slots: tiles differ by <= 1 ray
    slot = torch.zeros((cap, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    slot[:packed.shape[0]] = packed
    if dist.get_backend() != "nccl":           # CPU-side backends (gloo in the tests) gather host copies
        host = slot.cpu()
        slots = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(slots, host)
        slots = [s_.to(slot.device) for s_ in slots]
    else:
        slots = [torch.empty_like(slot) for _ in range(world)]
        dist.all_gather(slots, slot)
    full = torch.cat([slots[r][:b[r + 1] - b[r]] for r in range(world)], 0)
    outs, c = [], 0
    for x, w in zip(mine, widths):
        o = full[:, c:c + w]
        outs.append(o.contiguous() if x.dim() == 2 else o.reshape(-1).contiguous())
        c += w
    return tuple(outs)


def distance_to_depth(K, dist):
    """util/camera.py:86-104 for a (H,W) distance image on the device: z = dist / |K^-1 [u, v, 1]|."""
    H, W = dist.shape
    dev = dist.device
    u, v = torch.meshgrid(torch.arange(W, device=dev), torch.arange(H, device=dev), indexing="xy")
    uvh = torch.stack([u.reshape(-1), v.reshape(-1), torch.ones(H * W, device=dev, dtype=torch.long)], -1).to(dist.dtype)
    tmp = uvh @ torch.inverse(torch.as_tensor(K, dtype=dist.dtype, device=dev)).T
    return dist.reshape(-1) / torch.linalg.norm(tmp, dim=1)


def create_instances_from_semantics(instances, semantics, thing_classes):
    """RP:422-427: prepend a column that is +inf for stuff pixels and -inf for thing pixels."""
    stuff = ~torch.isin(semantics.argmax(dim=1), torch.tensor(list(thing_classes), device=semantics.device))
    padded = torch.full((instances.shape[0], instances.shape[1] + 1), -float("inf"), device=instances.device)
    padded[:, 1:] = instances
    padded[stuff, 0] = float("inf")
    return padded


def _one_hot(all_labels, num_images, device):
    all_labels = all_labels + 1                                  # -1,0,..,K-1 -> 0,1,..,K
    n_lab = int(all_labels.max()) + 1
    onehot = torch.zeros((all_labels.shape[0], n_lab), dtype=torch.float64, device=device)
    onehot[torch.arange(all_labels.shape[0], device=device), all_labels] = 1
    return onehot.view(num_images, -1, n_lab)


@torch.no_grad()
def assign_clusters(all_thing_features, all_points_semantics, all_centroids, device, num_images):
    """RP:371-419: per thing class, nearest cached centroid (clift_nearest_centroid on the device); stuff pixels get
    label -1; labels of successive classes are offset so they stay disjoint; returns one-hot (num_images, P, K+1)."""
    feats = torch.as_tensor(all_thing_features, dtype=torch.float32, device=device)
    sem = torch.cat([s.to(device) for s in all_points_semantics], 0).argmax(-1)
    thing = feats[:, 0] == -float("inf")
    f = feats[:, 1:].contiguous()
    labels = torch.full((f.shape[0],), -1, dtype=torch.int64, device=device)
    max_label = 0
    for cls in torch.unique(sem[thing]).tolist():
        cent = torch.as_tensor(np.asarray(all_centroids[cls]), dtype=torch.float32, device=device).contiguous()
        valid = (thing & (sem == cls)).to(torch.uint8).contiguous()
        lab = torch.empty((f.shape[0],), dtype=torch.int32, device=device)
        _lib.call("clift_nearest_centroid", _lib.ptr(f), f.shape[1], f.shape[1], _lib.ptr(cent), cent.shape[0], _lib.ptr(valid),
                  f.shape[0], _lib.ptr(lab), _lib.stream())
        sel = valid.bool()
        labels[sel] = lab[sel].long() + max_label
        if bool(sel.any()):
            max_label = int(labels[sel].max()) + 1
    return _one_hot(labels, num_images, device)


def _hdbscan_fit(pts, cluster_size):
    """RP:236-241 / 321-326: HDBSCAN(min_cluster_size, min_samples=1, allow_single_cluster=True) on the rescaled subsample; returns
    (labels, centroids (K, d) or None when every point is noise).  centroid k = the membership-probability-weighted mean of cluster k's
    points -- the ``weighted_cluster_centroid`` of the hdbscan package the reference imports.  That package is not in this image: the fit
    comes from the package when it is importable and otherwise from ``sklearn.cluster.HDBSCAN`` (scikit-learn >= 1.3: the same algorithm,
    adopted from that package -- mutual-reachability MST, condensed tree, excess-of-mass selection; PARITY UNPINNED against the package
    itself)."""
    try:
        import hdbscan as _pkg                                     # the reference's dependency (requirements.txt)
        cl = _pkg.HDBSCAN(min_cluster_size=cluster_size, min_samples=1, prediction_data=True, allow_single_cluster=True).fit(pts)
        labels, prob = cl.labels_, cl.probabilities_
    except ImportError:
        from sklearn.cluster import HDBSCAN
        cl = HDBSCAN(min_cluster_size=cluster_size, min_samples=1, allow_single_cluster=True, copy=True).fit(pts)
        labels, prob = cl.labels_, cl.probabilities_
    ids = [k for k in np.unique(labels) if k != -1]
    if not ids:
        return labels, None
    return labels, np.stack([np.average(pts[labels == k], weights=prob[labels == k], axis=0) for k in ids])


def _nearest_centroid(points, centroids, device):
    """RP:244-254: every point to its nearest centroid (squared distances through clift_nearest_centroid on the device; ties -> lowest index,
    as torch.argmin over cdist)."""
    f = torch.as_tensor(np.ascontiguousarray(points), dtype=torch.float32, device=device).contiguous()
    c = torch.as_tensor(np.ascontiguousarray(centroids), dtype=torch.float32, device=device).contiguous()
    if f.device.type != "cuda":
        return torch.argmin(torch.cdist(f, c), dim=-1).cpu().numpy()
    valid = torch.ones((f.shape[0],), dtype=torch.uint8, device=device)
    lab = torch.empty((f.shape[0],), dtype=torch.int32, device=device)
    _lib.call("clift_nearest_centroid", _lib.ptr(f), f.shape[1], f.shape[1], _lib.ptr(c), c.shape[0], _lib.ptr(valid), f.shape[0], _lib.ptr(lab),
              _lib.stream())
    return lab.cpu().numpy().astype(np.int64)


def cluster(all_thing_features, bandwidth, device, num_images, num_points=50000, use_silverman=False, use_dbscan=False, cluster_size=500):
    """RP:196-263: 3-sigma outlier filter, per-axis rescale to the
    unit box, a 50000-point subsample drawn with ``np.random.choice`` from numpy's GLOBAL generator exactly as the reference
    does (seed it with ``np.random.seed`` for reproducible runs), sklearn MeanShift (optionally with Silverman's bandwidth) or, with
    ``use_dbscan``, HDBSCAN (``_hdbscan_fit``), then every pixel is assigned to its nearest cluster.  Returns (one-hot (num_images, P, K+1) float64, centroids in feature
    units).  Scenes with fewer thing pixels than ``num_points`` use all of them (the reference raises there)."""
    from sklearn.cluster import MeanShift
    feats = np.asarray(all_thing_features)
    thing = feats[..., 0] == -float("inf")
    f_th = feats[thing][:, 1:]
    f_all = feats[:, 1:]
    mu, sd = f_th.mean(axis=0), f_th.std(axis=0)
    keep = np.all(np.abs(f_th - mu) < 3 * sd, axis=1)
    cf = f_th[keep]
    bias = cf.min(axis=0)
    factor = 1 / (cf.max(axis=0) - cf.min(axis=0))
    cr = (cf - bias) * factor
    idx = np.random.choice(cr.shape[0], num_points, replace=False) if cr.shape[0] >= num_points else np.arange(cr.shape[0])
    pts = cr[idx]
    if use_silverman:
        from scipy.stats import gaussian_kde
        bandwidth = gaussian_kde(pts.T, bw_method="silverman").covariance_factor()
    if use_dbscan:                                                # RP:236-255: HDBSCAN on the subsample, then EVERY point to its nearest centroid
        _, centers = _hdbscan_fit(pts, cluster_size)
        if centers is None:
            raise _lib.CliftError("HDBSCAN found no cluster (every sampled point is noise); the reference fails here too (np.stack of an empty list)")
        all_labels = _nearest_centroid((f_all.reshape(-1, f_all.shape[-1]) - bias) * factor, centers, device)
    else:
        ms = MeanShift(bandwidth=bandwidth, cluster_all=False, bin_seeding=True, min_bin_freq=10).fit(pts)
        all_labels = ms.predict((f_all.reshape(-1, f_all.shape[-1]) - bias) * factor)
        centers = ms.cluster_centers_
    all_labels[~thing] = -1
    all_labels = all_labels + 1                                   # -1,0,..,K-1 -> 0,1,..,K
    K1 = centers.shape[0] + 1                                      # width = number of centroids + 1 (not max label + 1)
    onehot = torch.zeros((all_labels.shape[0], K1), dtype=torch.float64, device=device)
    onehot[torch.arange(all_labels.shape[0], device=device), torch.as_tensor(all_labels, dtype=torch.int64, device=device)] = 1
    return onehot.view(num_images, -1, K1), centers / factor + bias


def cluster_segmentwise(all_thing_features, all_points_semantics, bandwidth, device, num_images, num_points=50000, use_silverman=False,
                        use_dbscan=False, cluster_size=500):
    """RP:265-368: the clustering of ``cluster`` (MeanShift, or HDBSCAN with ``use_dbscan``) run separately inside every predicted thing class, labels of
    successive classes offset so they stay disjoint; classes with fewer than 100 (filtered) points get no instances (-1).
    Returns (one-hot (num_images, P, max label + 2) float64, concatenated centroids in feature units).  Like the reference,
    a class that is skipped for having too few points still appends the previous class's centroids (rescaled with its own
    statistics) to the returned list -- only the one-hot output is consumed by the render script."""
    from sklearn.cluster import MeanShift
    sem = torch.cat([s_.cpu() for s_ in all_points_semantics], 0).argmax(-1).numpy()
    feats = np.asarray(all_thing_features)
    thing = feats[..., 0] == -float("inf")
    f_th = feats[thing][:, 1:]
    n_all = feats.shape[0]
    th_sem = sem[thing]
    all_labels = np.zeros(n_all, dtype=np.int32)
    th_labels = np.zeros(f_th.shape[0], dtype=np.int32)
    max_label = 0
    cents_all, centroids = [], None
    for cls in np.unique(th_sem):
        m = th_sem == cls
        fc = f_th[m]
        mu, sd = fc.mean(axis=0), fc.std(axis=0)
        cf = fc[np.all(np.abs(fc - mu) < 3 * sd, axis=1)]
        if cf.shape[0] == 0:
            th_labels[m] = -1
            continue
        bias = cf.min(axis=0)
        factor = 1 / (cf.max(axis=0) - cf.min(axis=0))
        cr = (cf - bias) * factor
        idx = np.arange(cr.shape[0]) if cr.shape[0] < num_points else np.random.choice(cr.shape[0], num_points, replace=False)
        pts = cr[idx]
        if use_dbscan:                                              # RP:320-342
            _, cents = _hdbscan_fit(pts, cluster_size)
            if cents is not None:
                centroids = cents
                lab = _nearest_centroid((fc.reshape(-1, fc.shape[-1]) - bias) * factor, cents, device).astype(np.int32)
            else:
                lab = -1 * np.ones(fc.shape[0], dtype=np.int32)
        elif pts.shape[0] < 100:                                    # too few points for MeanShift
            lab = -1 * np.ones(fc.shape[0], dtype=np.int32)
        else:
            bw = bandwidth
            if use_silverman:
                from scipy.stats import gaussian_kde
                bw = gaussian_kde(pts.T, bw_method="silverman").covariance_factor()
            ms = MeanShift(bandwidth=bw, cluster_all=False, bin_seeding=True, min_bin_freq=10).fit(pts)
            centroids = ms.cluster_centers_
            lab = ms.predict((fc.reshape(-1, fc.shape[-1]) - bias) * factor)
        if centroids is not None:
            cents_all.append(centroids / factor + bias)
        lab[lab != -1] += max_label
        if np.any(lab != -1):
            max_label = lab.max() + 1
        th_labels[m] = lab
    all_labels[thing] = th_labels
    all_labels[~thing] = -1
    onehot = _one_hot(torch.as_tensor(all_labels, dtype=torch.int64, device=device), num_images, device)
    return onehot, (np.concatenate(cents_all, axis=0) if cents_all else np.zeros((0, f_th.shape[1])))


def psnr(image_pred, image_gt):
    """util/metrics.py:25-26."""
    return -10 * torch.log10(torch.mean((image_pred.detach() - image_gt) ** 2))


class ConfusionMatrix:
    """util/metrics.py:29-75: confusion counts + robust mIoU (classes that are rare in both marginals are ignored)."""

    def __init__(self, num_classes, ignore_class=None, robust=0.005):
        self.n, self.ignore, self.robust = num_classes, list(ignore_class or []), robust
        self.cm = np.zeros((num_classes, num_classes))

    def _matrix(self, gt, pred):
        m = (gt >= 0) & (gt < self.n)
        return np.bincount(self.n * gt[m].astype(int) + pred[m], minlength=self.n ** 2).reshape(self.n, self.n)

    def _miou(self, cm):
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
        a0, a1, tot = cm.sum(0), cm.sum(1), cm.sum()
        nonrobust = np.where((a0 / tot < self.robust) & (a1 / tot < self.robust))[0].tolist()
        for i in self.ignore + nonrobust:
            iou[i] = np.nan
        return np.nanmean(iou)

    def add_batch(self, gt_image, pre_image, return_miou=False):
        cm = self._matrix(np.asarray(gt_image), np.asarray(pre_image))
        self.cm += cm
        if return_miou:
            return self._miou(cm)

    def get_miou(self):
        return self._miou(self.cm)
