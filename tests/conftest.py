import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota).  On the GPU boxes the affinity mask (and
    torch's default thread count) is the whole host while the container's quota is a fraction of it: the CPU oracle runs with hundreds
    of threads on a dozen cores crawl (the round-3 suite went from minutes to a timeout that way)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


torch.set_num_threads(usable_cores())


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def rel_close(a, b, rtol=1e-3, atol=None, what=""):
    """Tolerance used across the parity tests (BASELINE.json north_star: 1e-3 relative, fp32):
    |a-b| <= rtol*|b| + atol with atol defaulting to rtol * 1e-2 * max|b| (guards exact zeros)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if atol is None:
        atol = rtol * 1e-2 * float(b.abs().max()) if b.numel() else 0.0
    err = (a - b).abs()
    bound = rtol * b.abs() + atol
    bad = err > bound
    if bool(bad.any()):
        i = int(torch.argmax(err - bound))
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} outside rtol={rtol} atol={atol:.3g}; "
                             f"worst a={a.reshape(-1)[i].item():.8g} b={b.reshape(-1)[i].item():.8g}")


def grad_close(a, b, what="", rtol=2e-3, scale_atol=1e-3, outlier_frac=1e-3, outlier_cap=1e-2):
    """Gradient tensors are sums of mixed-sign per-sample terms, so their error is relative to the tensor's scale.
    All elements within rtol*|b| + scale_atol*max|b|, except at most ``outlier_frac`` of them (samples whose weight sits
    within float round-off of the 1e-4 activity threshold flip in/out of the appearance pass between implementations),
    which must still be within outlier_cap*max|b|."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    mx = float(b.abs().max()) if b.numel() else 0.0
    err = (a - b).abs()
    bad = err > rtol * b.abs() + scale_atol * mx + 1e-12
    nbad = int(bad.sum())
    # (a hidden unit on the other side of its ReLU kink for ONE sample moves one entry of a bias gradient: small tensors are allowed two such
    # entries -- two of 256 were seen with the fp32x6 kernels forced on, whose round-off differs from the oracle's as the exact kernels' does)
    allow = max(2 if b.numel() <= 1024 else 1, int(outlier_frac * b.numel()))
    assert nbad <= allow, f"{what}: {nbad}/{b.numel()} elements outside tolerance (max|b|={mx:.3g})"
    assert float(err.max()) <= outlier_cap * mx + 1e-12, f"{what}: max error {float(err.max()):.3g} vs scale {mx:.3g}"


@pytest.fixture(autouse=True)
def _restore_default_mlp_mode():
    """several tests switch the process-wide MLP arithmetic and used to leave it at "fp32", so that what the rest of the suite
    exercised depended on test order.  Every test now starts and ends in the shipped default (or CLIFT_MLP_DTYPE)."""
    from contrastive_lift_amd import engine
    want = os.environ.get("CLIFT_MLP_DTYPE", engine.DEFAULT_MLP_DTYPE)
    engine.set_mlp_precision(want)
    yield
    engine.set_mlp_precision(want)
