import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


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
