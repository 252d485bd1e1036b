import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def t(a, dtype=None, device="cpu"):
    a = np.asarray(a)
    x = torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x.to(device)


def relerr(a, b):
    """Norm-wise relative error ‖a−b‖/‖b‖ (b = reference); absolute if b is ~0."""
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    den = b.norm().item()
    num = (a - b).norm().item()
    return num / den if den > 1e-30 else num


def assert_close(a, b, rel=1e-4, abs_=0.0, what=""):
    e = relerr(a, b)
    a_ = torch.as_tensor(a).detach().double().cpu()
    b_ = torch.as_tensor(b).detach().double().cpu()
    assert a_.shape == b_.shape, f"{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    if e <= rel:
        return
    if abs_ > 0 and (a_ - b_).abs().max().item() <= abs_:
        return
    raise AssertionError(f"{what}: rel err {e:.3e} > {rel:.1e} (max abs diff {(a_ - b_).abs().max().item():.3e})")
