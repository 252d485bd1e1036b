import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

# Several tests import the read-only reference (/root/reference) when it is mounted: no bytecode is ever written next to its sources
# (67 .pyc files had appeared there), whichever test imports it first.
sys.dont_write_bytecode = True

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


def maxerr(a, b):
    """max|a-b| / max|b|: the element-wise twin of relerr — a wrong value at one pixel in a
    thousand moves this, not the norm."""
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    den = b.abs().max().item() if b.numel() else 0.0
    num = (a - b).abs().max().item() if b.numel() else 0.0
    return num / den if den > 1e-30 else num


def assert_grad_close(a, b, rel=1e-4, max_rel=None, masks=(), what=""):
    """Norm-wise AND max-abs comparison of a dense gradient, repeated on every sparse part
    (``masks``: {name: bool mask}) so that a scatter that is wrong on 0.1 % of the pixels cannot
    hide under the norm of the dense part."""
    assert_close(a, b, rel, what=what)
    max_rel = 10 * rel if max_rel is None else max_rel
    e = maxerr(a, b)
    assert e <= max_rel, f"{what}: max-abs err {e:.3e} of max|ref| > {max_rel:.1e}"
    a_ = torch.as_tensor(a).detach().cpu()
    b_ = torch.as_tensor(b).detach().cpu()
    for name, mask in dict(masks).items():
        mask = torch.as_tensor(mask).cpu().reshape(a_.shape)
        assert int(mask.sum()) > 0, f"{what}[{name}]: empty mask"
        assert_close(a_[mask], b_[mask], rel, what=f"{what}[{name}]")
        e = maxerr(a_[mask], b_[mask])
        assert e <= max_rel, f"{what}[{name}]: max-abs err {e:.3e} of max|ref| > {max_rel:.1e}"


def assert_close_or_reference_gap(a, truth, ref32, rel=1e-4, slack=2.0, what=""):
    """``a`` (ours, fp32 arithmetic) against the fp64 truth at ``rel`` — or, where the REFERENCE's own
    fp32 evaluation of the same quantity is further than that from the truth (heavily cancelling
    sums on i.i.d. inputs), no further than ``slack`` times the reference's gap,
    which is measured here, not assumed."""
    e = relerr(a, truth)
    gap = relerr(ref32, truth)
    bound = max(rel, slack * gap)
    assert e <= bound, f"{what}: rel err {e:.3e} > max({rel:.1e}, {slack} x fp32-reference gap {gap:.3e})"
    return e, gap


# ---- the stand-in package of the reference's module layout (bench_support/standin/flowmap): what flowmap_amd.install() patches where the reference is not mounted ----

STANDIN = str(ROOT / "bench_support" / "standin")


def forget_flowmap_modules():  # every module of whatever package called `flowmap` an earlier test imported (the real reference in the build container)
    for name in [n for n in sys.modules if n == "flowmap" or n.startswith("flowmap.")]:
        del sys.modules[name]


@pytest.fixture()
def standin():
    import flowmap_amd

    flowmap_amd.uninstall()
    forget_flowmap_modules()
    sys.path[:0] = [str(ROOT), STANDIN]
    import flowmap

    assert str(Path(flowmap.__file__).resolve()).startswith(STANDIN)
    yield
    flowmap_amd.uninstall()
    forget_flowmap_modules()
    sys.path.remove(STANDIN)
    sys.path.remove(str(ROOT))
