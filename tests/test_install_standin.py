"""flowmap_amd.install() on the stand-in package (tests/standin: the reference's module LAYOUT with the oracle's arithmetic), where the
reference itself cannot be: on the GPU box.  The rebinding (registries + import-site names) and the HIP library run in ONE process
here: the stand-in's Model and loss factory on cuda:0 after install(), against the golden numbers the real reference produced
(tests/golden/step_*.npz).  The CPU suite runs the same on the host double, and first checks that the stand-in, left alone, reproduces
those goldens (its glue is a faithful layout).  tests/test_install_reference.py does all this with the real package where it is mounted."""

import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
STANDIN = str(ROOT / "tests" / "standin")


def _forget_standin():  # every module of the package called `flowmap`
    for name in [n for n in sys.modules if n == "flowmap" or n.startswith("flowmap.")]:
        del sys.modules[name]


@pytest.fixture()
def standin():
    import flowmap_amd

    flowmap_amd.uninstall()
    _forget_standin()  # (whatever package of that name an earlier test imported — the real reference in the build container)
    sys.path[:0] = [str(ROOT), STANDIN]
    import flowmap

    assert str(Path(flowmap.__file__).resolve()).startswith(STANDIN)
    yield
    flowmap_amd.uninstall()
    _forget_standin()
    sys.path.remove(STANDIN)
    sys.path.remove(str(ROOT))


def _problem(name, with_tracks, dev):
    from conftest import load_golden, t

    import flowmap.loss as ref_loss
    from flowmap.dataset.types import Batch
    from flowmap.flow.flow_predictor import Flows
    from flowmap.loss.loss_flow import LossFlowCfg
    from flowmap.loss.loss_tracking import LossTrackingCfg
    from flowmap.loss.mapping import MappingHuberCfg
    from flowmap.model.backbone import BackboneExplicitDepthCfg
    from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap.model.intrinsics import IntrinsicsRegressedCfg
    from flowmap.model.model import Model, ModelCfg
    from flowmap.tracking.track_predictor import Tracks

    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    f, h, w = depth.shape
    npts = int(g["num_points"])
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", float(g["focal"])),
                           ExtrinsicsProcrustesCfg("procrustes", None if npts < 0 else npts, False), True), num_frames=f, image_shape=(h, w))
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    model = model.to(dev)
    batch = Batch(torch.zeros((1, f, 3, h, w), device=dev))
    flows = Flows(*(t(g[key]).to(dev) for key in ("fwd", "bwd", "fwd_mask", "bwd_mask")))
    cfgs = [LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))]
    tracks = None
    if with_tracks:
        cfgs.append(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
        tracks = [Tracks(t(g[f"trk{i}_xy"]).to(dev), t(g[f"trk{i}_vis"]).to(dev), int(g[f"trk{i}_start"])) for i in range(int(g["n_segments"]))]
    return g, model, batch, flows, tracks, ref_loss.get_losses(cfgs)


def _step_and_compare(g, model, batch, flows, tracks, losses):
    from conftest import assert_close, assert_close_or_reference_gap

    out = model(batch, flows, 0)
    total = sum(fn(batch, flows, tracks, out, 0) for fn in losses)
    total.backward()
    assert_close(total, g["total"], 1e-4, what="total")
    assert_close(out.extrinsics, g["extrinsics"], 1e-4, what="extrinsics")
    assert_close_or_reference_gap(model.backbone.depth.grad, g["f64_g_depth"], g["g_depth"], 1e-4, what="g_depth")
    assert_close_or_reference_gap(model.backbone.weights.grad, g["f64_g_wlogit"], g["g_wlogit"], 1e-4, what="g_wlogit")
    assert_close_or_reference_gap(model.intrinsics.focal_length.grad, g["f64_g_focal"], g["g_focal"], 1e-4, what="g_focal")
    return out


CASES = [("step_iid_flow", False), ("step_scene_flow_tracking", True)]


@pytest.mark.parametrize("name,with_tracks", CASES)
def test_the_standin_left_alone_reproduces_the_reference_goldens(standin, name, with_tracks):
    """Its registries, factories and import-site bindings compose the same step as the reference's (host tensors, nothing installed)."""
    _step_and_compare(*_problem(name, with_tracks, "cpu"))


def _installed_step(name, with_tracks, dev):
    import flowmap.loss as ref_loss
    import flowmap.model.extrinsics as ref_extr
    import flowmap.model.intrinsics as ref_intr
    import flowmap.model.model as ref_model

    import flowmap_amd
    from flowmap_amd import _ops
    from flowmap_amd.model.projection import LazySurfaces

    original_unproject = ref_model.unproject
    flowmap_amd.install()
    try:
        assert ref_loss.LOSSES["flow"] is flowmap_amd.loss.LossFlow and ref_loss.LOSSES["tracking"] is flowmap_amd.loss.LossTracking
        assert ref_extr.EXTRINSICS["procrustes"].__module__.startswith("flowmap_amd") and ref_intr.INTRINSICS["regressed"].__module__.startswith("flowmap_amd")
        assert ref_model.unproject is not original_unproject  # the name model.py bound at import now dispatches
        before = dict(_ops.counters)
        g, model, batch, flows, tracks, losses = _problem(name, with_tracks, dev)
        assert type(losses[0]) is flowmap_amd.loss.LossFlow and type(model.extrinsics).__module__.startswith("flowmap_amd")
        out = _step_and_compare(g, model, batch, flows, tracks, losses)
        assert isinstance(out.surfaces, LazySurfaces)  # the stand-in Model's unproject went lazy: the fused kernels consumed depth directly
        # a second and third step: the flow loss packs its constant inputs once the same flows come back — which only the fused path does
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            _step_and_compare(g, model, batch, flows, tracks, losses)
        moved = {key for key, value in _ops.counters.items() if value != before.get(key, 0)}
        assert "flow_packs" in moved, moved
    finally:
        flowmap_amd.uninstall()
    assert ref_loss.LOSSES["flow"].__module__.startswith("flowmap.") and ref_model.unproject is original_unproject


@pytest.mark.parametrize("name,with_tracks", CASES)
def test_install_on_the_standin_with_the_host_double(standin, name, with_tracks):
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    try:
        _installed_step(name, with_tracks, "cpu")
    finally:
        _lib.set_library_for_testing(None)


@pytest.mark.gpu
@pytest.mark.parametrize("name,with_tracks", CASES)
def test_install_on_the_standin_runs_the_hip_library(standin, name, with_tracks):
    """The rebinding and the HIP kernels in one process, on cuda:0."""
    from flowmap_amd import _lib

    assert not _lib.using_test_double()
    _installed_step(name, with_tracks, "cuda:0")
