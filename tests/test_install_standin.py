"""flowmap_amd.install() on the stand-in package (bench_support/standin: the reference's module LAYOUT with the oracle's arithmetic), where the
reference itself cannot be: on the GPU box.  The rebinding (registries + import-site names) and the HIP library run in ONE process
here: the stand-in's Model and loss factory on cuda:0 after install(), against the golden numbers the real reference produced
(tests/golden/step_*.npz).  The CPU suite runs the same on the host double, and first checks that the stand-in, left alone, reproduces
those goldens (its glue is a faithful layout).  tests/test_install_reference.py does all this with the real package where it is mounted."""

import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


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
    from flowmap_amd.model.projection import LazySurfaces, LazyWeights

    original_unproject = ref_model.unproject
    flowmap_amd.install()
    try:
        assert ref_loss.LOSSES["flow"] is flowmap_amd.loss.LossFlow and ref_loss.LOSSES["tracking"] is flowmap_amd.loss.LossTracking
        assert ref_extr.EXTRINSICS["procrustes"].__module__.startswith("flowmap_amd") and ref_intr.INTRINSICS["regressed"].__module__.startswith("flowmap_amd")
        assert ref_model.unproject is not original_unproject  # the name model.py bound at import now dispatches
        before = dict(_ops.counters)
        g, model, batch, flows, tracks, losses = _problem(name, with_tracks, dev)
        assert type(losses[0]) is flowmap_amd.loss.LossFlow and type(model.extrinsics).__module__.startswith("flowmap_amd")
        assert type(model.backbone).__module__ == "flowmap_amd.model.backbone"  # get_backbone -> BACKBONES["explicit_depth"], rebound
        out = _step_and_compare(g, model, batch, flows, tracks, losses)
        assert isinstance(out.surfaces, LazySurfaces)  # the stand-in Model's unproject went lazy: the fused kernels consumed depth directly
        assert isinstance(out.backward_correspondence_weights, LazyWeights)  # no sigmoid over the (f-1, h, w) logits: applied at the gathered pixels
        # a second and third step: the flow loss packs its constant inputs once the same flows come back — which only the fused path does
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            _step_and_compare(g, model, batch, flows, tracks, losses)
        moved = {key for key, value in _ops.counters.items() if value != before.get(key, 0)}
        assert "flow_packs" in moved, moved
    finally:
        flowmap_amd.uninstall()
    assert ref_loss.LOSSES["flow"].__module__.startswith("flowmap.") and ref_model.unproject is original_unproject
    import flowmap.model.backbone as ref_backbone

    assert ref_backbone.BACKBONES["explicit_depth"].__module__.startswith("flowmap.")


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


def _no_stray_torch_kernels(dev):
    """cases.case_step_torch_ops on the INSTALLED path: the stand-in's Model (get_backbone / get_intrinsics / get_extrinsics) and get_losses after
    install().  Beside the library's kernels a flow-only step launches nothing — in particular no sigmoid over the weight logits (the rebound
    backbone hands them on as LazyWeights) and no ones_like fill for backward(); flow + tracking adds the sum of the two losses and the two sums
    autograd forms where two consumers meet."""
    from torch.utils._python_dispatch import TorchDispatchMode

    import flowmap_amd
    from flowmap_amd import _ops

    quiet = ("view", "unsqueeze", "squeeze", "detach", "empty", "select.int", "slice", "expand", "reshape", "alias", "permute", "as_strided",
             "t.default", "transpose", "_local_scalar_dense", "lift_fresh", "unbind", "split", "flowmap_amd", "profiler", "_to_copy", "clone")
    flowmap_amd.install()
    try:
        for name, with_tracks, allowed in (("step_iid_flow", False, {}), ("step_scene_flow_tracking", True, {"aten.add.Tensor": 3})):
            g, model, batch, flows, tracks, losses = _problem(name, with_tracks, dev)

            def step():
                model.zero_grad(set_to_none=True)
                out = model(batch, flows, 0)
                total = losses[0](batch, flows, tracks, out, 0)
                for fn in losses[1:]:
                    total = total + fn(batch, flows, tracks, out, 0)
                assert type(total) is _ops.RootLoss
                total.backward()

            for _ in range(3):  # plans, packed inputs, arenas exist from the third step on
                step()
            seen = {}

            class Trace(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    op = str(func)
                    if not any(q in op for q in quiet):
                        seen[op] = seen.get(op, 0) + 1
                    return func(*args, **(kwargs or {}))

            with Trace():
                step()
            assert seen == allowed, (name, seen)
    finally:
        flowmap_amd.uninstall()


def test_an_installed_step_launches_no_stray_torch_kernels_host_double(standin):
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    try:
        _no_stray_torch_kernels("cpu")
    finally:
        _lib.set_library_for_testing(None)


@pytest.mark.gpu
def test_an_installed_step_launches_no_stray_torch_kernels(standin):
    _no_stray_torch_kernels("cuda:0")


@pytest.mark.parametrize("name,with_tracks", CASES)
def test_host_tensors_after_install_go_back_to_the_packages_own_code(standin, name, with_tracks):
    """The ONE hand-over of the product path (flowmap_amd/_reference.py; SURVEY.md §8b, BASELINE.json configs[0]): with the real C ABI library
    selected (no test double), install() leaves host-tensor calls to the functions and classes it replaced — here the stand-in's own — so the
    step reproduces the goldens on the CPU, the rebound backbone included; without install() the same host tensors are refused
    (test_abi.py::test_no_cpu_fallback)."""
    import flowmap_amd
    from flowmap_amd import _lib, _reference

    _lib.set_library_for_testing(None)
    flowmap_amd.install()
    try:
        before = _reference.counters["host_calls"]
        g, model, batch, flows, tracks, losses = _problem(name, with_tracks, "cpu")
        assert type(model.backbone).__module__ == "flowmap_amd.model.backbone"
        out = _step_and_compare(g, model, batch, flows, tracks, losses)
        assert torch.is_tensor(out.surfaces) and torch.is_tensor(out.backward_correspondence_weights)  # the package's own unproject / sigmoid
        assert _reference.counters["host_calls"] > before
    finally:
        flowmap_amd.uninstall()
