"""flowmap_amd.install() against the REAL reference package (importable only in the build
container, /root/reference; skipped elsewhere): the reference's own Model, registries and
Loss classes, unmodified, must run their hot path on our kernels (host double here) and
reproduce the golden step results the unpatched reference produced."""

import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("FLOWMAP_REFERENCE", "/root/reference"))

pytestmark = pytest.mark.skipif(not (REF / "flowmap" / "model" / "projection.py").exists(), reason="reference not mounted")


@pytest.fixture()
def patched_reference():
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    import flowmap_amd
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.install()
    yield
    flowmap_amd.uninstall()
    _lib.set_library_for_testing(None)
    for p in added:
        sys.path.remove(p)


@pytest.mark.parametrize("name,with_tracks", [("step_iid_flow", False), ("step_scene_flow_tracking", True)])
def test_unmodified_reference_model_runs_on_our_kernels(patched_reference, name, with_tracks):
    from conftest import assert_close, load_golden, t

    import flowmap.loss as ref_loss
    import flowmap.model.extrinsics as ref_extr
    from flowmap.dataset.types import Batch
    from flowmap.flow.flow_predictor import Flows
    from flowmap.loss.loss_flow import LossFlowCfg
    from flowmap.loss.loss_tracking import LossTrackingCfg
    from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
    from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
    from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg
    from flowmap.model.model import Model, ModelCfg
    from flowmap.tracking.track_predictor import Tracks

    import flowmap_amd
    from flowmap_amd.model.projection import LazySurfaces

    # the registries now hand out our classes
    assert ref_loss.LOSSES["flow"] is flowmap_amd.loss.LossFlow
    assert ref_extr.EXTRINSICS["procrustes"].__module__.startswith("flowmap_amd")

    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    f, h, w = depth.shape
    npts = int(g["num_points"])
    cfg = ModelCfg(
        BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
        IntrinsicsRegressedCfg("regressed", float(g["focal"])),
        ExtrinsicsProcrustesCfg("procrustes", None if npts < 0 else npts, False),
        True,
    )
    model = Model(cfg, num_frames=f, image_shape=(h, w))  # the reference's Model, unmodified
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
    flows = Flows(t(g["fwd"]), t(g["bwd"]), t(g["fwd_mask"]), t(g["bwd_mask"]))
    cfgs = [LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))]
    tracks = None
    if with_tracks:
        cfgs.append(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
        tracks = [Tracks(t(g[f"trk{i}_xy"]), t(g[f"trk{i}_vis"]), int(g[f"trk{i}_start"])) for i in range(int(g["n_segments"]))]
    losses = ref_loss.get_losses(cfgs)  # the reference's factory -> our Loss classes
    # get_backbone (model/backbone/__init__.py:13-18) built this package's backbone: same parameter names, and a virtual subclass of the
    # reference's abstract base (what `-> Backbone` is checked against under the import hook)
    from flowmap.model.backbone.backbone import Backbone, BackboneOutput
    from flowmap_amd.model.projection import LazyWeights

    assert type(model.backbone).__module__ == "flowmap_amd.model.backbone" and isinstance(model.backbone, Backbone)
    assert [n for n, _ in model.backbone.named_parameters()] == ["depth", "weights"]
    out = model(batch, flows, 0)
    assert isinstance(out.surfaces, LazySurfaces)  # Model.forward's unproject went lazy
    assert isinstance(out.backward_correspondence_weights, LazyWeights)  # ... and no sigmoid ran over the (f-1, h, w) logits
    assert isinstance(model.backbone.forward(batch, flows), BackboneOutput)
    total = sum(fn(batch, flows, tracks, out, 0) for fn in losses)
    total.backward()
    assert_close(total, g["total"], 1e-4, what="total")
    assert_close(out.extrinsics, g["extrinsics"], 1e-4, what="extrinsics")
    # gradients against the reference's own fp64 evaluation of the same step (the fixture's f64_* keys), at 1e-4 or twice the gap of its fp32 gradients
    from conftest import assert_close_or_reference_gap

    assert_close_or_reference_gap(model.backbone.depth.grad, g["f64_g_depth"], g["g_depth"], 1e-4, what="g_depth")
    assert_close_or_reference_gap(model.backbone.weights.grad, g["f64_g_wlogit"], g["g_wlogit"], 1e-4, what="g_wlogit")
    assert_close_or_reference_gap(model.intrinsics.focal_length.grad, g["f64_g_focal"], g["g_focal"], 1e-4, what="g_focal")


def test_rebound_backbone_is_state_dict_compatible_with_the_reference(patched_reference):
    """BACKBONES["explicit_depth"] after install() (backbone/__init__.py:5-8 -> flowmap_amd/model/backbone.py): a checkpoint of the reference's
    module loads into it and the other way round (same parameter names and shapes); uninstall() puts the reference's class back."""
    import flowmap.model.backbone as ref_backbone
    from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg

    import flowmap_amd
    from flowmap_amd import _reference

    cfg = BackboneExplicitDepthCfg("explicit_depth", 1.5, 100.0)
    ours = ref_backbone.get_backbone(cfg, 4, (6, 8))
    assert type(ours).__module__ == "flowmap_amd.model.backbone"
    assert float(ours.depth[0, 0, 0]) == 1.5 and ours.num_frames == 4 and ours.image_shape == (6, 8)
    theirs = _reference.twins["BackboneExplicitDepth"](cfg, 4, (6, 8))
    with torch.no_grad():
        theirs.depth.mul_(2.0)
        theirs.weights.add_(0.25)
    ours.load_state_dict(theirs.state_dict())
    assert torch.equal(ours.depth, theirs.depth) and torch.equal(ours.weights, theirs.weights)
    theirs.load_state_dict(ours.state_dict())
    assert set(ours.state_dict()) == set(theirs.state_dict()) == {"depth", "weights"}
    flowmap_amd.uninstall()
    assert ref_backbone.BACKBONES["explicit_depth"].__module__ == "flowmap.model.backbone.backbone_explicit_depth"
    flowmap_amd.install()  # (the fixture's teardown uninstalls)


def test_reference_softmin_intrinsics_under_install():
    """IntrinsicsSoftmin (the reference's DEFAULT intrinsics for the first 1000 steps,
    intrinsics_softmin.py:63-141) calls unproject / align_surfaces / compute_backward_flow
    with batch 60 and 1-D grids.  Unpatched reference vs the same code after install()."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    try:
        from conftest import assert_close
        from flowmap.dataset.types import Batch
        from flowmap.flow.flow_predictor import Flows
        from flowmap.model.backbone.backbone import BackboneOutput
        from flowmap.model.intrinsics.intrinsics_softmin import IntrinsicsSoftmin, IntrinsicsSoftminCfg

        import flowmap_amd
        from flowmap_amd import _lib
        from helpers import build_host_sim
        from oracle import flowmap_oracle as orc

        f, h, w = 3, 12, 16
        depth, wlogit, fl = orc.synth_iid(f, h, w, seed=8)
        flows = Flows(fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)
        batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
        cfg = IntrinsicsSoftminCfg("softmin", 64, 0.5, 2.0, 12, None)

        def run():
            d = depth[None].clone().requires_grad_(True)
            wt = (100 * wlogit).sigmoid()[None].clone().requires_grad_(True)
            torch.manual_seed(0)  # the module draws torch.randperm
            k = IntrinsicsSoftmin(cfg).forward(batch, flows, BackboneOutput(d, wt), 0)
            (k * torch.arange(9.0).reshape(3, 3)).sum().backward()
            return k.detach(), d.grad, wt.grad

        k_ref, gd_ref, gw_ref = run()
        # the truth for the gradients: the oracle's fp64 evaluation of the same sweep (pinned to the reference module's own fp64 evaluation by
        # tests/test_oracle_golden.py::test_softmin_intrinsics) on the pixels the module draws; gates: 1e-4 or twice the reference's fp32 gap
        from conftest import assert_close_or_reference_gap

        torch.manual_seed(0)
        drawn = torch.randperm(h * w)[: cfg.num_procrustes_points]
        d64 = depth[None].double().requires_grad_(True)
        w64 = (100 * wlogit).sigmoid()[None].double().requires_grad_(True)
        cand = IntrinsicsSoftmin(cfg).focal_length_candidates.double()
        k64 = orc.softmin_intrinsics(d64, w64, fl.backward.double(), cand, drawn, (h, w))
        (k64[:, None].expand(1, f, 3, 3) * torch.arange(9.0, dtype=torch.float64).reshape(3, 3)).sum().backward()
        assert_close(k_ref, k64[:, None].expand(1, f, 3, 3), 1e-5, what="the oracle's sweep is the reference's")
        _lib.set_library_for_testing(build_host_sim())
        flowmap_amd.install(fused_softmin=False)  # the reference's class on our function-level kernels
        try:
            k_ours, gd_ours, gw_ours = run()
        finally:
            flowmap_amd.uninstall()
        assert_close(k_ours, k_ref, 1e-4, what="softmin intrinsics")
        assert_close_or_reference_gap(gd_ours, d64.grad, gd_ref, 1e-4, what="g_depth")
        assert_close_or_reference_gap(gw_ours, w64.grad, gw_ref, 1e-4, what="g_weights")

        # the fused candidate sweep registered by install(): same numbers, no 60x repeats
        import flowmap.model.intrinsics as ref_intr

        flowmap_amd.install()
        try:
            fused_cls = ref_intr.INTRINSICS["softmin"]
            assert fused_cls.__module__.startswith("flowmap_amd")

            def run_fused(lazy_weights):
                d = depth[None].clone().requires_grad_(True)
                logits = wlogit[None].clone().requires_grad_(True)
                if lazy_weights:
                    wt = flowmap_amd.model.projection.LazyWeights(logits, 100.0)
                else:
                    wt = (100 * logits).sigmoid()
                torch.manual_seed(0)
                module = fused_cls(cfg)
                # same pixels as the reference run: its sampling (our own draws an equivalent
                # random subset more cheaply, from a different random stream)
                module._draw_indices = lambda count, device: torch.randperm(count, device=device)[: cfg.num_procrustes_points]
                k = module.forward(batch, flows, BackboneOutput(d, wt), 0)
                (k * torch.arange(9.0).reshape(3, 3)).sum().backward()
                return k.detach(), d.grad, logits.grad

            sig = (100 * wlogit).sigmoid()[None]
            gl_ref = gw_ref * 100 * sig * (1 - sig)  # chain rule through the backbone's sigmoid
            sig64 = (100 * wlogit.double()).sigmoid()[None]
            gl64 = w64.grad * 100 * sig64 * (1 - sig64)
            for lazy_weights in (False, True):
                k_f, gd_f, gl_f = run_fused(lazy_weights)
                assert_close(k_f, k_ref, 1e-4, what="fused softmin intrinsics")
                assert_close_or_reference_gap(gd_f, d64.grad, gd_ref, 1e-4, what="fused g_depth")
                assert_close_or_reference_gap(gl_f, gl64, gl_ref, 1e-4, what="fused g_logits")
        finally:
            flowmap_amd.uninstall()
            _lib.set_library_for_testing(None)
    finally:
        for p in added:
            sys.path.remove(p)


def test_reference_softmin_model_under_install_without_the_fused_sweep():
    """install(fused_softmin=False) keeps the REFERENCE's IntrinsicsSoftmin while the backbone is the rebound lazy one: the module reads
    `backbone_output.weights[:, :1]` with einops (intrinsics_softmin.py:100,120), so frame slices of the lazy weights must come back as tensors
    there.  The reference's own Model (explicit depth -> softmin -> Procrustes) + flow loss, unpatched vs installed that way: same pixels drawn
    (same seed, same torch.randperm), same loss and gradients."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    try:
        from conftest import assert_close
        from flowmap.dataset.types import Batch
        from flowmap.flow.flow_predictor import Flows
        from flowmap.loss import get_losses
        from flowmap.loss.loss_flow import LossFlowCfg
        from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
        from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
        from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
        from flowmap.model.intrinsics.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg
        from flowmap.model.model import Model, ModelCfg

        import flowmap_amd
        from flowmap_amd import _lib
        from flowmap_amd.model.projection import LazyWeights
        from helpers import build_host_sim
        from oracle import flowmap_oracle as orc

        f, h, w = 4, 16, 20
        scene = orc.synth_scene(f, h, w, seed=3)
        depth, fl = scene["depth_init"], scene["flows"]
        wlogit = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(3))
        flows = Flows(fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)
        batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
        cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsSoftminCfg("softmin", 64, 0.5, 2.0, 12, RegressionCfg(5, 2)),
                       ExtrinsicsProcrustesCfg("procrustes", 50, False), True)

        def run():
            model = Model(cfg, num_frames=f, image_shape=(h, w))
            model.backbone.depth.data = depth.clone()
            model.backbone.weights.data = wlogit.clone()
            (loss_fn,) = get_losses([LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))])
            torch.manual_seed(0)  # IntrinsicsSoftmin draws torch.randperm
            out = model(batch, flows, 0)
            loss = loss_fn(batch, flows, None, out, 0)
            loss.backward()
            # (out.extrinsics: under install() a LazyExtrinsics of this flow-only step — evaluated here, while the host double is still in place)
            out.extrinsics = torch.as_tensor(out.extrinsics).detach().clone()
            return type(model.backbone), type(model.intrinsics), out, loss.detach(), model.backbone.depth.grad, model.backbone.weights.grad

        _, ref_intr_cls, out_ref, loss_ref, gd_ref, gw_ref = run()
        _lib.set_library_for_testing(build_host_sim())
        flowmap_amd.install(fused_softmin=False)
        try:
            backbone_cls, intr_cls, out, loss, gd, gw = run()
            # the lazy backbone is in, the reference's sweep stayed, and a frame slice of its weights is a tensor (all of them: still lazy)
            assert backbone_cls.__module__.startswith("flowmap_amd") and intr_cls is ref_intr_cls
            sliced = LazyWeights(wlogit[None], 100.0, lazy_slices=False)[:, :1]
            assert torch.is_tensor(sliced) and sliced.shape == (1, 1, h, w)
            assert_close(sliced, (100 * wlogit[None, :1]).sigmoid(), 1e-7, what="the slice's values")
        finally:
            flowmap_amd.uninstall()
            _lib.set_library_for_testing(None)
        assert isinstance(LazyWeights(wlogit[None], 100.0)[:, :1], LazyWeights)  # the default (this package's own sweep reads the logits)
        assert_close(out.intrinsics, out_ref.intrinsics, 1e-5, what="softmin intrinsics")
        assert_close(out.extrinsics, out_ref.extrinsics, 1e-4, what="extrinsics")
        assert_close(loss, loss_ref, 1e-4, what="flow loss")
        assert_close(gd, gd_ref, 2e-4, what="g_depth")
        assert_close(gw, gw_ref, 2e-4, abs_=1e-9, what="g_weight_logits")
    finally:
        for p in added:
            sys.path.remove(p)


def test_reference_flow_predictor_under_install(patched_reference):
    """A predictor subclassing the REFERENCE's FlowPredictor inherits the fused
    post-processing after install(), returns the reference's Flows type, and reproduces what
    the unpatched reference computed (golden); uninstall() restores the static method."""
    from conftest import assert_close, load_golden, t

    import flowmap.flow.flow_predictor as ref_fp
    from flowmap.dataset.types import Batch

    import flowmap_amd
    from oracle import flowmap_oracle as orc

    g = load_golden("fn_flow_preprocess")
    videos, raw = t(g["a_videos"]), t(g["a_raw"])
    shape = tuple(int(x) for x in g["a_shape"])

    class StandIn(ref_fp.FlowPredictor):
        def forward(self, v):
            return raw if torch.equal(v, videos) else orc.standin_predictor(v)

    assert ref_fp.FlowPredictor.compute_bidirectional_flow.__module__.startswith("flowmap_amd")
    flows = StandIn(None).compute_bidirectional_flow(Batch(videos, None, None, None), shape)
    assert isinstance(flows, ref_fp.Flows)
    for name in ("forward", "backward", "forward_mask", "backward_mask"):
        assert_close(getattr(flows, name), g[f"a_{name}"], 2e-5, what=name)
    assert_close(StandIn(None).compute_consistency_mask(videos, raw), g["a_mask_full"], 2e-5, what="mask (static, via instance)")

    flowmap_amd.uninstall()
    assert ref_fp.FlowPredictor.compute_bidirectional_flow.__module__ == "flowmap.flow.flow_predictor"
    assert isinstance(ref_fp.FlowPredictor.__dict__["compute_consistency_mask"], staticmethod)
    again = StandIn(None).compute_bidirectional_flow(Batch(videos, None, None, None), shape)  # reference code, CPU
    assert_close(again.forward_mask, g["a_forward_mask"], 1e-6, what="unpatched")


def test_model_output_survives_a_field_checking_constructor():
    """Under jaxtyping's import hook (flowmap/overfit.py:15-19) the reference's dataclasses check
    their fields on construction.  Simulated here by a constructor that insists on tensors: after
    install() Model.forward still returns a ModelOutput (a subclass with a plain constructor) that
    carries the LazySurfaces; uninstall() restores the class."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    try:
        import flowmap.model.model as ref_model
        from flowmap.dataset.types import Batch
        from flowmap.flow.flow_predictor import Flows
        from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
        from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
        from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg

        import flowmap_amd
        from flowmap_amd import _lib
        from flowmap_amd.model.projection import LazySurfaces
        from helpers import build_host_sim
        from oracle import flowmap_oracle as orc

        original_cls, original_init = ref_model.ModelOutput, ref_model.ModelOutput.__init__

        def checking_init(self, depths, surfaces, intrinsics, extrinsics, backward_correspondence_weights):
            for value in (depths, surfaces, intrinsics, extrinsics, backward_correspondence_weights):
                if not isinstance(value, torch.Tensor):
                    raise TypeError("field is not a Tensor")
            original_init(self, depths, surfaces, intrinsics, extrinsics, backward_correspondence_weights)

        ref_model.ModelOutput.__init__ = checking_init
        _lib.set_library_for_testing(build_host_sim())
        try:
            flowmap_amd.install()
            f, h, w = 4, 12, 16
            depth, _, of = orc.synth_iid(f, h, w, seed=3)
            cfg = ref_model.ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                                     ExtrinsicsProcrustesCfg("procrustes", 50, False), True)
            model = ref_model.Model(cfg, num_frames=f, image_shape=(h, w))
            model.backbone.depth.data = depth
            flows = Flows(of.forward, of.backward, of.forward_mask, of.backward_mask)
            out = model(Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"]), flows, 0)
            assert isinstance(out, original_cls) and type(out) is not original_cls and type(out).__name__ == "ModelOutput"
            assert isinstance(out.surfaces, LazySurfaces) and out.extrinsics.shape == (1, f, 4, 4)
            assert [fl.name for fl in __import__("dataclasses").fields(out)][-1] == "backward_correspondence_weights"
            with pytest.raises(TypeError):
                type(out)(out.depths, out.surfaces)  # still needs every field
            flowmap_amd.uninstall()
            assert ref_model.ModelOutput is original_cls
            with pytest.raises(TypeError, match="not a Tensor"):
                original_cls(out.depths, out.surfaces, out.intrinsics, out.extrinsics, out.backward_correspondence_weights)
        finally:
            flowmap_amd.uninstall()
            ref_model.ModelOutput.__init__ = original_init
            _lib.set_library_for_testing(None)
    finally:
        for p in added:
            sys.path.remove(p)


def test_reference_regressed_intrinsics_under_install(patched_reference):
    """install() registers the one-launch IntrinsicsRegressed: same parameter name, K bit-identical
    to the reference class, same focal-length gradient; uninstall() restores the registry."""
    from conftest import assert_close

    import flowmap.model.intrinsics as ref_intr
    from flowmap.dataset.types import Batch
    from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressed as RefRegressed
    from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg

    import flowmap_amd

    cfg = IntrinsicsRegressedCfg("regressed", 0.85)
    ours = ref_intr.get_intrinsics(cfg)
    assert type(ours).__module__.startswith("flowmap_amd")
    theirs = RefRegressed(cfg)
    assert list(ours.state_dict()) == list(theirs.state_dict()) == ["focal_length"]
    batch = Batch(torch.zeros((2, 5, 3, 30, 44)), None, None, None)
    k_ours, k_theirs = ours(batch, None, None, 0), theirs(batch, None, None, 0)
    assert k_ours.shape == k_theirs.shape and torch.equal(k_ours, k_theirs)
    cot = torch.randn(k_ours.shape, generator=torch.Generator().manual_seed(0))
    (k_ours * cot).sum().backward()
    (k_theirs * cot).sum().backward()
    assert_close(ours.focal_length.grad, theirs.focal_length.grad, 2e-6, what="g_focal")
    flowmap_amd.uninstall()
    assert ref_intr.INTRINSICS["regressed"] is RefRegressed


def test_reference_cropping_under_install(patched_reference):
    """After install() the reference's cropping module runs the one-pass resize+crop on the
    reference's own Batch type and reproduces the unpatched outputs (golden); uninstall() restores."""
    from cases import CROPPING_CASES
    from conftest import assert_close, load_golden, t

    import flowmap.misc.cropping as ref_cropping
    from flowmap.dataset.types import Batch

    import flowmap_amd

    g = load_golden("fn_cropping")
    image_shape, mult, patch = CROPPING_CASES["b"]
    cfg = ref_cropping.CroppingCfg(image_shape, mult, patch)
    batch = Batch(t(g["b_videos"]), torch.arange(2)[None], ["s"], ["d"], None, t(g["b_intrinsics"]))
    assert ref_cropping.crop_and_resize_batch_for_model.__module__ == "flowmap_amd.misc.cropping"
    model_batch, pre_crop = ref_cropping.crop_and_resize_batch_for_model(batch, cfg)
    flow_batch = ref_cropping.crop_and_resize_batch_for_flow(batch, cfg)
    assert isinstance(model_batch, Batch) and model_batch.scenes == ["s"]
    assert tuple(pre_crop) == tuple(int(x) for x in g["b_pre_crop"])
    assert_close(model_batch.videos, g["b_model_videos"], 2e-6, what="model videos")
    assert_close(flow_batch.videos, g["b_flow_videos"], 2e-6, what="flow videos")
    assert_close(flow_batch.intrinsics, g["b_flow_intrinsics"], 1e-6, what="flow intrinsics")

    flowmap_amd.uninstall()
    assert ref_cropping.crop_and_resize_batch_for_model.__module__ == "flowmap.misc.cropping"
    again, _ = ref_cropping.crop_and_resize_batch_for_model(batch, cfg)
    assert_close(again.videos, g["b_model_videos"], 1e-6, what="unpatched")


def test_reference_overfit_loop_under_install_tracks_the_unpatched_loop():
    """The loop body of ModelWrapperOverfit.training_step + Adam (model_wrapper_overfit.py:51-62,
    104-105) for a few steps: the unmodified reference (CPU, torch.optim.Adam) against the same
    reference Model / losses after install() with flowmap_amd.FusedAdam — parameters stay together."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    try:
        from conftest import assert_close, load_golden, t

        import flowmap.loss as ref_loss
        from flowmap.dataset.types import Batch
        from flowmap.flow.flow_predictor import Flows
        from flowmap.loss.loss_flow import LossFlowCfg
        from flowmap.loss.loss_tracking import LossTrackingCfg
        from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
        from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
        from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
        from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg
        from flowmap.model.model import Model, ModelCfg
        from flowmap.tracking.track_predictor import Tracks

        import flowmap_amd
        from flowmap_amd import _lib
        from helpers import build_host_sim

        g = load_golden("step_scene_flow_tracking")
        depth, wlogit = t(g["depth"]), t(g["wlogit"])
        f, h, w = depth.shape
        flows = Flows(t(g["fwd"]), t(g["bwd"]), t(g["fwd_mask"]), t(g["bwd_mask"]))
        tracks = [Tracks(t(g[f"trk{i}_xy"]), t(g[f"trk{i}_vis"]), int(g[f"trk{i}_start"])) for i in range(int(g["n_segments"]))]
        batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
        cfgs = [LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)), LossTrackingCfg(2, 100.0, "tracking", MappingHuberCfg("huber", 0.01))]

        def loop(make_optimizer):
            cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", float(g["focal"])),
                           ExtrinsicsProcrustesCfg("procrustes", None if int(g["num_points"]) < 0 else int(g["num_points"]), False), True)
            model = Model(cfg, num_frames=f, image_shape=(h, w))
            model.backbone.depth.data = depth.clone()
            model.backbone.weights.data = wlogit.clone()
            losses = ref_loss.get_losses(cfgs)
            opt = make_optimizer(model.parameters())
            history = []
            for step in range(6):  # the tracking loss switches on at step 2 (enable_after)
                opt.zero_grad()
                out = model(batch, flows, step)
                total = sum(fn(batch, flows, tracks, out, step) for fn in losses)
                total.backward()
                opt.step()
                history.append(float(total.detach()))
            return history, model

        hist_ref, model_ref = loop(lambda params: torch.optim.Adam(params, lr=1e-3))
        _lib.set_library_for_testing(build_host_sim())
        flowmap_amd.install()
        try:
            hist_ours, model_ours = loop(lambda params: flowmap_amd.FusedAdam(params, lr=1e-3))
        finally:
            flowmap_amd.uninstall()
            _lib.set_library_for_testing(None)
        assert hist_ref[2] != hist_ref[1]  # the tracking loss came in at step 2
        assert_close(torch.tensor(hist_ours), torch.tensor(hist_ref), 2e-4, what="loss history")
        assert_close(model_ours.backbone.depth, model_ref.backbone.depth, 1e-5, what="depth after 6 steps")
        assert_close(model_ours.backbone.weights, model_ref.backbone.weights, 1e-4, abs_=1e-6, what="weight logits after 6 steps")
        assert_close(model_ours.intrinsics.focal_length, model_ref.intrinsics.focal_length, 1e-5, what="focal length after 6 steps")
    finally:
        for p in added:
            sys.path.remove(p)


@pytest.mark.parametrize(
    "f,h,w,points,mapping,use_weights,randomize",
    [
        (4, 10, 12, 60, "huber", True, True),  # randomize_points: torch.randint indices, duplicates included
        (4, 10, 12, 30, "huber", True, False),
        (3, 9, 14, None, "l1", True, False),  # every pixel a correspondence
        (5, 8, 12, 40, "l2", False, False),  # use_correspondence_weights: false -> constant ones
    ],
)
def test_reference_configurations_under_install(f, h, w, points, mapping, use_weights, randomize):
    """The reference's Model + flow loss in several configurations, unpatched vs after install()."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    try:
        from conftest import assert_close

        import flowmap.loss as ref_loss
        from flowmap.dataset.types import Batch
        from flowmap.flow.flow_predictor import Flows
        from flowmap.loss.loss_flow import LossFlowCfg
        from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
        from flowmap.loss.mapping.mapping_l1 import MappingL1Cfg
        from flowmap.loss.mapping.mapping_l2 import MappingL2Cfg
        from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
        from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
        from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg
        from flowmap.model.model import Model, ModelCfg

        import flowmap_amd
        from flowmap_amd import _lib
        from helpers import build_host_sim

        mapping_cfg = {"huber": MappingHuberCfg("huber", 0.01), "l1": MappingL1Cfg("l1"), "l2": MappingL2Cfg("l2")}[mapping]

        def run():
            g = torch.Generator().manual_seed(f * h + w)
            depth = 1.0 + 0.3 * torch.rand((f, h, w), generator=g)
            logits = 0.01 * torch.randn((f - 1, h, w), generator=g)
            flows = Flows(0.02 * torch.randn((1, f - 1, h, w, 2), generator=g), 0.02 * torch.randn((1, f - 1, h, w, 2), generator=g),
                          torch.rand((1, f - 1, h, w), generator=g), torch.rand((1, f - 1, h, w), generator=g))
            cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.8),
                           ExtrinsicsProcrustesCfg("procrustes", points, randomize), use_weights)
            model = Model(cfg, num_frames=f, image_shape=(h, w))
            model.backbone.depth.data = depth.clone()
            model.backbone.weights.data = logits.clone()
            batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
            losses = ref_loss.get_losses([LossFlowCfg(0, 1000.0, "flow", mapping_cfg)])
            torch.manual_seed(3)  # randomize_points draws torch.randint
            out = model(batch, flows, 0)
            total = sum(fn(batch, flows, None, out, 0) for fn in losses)
            total.backward()
            return (total.detach(), out.extrinsics.detach(), model.backbone.depth.grad, model.backbone.weights.grad,
                    model.intrinsics.focal_length.grad)

        ref = run()
        _lib.set_library_for_testing(build_host_sim())
        flowmap_amd.install()
        try:
            ours = run()
        finally:
            flowmap_amd.uninstall()
            _lib.set_library_for_testing(None)
        scale = abs(float(ref[0]))
        assert_close(ours[0], ref[0], 1e-4, what="total")
        assert_close(ours[1], ref[1], 1e-4, what="extrinsics")
        assert_close(ours[2], ref[2], 1e-3, abs_=1e-4 * scale, what="g_depth")
        assert_close(ours[4], ref[4], 1e-2, abs_=1e-4 * scale, what="g_focal")
        if ref[3] is None:
            assert ours[3] is None or float(ours[3].abs().max()) == 0.0
        else:
            assert_close(ours[3], ref[3], 2e-3, abs_=1e-6, what="g_weight_logits")
    finally:
        for p in added:
            sys.path.remove(p)


# ---- host (CPU) tensors after install(): the reference's own functions, saved by install() ----


@pytest.fixture()
def patched_reference_real_library():
    """install() with the REAL C ABI library selected (no test double): host tensors cannot run on it."""
    sys.dont_write_bytecode = True
    added = [str(ROOT / "oracle" / "refstubs"), str(REF)]
    sys.path[:0] = added
    import flowmap_amd
    from flowmap_amd import _lib

    _lib.set_library_for_testing(None)
    flowmap_amd.install()
    yield
    flowmap_amd.uninstall()
    for p in added:
        sys.path.remove(p)


@pytest.mark.parametrize("name,with_tracks", [("step_iid_flow", False), ("step_scene_flow_tracking", True)])
def test_host_tensors_after_install_take_the_reference_functions(patched_reference_real_library, name, with_tracks):
    """BASELINE.json configs[0] ("PyTorch CPU path via overfit.py: plumbing, no GPU") after flowmap_amd.install(): the unmodified
    reference Model + get_losses on HOST tensors — every rebound name and registry class hands the call to the original install()
    saved, so the step reproduces the golden numbers the unpatched reference produced, to rounding (it IS the reference's arithmetic);
    nothing touches the HIP library (which cannot take host tensors) or oracle/."""
    from conftest import assert_close, load_golden, t

    import flowmap.loss as ref_loss
    from flowmap.dataset.types import Batch
    from flowmap.flow.flow_predictor import Flows
    from flowmap.loss.loss_flow import LossFlowCfg
    from flowmap.loss.loss_tracking import LossTrackingCfg
    from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
    from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
    from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg
    from flowmap.model.model import Model, ModelCfg
    from flowmap.tracking.track_predictor import Tracks

    import flowmap_amd
    from flowmap_amd import _lib, _reference
    from flowmap_amd.model.projection import LazySurfaces

    assert not _lib.using_test_double()
    assert ref_loss.LOSSES["flow"] is flowmap_amd.loss.LossFlow  # the registries hold OUR classes; they dispatch per call
    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    f, h, w = depth.shape
    npts = int(g["num_points"])
    cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", float(g["focal"])),
                   ExtrinsicsProcrustesCfg("procrustes", None if npts < 0 else npts, False), True)
    model = Model(cfg, num_frames=f, image_shape=(h, w))
    assert type(model.intrinsics).__module__.startswith("flowmap_amd") and type(model.extrinsics).__module__.startswith("flowmap_amd")
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
    flows = Flows(t(g["fwd"]), t(g["bwd"]), t(g["fwd_mask"]), t(g["bwd_mask"]))
    cfgs = [LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))]
    tracks = None
    if with_tracks:
        cfgs.append(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
        tracks = [Tracks(t(g[f"trk{i}_xy"]), t(g[f"trk{i}_vis"]), int(g[f"trk{i}_start"])) for i in range(int(g["n_segments"]))]
    losses = ref_loss.get_losses(cfgs)
    before = _reference.counters["host_calls"]
    out = model(batch, flows, 0)
    assert torch.is_tensor(out.surfaces) and not isinstance(out.surfaces, LazySurfaces)  # the reference's unproject: a real tensor
    total = sum(fn(batch, flows, tracks, out, 0) for fn in losses)
    total.backward()
    assert _reference.counters["host_calls"] > before
    # the reference's own arithmetic on this torch build: the fixtures to fp32 rounding
    assert_close(total, g["total"], 2e-6, what="total")
    assert_close(out.extrinsics, g["extrinsics"], 2e-6, what="extrinsics")
    assert_close(model.backbone.depth.grad, g["g_depth"], 2e-5, what="g_depth")
    assert_close(model.backbone.weights.grad, g["g_wlogit"], 2e-5, what="g_wlogit")
    assert_close(model.intrinsics.focal_length.grad, g["g_focal"], 1e-4, abs_=1e-6 * abs(float(g["total"])), what="g_focal")


def test_host_softmin_and_functions_after_install(patched_reference_real_library):
    """The softmin sweep (reference forward run on OUR module's state) and the rebound projection functions with host tensors."""
    from conftest import assert_close, load_golden, t

    import flowmap.model.intrinsics as ref_intr
    import flowmap.model.projection as ref_projection
    from flowmap.dataset.types import Batch
    from flowmap.flow.flow_predictor import Flows
    from flowmap.model.backbone.backbone import BackboneOutput
    from flowmap.model.intrinsics.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg

    g = load_golden("fn_softmin")  # the unpatched reference's IntrinsicsSoftmin.forward (oracle/make_golden.py: gold_softmin)
    cls = ref_intr.INTRINSICS["softmin"]
    assert cls.__module__.startswith("flowmap_amd")
    depth, weights, bwd = t(g["depth"])[None], t(g["weights"]), t(g["bwd"])
    b, f, h, w = depth.shape
    indices = t(g["indices"]).to(torch.int64)
    rest = torch.tensor(sorted(set(range(h * w)) - set(indices.tolist())), dtype=torch.int64)
    perm = torch.cat([indices, rest])  # a permutation whose first P entries are the recorded sample
    module = cls(IntrinsicsSoftminCfg("softmin", indices.numel(), 0.5, 2.0, int(g["candidates"].shape[0]), RegressionCfg(5, 2)))
    module.train()
    batch = Batch(torch.zeros((b, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
    flows = Flows(torch.zeros_like(bwd), bwd, torch.ones(bwd.shape[:-1]), torch.ones(bwd.shape[:-1]))
    real_randperm = torch.randperm
    torch.randperm = lambda *a, **k: perm.clone()
    try:
        k = module(batch, flows, BackboneOutput(depth, weights), 4)  # inside the window: the reference's forward appends to OUR module's list
    finally:
        torch.randperm = real_randperm
    assert_close(k[0, 0], g["intrinsics"], 2e-6, what="softmin intrinsics on the host")
    assert len(module.window) == 1
    k5 = module(batch, flows, BackboneOutput(depth, weights), 5)  # the hand-over: regressed focal length = the window's mean
    assert_close(k5[0, 0, 0, 0] * w / (h * w) ** 0.5, module.window[0], 1e-6, what="hand-over focal length")
    # a rebound function with host tensors: the reference's own
    xy, _ = ref_projection.sample_image_grid((4, 6), "cpu")
    assert xy.shape == (4, 6, 2) and ref_projection.unproject._fm_reference is not ref_projection.unproject


def _real_overfit_wrapper(name="step_scene_flow_tracking", enable_tracking_after=0, with_truth=False):
    """The reference's OWN ModelWrapperOverfit (model_wrapper_overfit.py:25-43) around its own Model and get_losses — under oracle/refstubs'
    import-only LightningModule (a torch module with `log` and `global_step`: this image has no lightning)."""
    from conftest import load_golden, t

    import flowmap.loss as ref_loss
    from flowmap.dataset.types import Batch
    from flowmap.flow.flow_predictor import Flows
    from flowmap.loss.loss_flow import LossFlowCfg
    from flowmap.loss.loss_tracking import LossTrackingCfg
    from flowmap.loss.mapping.mapping_huber import MappingHuberCfg
    from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg
    from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg
    from flowmap.model.model import Model, ModelCfg
    from flowmap.model.model_wrapper_overfit import ModelWrapperOverfit, ModelWrapperOverfitCfg
    from flowmap.tracking.track_predictor import Tracks

    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    f, h, w = depth.shape
    npts = int(g["num_points"])
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", float(g["focal"])),
                           ExtrinsicsProcrustesCfg("procrustes", None if npts < 0 else npts, False), True), num_frames=f, image_shape=(h, w))
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    truth = None
    if with_truth:  # a dataset with ground-truth intrinsics: training_step also logs the focal-length errors (model_wrapper_overfit.py:65-73)
        truth = torch.eye(3)[None, None].repeat(1, f, 1, 1)
        truth[..., 0, 0], truth[..., 1, 1], truth[..., 0, 2], truth[..., 1, 2] = 0.9, 1.2, 0.5, 0.5
    batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"], None, truth)
    flows = Flows(t(g["fwd"]), t(g["bwd"]), t(g["fwd_mask"]), t(g["bwd_mask"]))
    tracks = [Tracks(t(g[f"trk{i}_xy"]), t(g[f"trk{i}_vis"]), int(g[f"trk{i}_start"])) for i in range(int(g["n_segments"]))]
    losses = ref_loss.get_losses([LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)),
                                  LossTrackingCfg(enable_tracking_after, 100.0, "tracking", MappingHuberCfg("huber", 0.01))])
    wrapper = ModelWrapperOverfit(ModelWrapperOverfitCfg(1e-3, 32), model, batch, flows, tracks, losses, [])
    wrapper.train()
    return g, wrapper


def test_install_graph_on_the_references_own_overfit_wrapper(patched_reference):
    """install(graph=True) on the REAL flowmap.model.model_wrapper_overfit: training_step is rebound around the reference's own method (whose source
    file is the reference's), configure_optimizers builds FusedAdam, and a trainer's order of calls — training_step, loss / 1, zero_grad, backward,
    step — runs the reference's method on this package's kernels (host double: nothing is captured here), logging under the reference's names."""
    import flowmap.model.model_wrapper_overfit as ref_wrapper
    from conftest import assert_close

    import flowmap_amd

    original = ref_wrapper.ModelWrapperOverfit.training_step
    assert str(REF) in original.__code__.co_filename
    flowmap_amd.install(graph=True)
    rebound = ref_wrapper.ModelWrapperOverfit.training_step
    assert rebound is not original and rebound.__wrapped__ is original
    g, wrapper = _real_overfit_wrapper(enable_tracking_after=1)
    optimizer = wrapper.configure_optimizers()
    assert type(optimizer).__module__ == "flowmap_amd.optim"
    history = []
    for _ in range(4):
        loss = wrapper.training_step(None) / 1
        history.append(float(loss.detach()))
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        optimizer.step()
        wrapper.global_step += 1
    state = wrapper.__dict__["_fm_graphed_training"]
    assert state.captures == 0 and state.disabled is None  # (the host double: no hipGraph can exist; the reference's method ran every time)
    assert set(wrapper.logged) == {"train/loss/flow", "train/loss/tracking"}
    assert history[2] < history[1]  # (step 0 has no tracking term yet: enable_after = 1)
    assert abs(float(wrapper.logged["train/loss/flow"]) + float(wrapper.logged["train/loss/tracking"]) - history[-1]) <= 1e-6 * abs(history[-1])
    flowmap_amd.uninstall()
    assert ref_wrapper.ModelWrapperOverfit.training_step is original
    flowmap_amd.install()  # (the fixture's teardown uninstalls)
    _, fresh = _real_overfit_wrapper()
    assert_close(fresh.training_step(None), g["total"], 1e-4, what="the reference's training_step on our kernels vs the unpatched reference's total")


def test_the_captured_body_is_the_references_training_step(patched_reference):
    """What install(graph=True) captures is a restatement of training_step's body (flowmap_amd/training.py: GraphedTraining.forward + .log):
    against the reference's OWN method on the same wrapper — same total, same logged names and values, with and without ground-truth
    intrinsics in the batch (the focal-length errors, model_wrapper_overfit.py:65-73)."""
    import flowmap.model.model_wrapper_overfit as ref_wrapper

    from flowmap_amd import training

    real = ref_wrapper.ModelWrapperOverfit.training_step
    for with_truth in (False, True):
        _, wrapper = _real_overfit_wrapper(with_truth=with_truth)
        theirs = real(wrapper, None)
        logged_by_the_reference = {k: float(v) for k, v in wrapper.logged.items()}
        wrapper.logged.clear()
        state = training.GraphedTraining(real)
        state.total, state.values, state.errors = state.forward(wrapper)
        state.log(wrapper)
        assert float(state.total.detach()) == float(theirs.detach())
        assert {k: float(v) for k, v in wrapper.logged.items()} == logged_by_the_reference
        assert set(logged_by_the_reference) == {"train/loss/flow", "train/loss/tracking"} | ({"train/intrinsics/fx_error", "train/intrinsics/fy_error"} if with_truth else set())
        assert state.signature(wrapper) is not None  # the reference's wrapper around rebound parts has a phase: on the GPU it would be captured
