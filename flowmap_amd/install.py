"""Rebind an importable reference ``flowmap`` package to the HIP implementations.

The reference dispatches by registry (``LOSSES[cfg.name]``, ``EXTRINSICS[cfg.name]``)
and by names bound at import time (``from ..model.projection import unproject``), so
both have to be patched.  After ``install()``, an unmodified
``python -m flowmap.overfit`` runs its hot path on the kernels of this package.
"""

from __future__ import annotations

import dataclasses
import importlib
from typing import Dict, List, Tuple

_saved: List[Tuple[object, str, object]] = []

# module -> names it bound with ``from ..model.projection import ...`` (file:line)
_IMPORT_SITES: Dict[str, Tuple[str, ...]] = {
    "flowmap.model.model": ("sample_image_grid", "unproject"),  # model/model.py:13
    "flowmap.model.extrinsics.extrinsics_procrustes": ("align_surfaces",),  # extrinsics_procrustes.py:11
    "flowmap.model.extrinsics.extrinsics_regressed": ("get_extrinsics",),  # extrinsics_regressed.py:12
    "flowmap.model.intrinsics.intrinsics_softmin": (  # intrinsics_softmin.py:13-18
        "align_surfaces", "compute_backward_flow", "sample_image_grid", "unproject",
    ),
    "flowmap.loss.loss_flow": ("compute_backward_flow", "compute_forward_flow", "sample_image_grid"),  # loss_flow.py:10-14
    "flowmap.loss.loss_tracking": ("compute_track_flow",),  # loss_tracking.py:11
    "flowmap.visualization.visualizer_summary": ("compute_backward_flow", "compute_forward_flow"),  # visualizer_summary.py:13
    "flowmap.model.projection": ("align_rigid",),  # projection.py:8
}


_MISSING = object()


def _set(obj, name, value):
    # classes: keep the raw descriptor (staticmethod objects survive the round trip)
    old = obj.__dict__.get(name, _MISSING) if isinstance(obj, type) else getattr(obj, name, None)
    _saved.append((obj, name, old))
    setattr(obj, name, value)


_CROPPING_NAMES = ("resize_batch", "crop_and_resize_batch_for_model", "crop_and_resize_batch_for_flow")
# modules that bind those names at import (overfit.py:29-32, model_wrapper_pretrain.py:12-16)
_CROPPING_SITES = ("flowmap.overfit", "flowmap.model.model_wrapper_pretrain")


def install(lazy_surfaces: bool = True, fused_softmin: bool = True, flow_postprocess: bool = True, fused_adam: bool = True,
            cropping: bool = True, fused_regressed: bool = True, lazy_backbone: bool = True, graph: bool | None = None,
            options: dict | None = None) -> None:
    """Patch the reference in place.  ``lazy_surfaces=True`` additionally lets
    ``Model.forward``'s ``unproject`` hand a LazySurfaces to the fused consumers;
    ``fused_softmin=True`` registers the fused candidate sweep as INTRINSICS["softmin"]
    (flowmap/model/intrinsics/__init__.py:6-10); ``flow_postprocess=True`` rebinds
    ``FlowPredictor.compute_consistency_mask`` / ``compute_bidirectional_flow``
    (flowmap/flow/flow_predictor.py:59-102) on the reference base class, so every concrete
    predictor (RAFT, GMFlow) inherits the fused post-processing; ``fused_adam=True`` makes
    ``ModelWrapperOverfit.configure_optimizers`` (model_wrapper_overfit.py:104-105) build
    ``flowmap_amd.FusedAdam`` (skipped when lightning is not importable); ``cropping=True`` rebinds
    ``resize_batch`` / ``crop_and_resize_batch_for_model`` / ``_for_flow`` (flowmap/misc/cropping.py)
    to the one-pass resize+crop, which uploads a host batch once and prepares both videos in HBM;
    ``fused_regressed=True`` registers INTRINSICS["regressed"] = ``flowmap_amd``'s IntrinsicsRegressed
    (same cfg and parameter name; K and K⁻¹ for all frames from one launch, one-launch backward);
    ``lazy_backbone=True`` registers BACKBONES["explicit_depth"] (flowmap/model/backbone/__init__.py:5-8) = ``flowmap_amd``'s
    BackboneExplicitDepth (same cfg, same parameter names ``depth`` / ``weights``: state_dict-compatible), whose forward hands the weight
    logits on unevaluated when ``lazy_surfaces`` is on (flowmap_amd/model/backbone.py) — the step an unmodified ``overfit.py`` then runs is
    the step ``bench.py`` times; ``graph=True`` (default: the environment's ``FLOWMAP_AMD_GRAPH=1``, else off) rebinds
    ``ModelWrapperOverfit.training_step`` (model_wrapper_overfit.py:51-73) to a step that replays the model's forward + the losses and the
    loss's ``backward()`` as two hipGraphs while the optimisation's host-side control flow stands still, and runs the reference's own method
    otherwise (flowmap_amd/training.py: what the reference's default ≈ 180×240 resolution needs, where the eager step is host-bound; videos
    whose depth maps exceed 128 MB are HBM-bound and keep the eager step); ``options``: run-time switches of the package
    (``flowmap_amd.config.Options`` field names: ``tap_exchange``, ``tap_exchange_min_bytes``, ``unit_seed``, ``packed_inputs``, …) set for the
    rest of the process — the one place, with ``flowmap_amd.config``, where such switches live."""
    if options:
        from . import config

        config.configure(**dict(options))
    from . import loss as our_loss
    from .loss import mapping as our_mapping
    from .model import procrustes as our_procrustes
    from .model import projection as our_projection
    from .model.extrinsics_procrustes import ExtrinsicsProcrustes

    if _saved:
        uninstall()

    # Import every site module BEFORE patching anything, so that modules imported for the
    # first time here bind (and uninstall() later restores) the reference's own functions.
    sites = {}
    for mod_name in _IMPORT_SITES:
        try:
            sites[mod_name] = importlib.import_module(mod_name)
        except Exception:  # optional dependency of that module is absent (e.g. flow_vis_torch)
            pass
    ref_projection = importlib.import_module("flowmap.model.projection")
    ref_procrustes = importlib.import_module("flowmap.model.procrustes")
    ref_loss = importlib.import_module("flowmap.loss")
    ref_mapping = importlib.import_module("flowmap.loss.mapping")
    ref_extr = importlib.import_module("flowmap.model.extrinsics")
    # Every name is bound to a DISPATCHER: tensors on the GPU go to this package's kernels, tensors on the host to the reference's own
    # function, which is right here (flowmap_amd/_reference.py; SURVEY.md §8b: "CPU tensors -> the reference-equivalent torch path",
    # BASELINE.json configs[0]).  The classes in the registries dispatch inside their forward() the same way.
    from . import _reference

    _reference.twins.clear()
    public = [n for n in dir(our_projection) if not n.startswith("_") and hasattr(ref_projection, n) and callable(getattr(our_projection, n))]
    originals = {name: getattr(ref_projection, name) for name in public}
    originals["align_rigid"] = ref_procrustes.align_rigid
    bound = {name: _reference.dispatching(name, getattr(our_projection, name), originals[name]) for name in public}
    bound["align_rigid"] = _reference.dispatching("align_rigid", our_procrustes.align_rigid, originals["align_rigid"])
    # IntrinsicsSoftmin reshapes / indexes the surfaces with einops right away (intrinsics_softmin.py:104-109): the materialising variant
    bound_unproject_dense = _reference.dispatching("unproject", our_projection.unproject_dense, originals["unproject"])
    for name in public:
        _set(ref_projection, name, bound[name])
    _set(ref_procrustes, "align_rigid", bound["align_rigid"])

    for mod_name, names in _IMPORT_SITES.items():
        mod = sites.get(mod_name)
        if mod is None:
            continue
        for name in names:
            value = bound_unproject_dense if (name == "unproject" and mod_name.endswith("intrinsics_softmin")) else bound[name]
            _set(mod, name, value)

    _reference.twins.update({"LossFlow": ref_loss.LOSSES["flow"], "LossTracking": ref_loss.LOSSES["tracking"],
                             "ExtrinsicsProcrustes": ref_extr.EXTRINSICS["procrustes"]})
    for kind, cls in ref_mapping.MAPPINGS.items():
        _reference.twins["Mapping:" + kind] = cls
    _set(ref_loss, "LOSSES", {**ref_loss.LOSSES, "flow": our_loss.LossFlow, "tracking": our_loss.LossTracking})
    _set(ref_mapping, "MAPPINGS", {**ref_mapping.MAPPINGS, **our_mapping.MAPPINGS})
    _set(ref_extr, "EXTRINSICS", {**ref_extr.EXTRINSICS, "procrustes": ExtrinsicsProcrustes})
    if fused_softmin:
        from .model.intrinsics_softmin import IntrinsicsSoftmin

        ref_intr = importlib.import_module("flowmap.model.intrinsics")
        _reference.twins["IntrinsicsSoftmin"] = ref_intr.INTRINSICS["softmin"]
        _set(ref_intr, "INTRINSICS", {**ref_intr.INTRINSICS, "softmin": IntrinsicsSoftmin})

    if fused_regressed:
        from .model.model import IntrinsicsRegressed

        ref_intr = importlib.import_module("flowmap.model.intrinsics")
        _reference.twins["IntrinsicsRegressed"] = ref_intr.INTRINSICS["regressed"]
        _set(ref_intr, "INTRINSICS", {**ref_intr.INTRINSICS, "regressed": IntrinsicsRegressed})

    if lazy_backbone:
        from .model import backbone as our_backbone

        ref_backbone = importlib.import_module("flowmap.model.backbone")
        _reference.twins["BackboneExplicitDepth"] = ref_backbone.BACKBONES["explicit_depth"]
        _set(ref_backbone, "BACKBONES", {**ref_backbone.BACKBONES, "explicit_depth": our_backbone.BackboneExplicitDepth})
        try:  # backbone/backbone.py:13-17 (the package's __init__ does not re-export it)
            ref_output = importlib.import_module("flowmap.model.backbone.backbone").BackboneOutput
        except Exception:
            ref_output = ref_backbone.BackboneOutput
        our_backbone.set_output_type(_plain_constructor_subclass(ref_output), lazy_slices=fused_softmin)

    # The reference's factories are annotated with its abstract bases (`get_backbone(...) -> Backbone`, backbone/__init__.py:13-18; likewise
    # get_extrinsics / get_intrinsics / get_losses / get_mapping), which beartype checks under the import hook of overfit.py:15-19: the classes
    # registered above are declared virtual subclasses of those bases (ABCMeta.register; all four bases are ABCs).
    for mod_name, base_name, ours in _virtual_bases(our_loss, our_mapping, ExtrinsicsProcrustes, fused_softmin, fused_regressed, lazy_backbone):
        try:
            base = getattr(importlib.import_module(mod_name), base_name)
            for cls in ours:
                base.register(cls)
        except Exception:  # a stand-in package without that base class, or a base that is not an ABC: nothing to declare
            pass

    if flow_postprocess:
        from . import _ops

        ref_fp = importlib.import_module("flowmap.flow.flow_predictor")

        ref_bidirectional = ref_fp.FlowPredictor.compute_bidirectional_flow

        def compute_bidirectional_flow(self, batch, flow_shape):
            if _reference.on_host(batch):  # host tensors: the reference's own method
                return ref_bidirectional(self, batch, flow_shape)
            videos = batch.videos
            forward, forward_mask = _ops.flow_postprocess(videos, self.forward(videos), flow_shape, reverse=False)
            backward, backward_mask = _ops.flow_postprocess(videos, self.forward(videos.flip(dims=(1,))), flow_shape, reverse=True)
            return ref_fp.Flows(forward, backward, forward_mask, backward_mask)

        ref_consistency = ref_fp.FlowPredictor.__dict__["compute_consistency_mask"]
        ref_consistency = ref_consistency.__func__ if isinstance(ref_consistency, staticmethod) else ref_consistency
        _set(ref_fp.FlowPredictor, "compute_consistency_mask", staticmethod(_reference.dispatching("compute_consistency_mask", _ops.consistency_mask, ref_consistency)))
        _set(ref_fp.FlowPredictor, "compute_bidirectional_flow", compute_bidirectional_flow)

    if fused_adam:
        try:
            ref_wrapper = importlib.import_module("flowmap.model.model_wrapper_overfit")
        except Exception:  # lightning / hydra are not installed here; nothing to rebind
            ref_wrapper = None
        if ref_wrapper is not None:
            from .optim import FusedAdam

            ref_configure = ref_wrapper.ModelWrapperOverfit.configure_optimizers

            def configure_optimizers(self):
                if not _lib_is_double() and any(p.device.type == "cpu" for p in self.parameters()):
                    return ref_configure(self)  # host parameters: torch.optim.Adam, as the reference builds it (model_wrapper_overfit.py:104-105)
                return FusedAdam(self.parameters(), lr=self.cfg.lr)

            _set(ref_wrapper.ModelWrapperOverfit, "configure_optimizers", configure_optimizers)

    if graph is None:
        import os

        graph = os.environ.get("FLOWMAP_AMD_GRAPH", "0").lower() not in ("", "0", "false", "off")
    if graph:
        try:
            ref_wrapper = importlib.import_module("flowmap.model.model_wrapper_overfit")
        except Exception as exc:  # lightning / hydra are not installed here; nothing to rebind — said aloud: the caller asked for it
            import warnings

            warnings.warn(f"flowmap_amd.install(graph=True): flowmap.model.model_wrapper_overfit is not importable ({type(exc).__name__}: {exc}); "
                          "training_step stays as it is")
            ref_wrapper = None
        if ref_wrapper is not None:
            from .training import make_training_step

            eager_step = ref_wrapper.ModelWrapperOverfit.__dict__.get("training_step", ref_wrapper.ModelWrapperOverfit.training_step)
            _set(ref_wrapper.ModelWrapperOverfit, "training_step", make_training_step(eager_step))

    if cropping:
        from .misc import cropping as our_cropping

        ref_cropping = importlib.import_module("flowmap.misc.cropping")
        crop_sites = []
        for mod_name in _CROPPING_SITES:  # import first: they must bind the reference's functions
            try:
                crop_sites.append(importlib.import_module(mod_name))
            except Exception:  # hydra / lightning are not installed
                pass
        import torch

        def without_gpu(ours, ref):  # (the fused resize + crop takes a HOST batch and uploads it: what decides is whether there is a GPU at all)
            import functools

            @functools.wraps(ours)
            def call(*args, **kwargs):
                if not torch.cuda.is_available() and not _lib_is_double():
                    return ref(*args, **kwargs)
                return ours(*args, **kwargs)

            return call

        for name in _CROPPING_NAMES:
            value = without_gpu(getattr(our_cropping, name), getattr(ref_cropping, name))
            _set(ref_cropping, name, value)
            for mod in crop_sites:
                if hasattr(mod, name):
                    _set(mod, name, value)

    if lazy_surfaces:
        # flowmap/overfit.py:15-19 imports the package under jaxtyping's import hook, which type-checks
        # dataclass fields on construction: ModelOutput(surfaces=<LazySurfaces>) would be rejected
        # (model/model.py:24-30,83-89).  Model.forward therefore builds a subclass with the same fields
        # and a plain constructor; every `model_output: ModelOutput` annotation still accepts it.
        ref_model = importlib.import_module("flowmap.model.model")
        _set(ref_model, "ModelOutput", _plain_constructor_subclass(ref_model.ModelOutput))

    our_projection.set_lazy_surfaces(lazy_surfaces)


def _virtual_bases(our_loss, our_mapping, extrinsics_cls, fused_softmin, fused_regressed, lazy_backbone):
    """(module, abstract base, this package's classes registered under it)."""
    out = [("flowmap.loss.loss", "Loss", (our_loss.LossFlow, our_loss.LossTracking)),
           ("flowmap.loss.mapping.mapping", "Mapping", tuple(our_mapping.MAPPINGS.values())),
           ("flowmap.model.extrinsics.extrinsics", "Extrinsics", (extrinsics_cls,))]
    intrinsics = []
    if fused_softmin:
        from .model.intrinsics_softmin import IntrinsicsSoftmin

        intrinsics.append(IntrinsicsSoftmin)
    if fused_regressed:
        from .model.model import IntrinsicsRegressed

        intrinsics.append(IntrinsicsRegressed)
    if intrinsics:
        out.append(("flowmap.model.intrinsics.intrinsics", "Intrinsics", tuple(intrinsics)))
    if lazy_backbone:
        from .model.backbone import BackboneExplicitDepth

        out.append(("flowmap.model.backbone.backbone", "Backbone", (BackboneExplicitDepth,)))
    return out


def _lib_is_double() -> bool:
    from . import _lib

    return _lib.using_test_double()


def _plain_constructor_subclass(base):
    names = tuple(f.name for f in dataclasses.fields(base))

    def __init__(self, *args, **kwargs):
        values = dict(zip(names, args))
        values.update(kwargs)
        if set(values) != set(names) or len(args) > len(names):
            raise TypeError(f"{base.__name__} takes the fields {names}")
        for name in names:
            setattr(self, name, values[name])

    return type(base.__name__, (base,), {"__init__": __init__, "__module__": base.__module__, "__doc__": base.__doc__})


def uninstall() -> None:
    from .model import projection as our_projection

    from . import _reference

    _reference.twins.clear()
    while _saved:
        obj, name, old = _saved.pop()
        if old is _MISSING:
            delattr(obj, name)
        else:
            setattr(obj, name, old)
    our_projection.set_lazy_surfaces(False)
    from .model import backbone as our_backbone

    our_backbone.set_output_type(None)
