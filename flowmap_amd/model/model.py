"""Caller glue of the hot path: a mirror of flowmap/model/model.py:41-90 with the two
parameter-only front ends the benchmarks use (BackboneExplicitDepth,
IntrinsicsRegressed).  These modules are NOT part of the accelerated path — they are
the thinnest possible PyTorch producers of its inputs, kept so that bench.py, the tests
and `smoke()` drive the kernels exactly the way ``ModelWrapperOverfit.training_step``
(model_wrapper_overfit.py:51-62) drives the reference.  An unmodified reference
``Model`` works too after ``flowmap_amd.install()``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _ops, _reference
from ..types import ModelOutput
from .backbone import BackboneExplicitDepth, BackboneExplicitDepthCfg  # noqa: F401  (flowmap/model/backbone/backbone_explicit_depth.py)
from .extrinsics_procrustes import ExtrinsicsProcrustes, ExtrinsicsProcrustesCfg
from .projection import sample_image_grid, unproject


_K_CONSTANTS: dict = {}


def _k_constants(image_shape: Tuple[int, int], device) -> Tuple[Tensor, Tensor]:
    """(offset, divisor) with K = offset + f / divisor: the divisor is inf off the two focal
    entries, so f/inf adds an exact 0 to the 0.5 / 1 / 0 entries of the offset."""
    key = (tuple(image_shape), str(device))
    if key not in _K_CONSTANTS:
        h, w = image_shape
        offset = torch.tensor([[0.0, 0.0, 0.5], [0.0, 0.0, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32, device=device)
        divisor = torch.full((3, 3), float("inf"), dtype=torch.float32, device=device)
        divisor[0, 0], divisor[1, 1] = float(w), float(h)
        _K_CONSTANTS[key] = (offset, divisor)
    return _K_CONSTANTS[key]


def focal_lengths_to_intrinsics(focal_lengths: Tensor, image_shape: Tuple[int, int]) -> Tensor:
    """flowmap/model/intrinsics/common.py:6-20: fx = f·√(hw)/w, fy = f·√(hw)/h, cx = cy = 0.5 —
    the same two roundings (multiply, then divide), as one fused multiply-free launch pair
    instead of eye / fill / broadcast-copy / two indexed assignments (and their backward)."""
    h, w = image_shape
    offset, divisor = _k_constants(image_shape, focal_lengths.device)
    scaled = focal_lengths * (h * w) ** 0.5
    return torch.addcdiv(offset, scaled[..., None, None], divisor)


@dataclass
class IntrinsicsRegressedCfg:
    """flowmap/model/intrinsics/intrinsics_regressed.py:16-19"""

    name: Literal["regressed"]
    initial_focal_length: float


class IntrinsicsRegressed(nn.Module):
    """flowmap/model/intrinsics/intrinsics_regressed.py:22-41"""

    def __init__(self, cfg: IntrinsicsRegressedCfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.focal_length = nn.Parameter(torch.full(tuple(), cfg.initial_focal_length, dtype=torch.float32))

    def forward(self, batch, flows, backbone_output, global_step: int) -> Tensor:
        ref_cls = _reference.host_twin("IntrinsicsRegressed", batch)
        if ref_cls is not None:  # host tensors after install(): the reference's forward on THIS module's parameter (same name: focal_length)
            return ref_cls.forward(self, batch, flows, backbone_output, global_step)
        b, f, _, h, w = batch.videos.shape
        # the reference returns focal_lengths_to_intrinsics(...) as an expanded view; every consumer
        # here wants (b,f,3,3) in memory (and its inverse), so one launch writes both
        return _ops.focal_intrinsics(self.focal_length, (b, f), (h, w))


@dataclass
class ModelCfg:
    """flowmap/model/model.py:16-21"""

    backbone: BackboneExplicitDepthCfg
    intrinsics: "IntrinsicsRegressedCfg | IntrinsicsSoftminCfg"
    extrinsics: ExtrinsicsProcrustesCfg
    use_correspondence_weights: bool = True


class Model(nn.Module):
    """flowmap/model/model.py:41-90"""

    def __init__(self, cfg: ModelCfg, num_frames: Optional[int] = None, image_shape: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.cfg = cfg
        self.backbone = BackboneExplicitDepth(cfg.backbone, num_frames, image_shape)
        if cfg.intrinsics.name == "softmin":  # model/intrinsics/__init__.py:16-20 registry, two entries here
            from .intrinsics_softmin import IntrinsicsSoftmin

            self.intrinsics = IntrinsicsSoftmin(cfg.intrinsics)
        else:
            self.intrinsics = IntrinsicsRegressed(cfg.intrinsics)
        self.extrinsics = ExtrinsicsProcrustes(cfg.extrinsics, num_frames)

    def forward(self, batch, flows, global_step: int) -> ModelOutput:
        device = batch.videos.device
        _, _, _, h, w = batch.videos.shape

        # Run the backbone, which provides depths and correspondence weights.
        backbone_out = self.backbone.forward(batch, flows)
        if not self.cfg.use_correspondence_weights:
            backbone_out.weights = torch.ones(backbone_out.weights.shape, dtype=torch.float32, device=device)

        # Compute the intrinsics.
        intrinsics = self.intrinsics.forward(batch, flows, backbone_out, global_step)

        # Use the intrinsics to calculate camera-space surfaces (lazy when enabled).
        xy, _ = sample_image_grid((h, w), device=device)
        surfaces = unproject(xy, backbone_out.depths, intrinsics[:, :, None, None])

        # Finally, compute the extrinsics.
        extrinsics = self.extrinsics.forward(batch, flows, backbone_out, surfaces)

        return ModelOutput(backbone_out.depths, surfaces, intrinsics, extrinsics, backbone_out.weights)
