"""Drop-in for flowmap/model/backbone/backbone_explicit_depth.py:19-41 — the explicit-depth backbone: depth and
correspondence-weight logits as free parameters (same cfg, same parameter names ``depth`` / ``weights``, so a ``state_dict`` of the
reference's module loads here and the other way round).

``install()`` registers this class as ``flowmap.model.backbone.BACKBONES["explicit_depth"]`` (backbone/__init__.py:5-8), so that
``get_backbone`` — i.e. an unmodified ``Model`` / ``overfit.py`` — builds it.  The one difference from the reference's module: with lazy
surfaces on, ``forward`` does not run ``sigmoid(sensitivity · weights)`` over the whole (f-1, h, w) tensor (1.1 GB of traffic forward,
1.6 GB backward at 150 x 720p for the 0.1 % of the pixels the Procrustes fit reads); it hands the logits on as a
:class:`~flowmap_amd.model.projection.LazyWeights`, and ``align_surfaces`` applies the sigmoid at the pixels it gathers and writes the
logits' gradient directly.  Anything else that touches the weights (a visualiser, the ablation's ``torch.ones_like``) sees the real values.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional, Tuple

import torch
from torch import nn

from .. import _reference
from ..types import BackboneOutput
from .projection import LazyWeights, lazy_surfaces_enabled

# The container forward() returns: this package's own, or — after install() — the reference's BackboneOutput (a subclass with a plain
# constructor, flowmap_amd/install.py: under jaxtyping's import hook the reference's dataclass checks its fields and would reject a
# LazyWeights, while every `backbone_output: BackboneOutput` annotation of the reference accepts the subclass).
_output_type = BackboneOutput
# Do frame slices of the weights stay lazy (LazyWeights.lazy_slices)?  Off under install(fused_softmin=False): the reference's own
# IntrinsicsSoftmin then reads `backbone_output.weights[:, :1]` with einops (intrinsics_softmin.py:100,120) and must get a tensor.
_lazy_slices = True


def set_output_type(cls=None, lazy_slices: bool = True) -> None:
    global _output_type, _lazy_slices
    _output_type = BackboneOutput if cls is None else cls
    _lazy_slices = bool(lazy_slices)


@dataclass
class BackboneExplicitDepthCfg:
    """flowmap/model/backbone/backbone_explicit_depth.py:12-16"""

    name: Literal["explicit_depth"]
    initial_depth: float
    weight_sensitivity: float


class BackboneExplicitDepth(nn.Module):
    """flowmap/model/backbone/backbone_explicit_depth.py:19-41 (and the attributes of its base, backbone.py:23-35)."""

    def __init__(self, cfg: BackboneExplicitDepthCfg, num_frames: Optional[int], image_shape: Optional[Tuple[int, int]]) -> None:
        super().__init__()
        self.cfg = cfg
        self.num_frames = num_frames
        self.image_shape = image_shape
        self.depth = nn.Parameter(torch.full((num_frames, *image_shape), cfg.initial_depth, dtype=torch.float32))
        self.weights = nn.Parameter(torch.full((num_frames - 1, *image_shape), 0, dtype=torch.float32))

    def forward(self, batch, flows):
        b = batch.videos.shape[0]
        assert b == 1
        ref_cls = _reference.host_twin("BackboneExplicitDepth", self.depth)
        if ref_cls is not None:  # host parameters after install(): the reference's forward on THIS module's parameters (same names)
            return ref_cls.forward(self, batch, flows)
        if lazy_surfaces_enabled():
            # same values, not stored: align_surfaces applies the sigmoid at the points it gathers
            return _output_type(self.depth[None], LazyWeights(self.weights[None], self.cfg.weight_sensitivity, _lazy_slices))
        return _output_type(self.depth[None], (self.cfg.weight_sensitivity * self.weights).sigmoid()[None])
