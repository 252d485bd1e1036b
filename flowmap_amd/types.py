"""Plain containers mirroring the reference's hot-path input/output types.

  Flows        flowmap/flow/flow_predictor.py:16-21
  Tracks       flowmap/tracking/track_predictor.py:13-20
  Batch        flowmap/dataset/types.py:12-19      (only .videos shape/device is read)
  ModelOutput  flowmap/model/model.py:24-30
  BackboneOutput  flowmap/model/backbone/backbone.py:14-17

The reference's own dataclasses are accepted everywhere these are (duck typing): the
drop-in never checks the class, only the attribute names.
"""

from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Any, Optional

from torch import Tensor


class _Movable:
    def to(self, device):
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.to(device) if hasattr(v, "to") else v
        return type(self)(**kw)


@dataclass
class Flows(_Movable):
    forward: Tensor  # (batch, pair, H, W, 2)
    backward: Tensor  # (batch, pair, H, W, 2)
    forward_mask: Tensor  # (batch, pair, H, W)
    backward_mask: Tensor  # (batch, pair, H, W)


@dataclass
class Tracks(_Movable):
    xy: Tensor  # (batch, frame, point, 2)
    visibility: Tensor  # (batch, frame, point) bool
    start_frame: int


@dataclass
class Batch(_Movable):
    videos: Tensor  # (batch, frame, 3, H, W)
    indices: Optional[Tensor] = None
    scenes: Optional[list] = None
    datasets: Optional[list] = None
    extrinsics: Optional[Tensor] = None
    intrinsics: Optional[Tensor] = None


@dataclass
class BackboneOutput:
    depths: Tensor  # (batch, frame, H, W)
    weights: Tensor  # (batch, frame-1, H, W)


@dataclass
class ModelOutput:
    depths: Tensor  # (batch, frame, H, W)
    surfaces: Any  # (batch, frame, H, W, 3) Tensor, or LazySurfaces
    intrinsics: Tensor  # (batch, frame, 3, 3)
    extrinsics: Tensor  # (batch, frame, 4, 4)
    backward_correspondence_weights: Tensor  # (batch, frame-1, H, W)
