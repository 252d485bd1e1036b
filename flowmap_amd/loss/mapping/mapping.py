"""Drop-in for flowmap/loss/mapping/mapping.py."""

from __future__ import annotations

from abc import ABC
from typing import Generic, Tuple, TypeVar

import torch
from torch import Tensor, nn

from ... import _ops
from ..._lib import check_device


def aspect_correction(image_shape: Tuple[int, int]) -> Tuple[float, float]:
    """The two factors of fix_aspect_ratio (flowmap/loss/mapping/mapping.py:17-23)."""
    h, w = image_shape
    scale = (h * w) ** 0.5
    return w / scale, h / scale


def fix_aspect_ratio(points: Tensor, image_shape: Tuple[int, int]) -> Tensor:
    """flowmap/loss/mapping/mapping.py:9-24 (a two-element scale; plain torch)."""
    ax, ay = aspect_correction(image_shape)
    return points * torch.tensor((ax, ay), dtype=points.dtype, device=points.device)


T = TypeVar("T")


class Mapping(nn.Module, ABC, Generic[T]):
    """flowmap/loss/mapping/mapping.py:30-51.  ``forward`` runs one fused HIP kernel
    (aspect fix of both operands, difference, robust map); subclasses only name the
    kernel variant.  ``forward_undistorted`` keeps the reference's hook for callers that
    already hold an aspect-corrected delta."""

    kind: str = "huber"

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @property
    def delta(self) -> float:
        return float(getattr(self.cfg, "delta", 0.0) or 0.0)

    def forward(self, a: Tensor, b: Tensor, image_shape: Tuple[int, int]) -> Tensor:
        check_device(a, b)
        shape = torch.broadcast_shapes(a.shape, b.shape)
        ax, ay = aspect_correction(image_shape)
        out = _ops.RobustMapping.apply(
            a.expand(shape).reshape(-1, 2), b.expand(shape).reshape(-1, 2), _ops.MAPPING_KINDS[self.kind], self.delta, ax, ay
        )
        return out.reshape(shape[:-1])

    def forward_undistorted(self, delta: Tensor) -> Tensor:
        check_device(delta)
        flat = delta.reshape(-1, 2)
        out = _ops.RobustMapping.apply(flat, torch.zeros_like(flat), _ops.MAPPING_KINDS[self.kind], self.delta, 1.0, 1.0)
        return out.reshape(delta.shape[:-1])
