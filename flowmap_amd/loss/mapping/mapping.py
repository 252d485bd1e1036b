"""Drop-in for flowmap/loss/mapping/mapping.py."""

from __future__ import annotations

from abc import ABC
from typing import Generic, Tuple, TypeVar

import torch
from torch import Tensor, nn

from ... import _ops, _reference
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

    def _host(self, *tensors):
        """Host tensors after install(): the reference's mapping class of the same kind (flowmap_amd/_reference.py), else None."""
        ref_cls = _reference.host_twin("Mapping:" + self.kind, *tensors)
        if ref_cls is None:
            return None
        twin = self.__dict__.get("_fm_host_twin")
        if twin is None or type(twin) is not ref_cls:
            twin = self.__dict__["_fm_host_twin"] = ref_cls(self.cfg)
        return twin

    def forward(self, a: Tensor, b: Tensor, image_shape: Tuple[int, int]) -> Tensor:
        twin = self._host(a, b)
        if twin is not None:
            return twin.forward(a, b, image_shape)
        check_device(a, b)
        shape = torch.broadcast_shapes(a.shape, b.shape)
        ax, ay = aspect_correction(image_shape)
        out = _ops.RobustMapping.apply(
            a.expand(shape).reshape(-1, 2), b.expand(shape).reshape(-1, 2), _ops.MAPPING_KINDS[self.kind], self.delta, ax, ay
        )
        return out.reshape(shape[:-1])

    def forward_undistorted(self, delta: Tensor) -> Tensor:
        twin = self._host(delta)
        if twin is not None:
            return twin.forward_undistorted(delta)
        check_device(delta)
        flat = delta.reshape(-1, 2)
        out = _ops.RobustMapping.apply(flat, torch.zeros_like(flat), _ops.MAPPING_KINDS[self.kind], self.delta, 1.0, 1.0)
        return out.reshape(delta.shape[:-1])


# --------------------------------------------------------------------------------------
# The three robust kernels.  In the reference each lives in its own module
# (mapping_huber.py:18-34, mapping_l1.py:15-20, mapping_l2.py:15-21) and overrides
# ``forward_undistorted`` with torch ops; here a subclass only selects the ``kind`` the
# fused HIP kernel is instantiated for, so they fit in one place.
# --------------------------------------------------------------------------------------
from dataclasses import dataclass  # noqa: E402
from typing import Literal  # noqa: E402


@dataclass
class MappingHuberCfg:
    name: Literal["huber"]
    delta: float  # knee of the Huber function, in aspect-corrected normalised units


@dataclass
class MappingL1Cfg:
    name: Literal["l1"]


@dataclass
class MappingL2Cfg:
    name: Literal["l2"]


class MappingHuber(Mapping[MappingHuberCfg]):
    """huber_loss(‖r‖, 0, delta) / delta: quadratic below the knee, ‖r‖ − delta/2 above, so the
    gradient magnitude in the linear region equals L1's."""

    kind = "huber"


class MappingL1(Mapping[MappingL1Cfg]):
    """‖r‖₂ (sub-gradient 0 at r = 0)."""

    kind = "l1"


class MappingL2(Mapping[MappingL2Cfg]):
    """½‖r‖² (the ½ matches huber's quadratic branch)."""

    kind = "l2"
