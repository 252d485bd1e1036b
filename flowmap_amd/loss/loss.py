"""Drop-in for flowmap/loss/loss.py."""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Optional, TypeVar

import torch
from torch import Tensor, nn


@dataclass
class LossCfgCommon:
    """flowmap/loss/loss.py:15-18"""

    enable_after: int
    weight: float


T = TypeVar("T", bound=LossCfgCommon)


class Loss(nn.Module, ABC, Generic[T]):
    """flowmap/loss/loss.py:24-58: gate on ``enable_after``, multiply by ``weight``."""

    cfg: T

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    def forward(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        enabled = global_step >= self.cfg.enable_after  # loss.py:39-41: a constant 0 until then
        if enabled:
            return self.compute_weighted_loss(batch, flows, tracks, model_output, global_step, self.cfg.weight)
        return torch.zeros((), dtype=torch.float32, device=batch.videos.device)

    def compute_weighted_loss(self, batch, flows, tracks, model_output, global_step: int, weight: float) -> Tensor:
        """weight × compute_unweighted_loss (loss.py:43-47).  Fused subclasses fold the
        weight into the kernel's normaliser instead of launching a scalar multiply."""
        return weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)

    @abstractmethod
    def compute_unweighted_loss(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        pass


def or_one(valid_sum: Tensor) -> Tensor:
    """``valid_sum or 1`` (loss_flow.py:70, loss_tracking.py:61) without the device->host
    sync of ``bool(tensor)``."""
    return torch.where(valid_sum == 0, torch.ones_like(valid_sum), valid_sum)
