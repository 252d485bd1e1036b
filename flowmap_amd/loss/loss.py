"""Drop-in for flowmap/loss/loss.py: the gate and the weight every loss shares."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn

from .. import _reference


def or_one(valid_sum: Tensor) -> Tensor:
    """``valid_sum or 1`` (loss_flow.py:70, loss_tracking.py:61) without the device->host
    sync of ``bool(tensor)``."""
    return torch.where(valid_sum == 0, torch.ones_like(valid_sum), valid_sum)


@dataclass
class LossCfgCommon:
    """flowmap/loss/loss.py:15-18: first step at which the loss counts, and its weight."""

    enable_after: int
    weight: float


class Loss(nn.Module):
    """flowmap/loss/loss.py:24-58.  ``Loss[Cfg]`` is accepted like the reference's generic base and
    means nothing at run time."""

    def __class_getitem__(cls, _cfg_type):
        return cls

    def __init__(self, cfg: LossCfgCommon) -> None:
        super().__init__()
        self.cfg = cfg

    # the reference class this one replaces in flowmap.loss.LOSSES (its name in flowmap_amd._reference.twins); None: no host path
    reference_name: Optional[str] = None

    def forward(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        ref_cls = _reference.host_twin(self.reference_name, batch) if self.reference_name else None
        if ref_cls is not None:  # host tensors after install(): the reference's own loss (flowmap_amd/_reference.py)
            twin = self.__dict__.get("_fm_host_twin")
            if twin is None or type(twin) is not ref_cls:
                twin = self.__dict__["_fm_host_twin"] = ref_cls(self.cfg)
            return twin.forward(batch, flows, tracks, model_output, global_step)
        if global_step >= self.cfg.enable_after:
            return self.compute_weighted_loss(batch, flows, tracks, model_output, global_step, self.cfg.weight)
        # loss.py:39-41: a constant 0 until the loss is switched on
        return torch.zeros((), dtype=torch.float32, device=batch.videos.device)

    def compute_weighted_loss(self, batch, flows, tracks, model_output, global_step: int, weight: float) -> Tensor:
        """weight × compute_unweighted_loss (loss.py:43-47).  Fused subclasses fold the
        weight into the kernel's normaliser instead of launching a scalar multiply."""
        return weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)

    def compute_unweighted_loss(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        raise NotImplementedError(f"{type(self).__name__} must implement compute_unweighted_loss (loss.py:49-58)")
