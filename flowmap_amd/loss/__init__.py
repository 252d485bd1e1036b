"""Losses of the hot path — counterpart of the reference package flowmap/loss (registry and
factory at flowmap/loss/__init__.py:5-14)."""

from typing import Iterable, List, Union

from .loss import Loss
from .loss_flow import LossFlow, LossFlowCfg
from .loss_tracking import LossTracking, LossTrackingCfg

LossCfg = Union[LossFlowCfg, LossTrackingCfg]

# cfg.name -> class, the same keys the reference's registry uses
LOSSES = {"flow": LossFlow, "tracking": LossTracking}


def get_losses(cfgs: Iterable[LossCfg]) -> List[Loss]:
    """One loss module per config, in order (ModelWrapperOverfit sums them)."""
    built = []
    for cfg in cfgs:
        built.append(LOSSES[cfg.name](cfg))
    return built


__all__ = ["LOSSES", "Loss", "LossCfg", "LossFlow", "LossFlowCfg", "LossTracking", "LossTrackingCfg", "get_losses"]
