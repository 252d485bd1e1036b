"""Stand-in: the loss base class (the gate and the weight every loss shares), the registry and its factory."""
from dataclasses import dataclass

import torch
from torch import nn


@dataclass
class LossCfgCommon:
    enable_after: int
    weight: float


class Loss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, batch, flows, tracks, model_output, global_step):
        switched_on = global_step >= self.cfg.enable_after
        if switched_on:
            return self.cfg.weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)
        return torch.zeros((), dtype=torch.float32, device=batch.videos.device)


# (after Loss: the two modules import it from here)
from .loss_flow import LossFlow, LossFlowCfg  # noqa: E402,F401
from .loss_tracking import LossTracking, LossTrackingCfg  # noqa: E402,F401

LOSSES = dict(flow=LossFlow, tracking=LossTracking)


def get_losses(cfgs):
    return [LOSSES[cfg.name](cfg) for cfg in cfgs]
