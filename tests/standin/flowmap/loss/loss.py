"""Stand-in: the gate and the weight every loss shares."""
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
        if global_step < self.cfg.enable_after:
            return torch.zeros((), dtype=torch.float32, device=batch.videos.device)
        return self.cfg.weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)
