"""Stand-in: poses from the weighted rigid fit of consecutive surfaces at sampled pixels."""
from dataclasses import dataclass
from typing import Optional

from torch import nn

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)

from ..projection import align_surfaces


@dataclass
class ExtrinsicsProcrustesCfg:
    name: str
    num_points: Optional[int]
    randomize_points: bool


class ExtrinsicsProcrustes(nn.Module):
    def __init__(self, cfg, num_frames):
        super().__init__()
        self.cfg = cfg

    def forward(self, batch, flows, backbone_output, surfaces):
        h, w = surfaces.shape[2:4]
        indices = orc.procrustes_indices((h, w), self.cfg.num_points, surfaces.device, self.cfg.randomize_points)
        return align_surfaces(surfaces, flows.backward, backbone_output.weights, indices)
