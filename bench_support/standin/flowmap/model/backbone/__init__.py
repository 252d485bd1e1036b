"""Stand-in: the backbone registry (explicit depth only: depth and weight logits as free parameters)."""
from dataclasses import dataclass

import torch
from torch import Tensor, nn


@dataclass
class BackboneOutput:
    depths: Tensor  # (batch, frame, height, width)
    weights: Tensor  # (batch, frame - 1, height, width)


@dataclass
class BackboneExplicitDepthCfg:
    name: str
    initial_depth: float
    weight_sensitivity: float


class BackboneExplicitDepth(nn.Module):
    def __init__(self, cfg, num_frames, image_shape):
        super().__init__()
        self.cfg = cfg
        self.depth = nn.Parameter(torch.full((num_frames, *image_shape), float(cfg.initial_depth)))
        self.weights = nn.Parameter(torch.zeros((num_frames - 1, *image_shape)))

    def forward(self, batch, flows):
        assert batch.videos.shape[0] == 1
        return BackboneOutput(self.depth[None], torch.sigmoid(self.cfg.weight_sensitivity * self.weights)[None])


BACKBONES = {"explicit_depth": BackboneExplicitDepth}


def get_backbone(cfg, num_frames, image_shape):
    return BACKBONES[cfg.name](cfg, num_frames, image_shape)
