"""Stand-in: the model glue — parts resolved through the registries, geometry through the names bound here at import."""
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from .backbone import get_backbone
from .extrinsics import get_extrinsics
from .intrinsics import get_intrinsics
from .projection import sample_image_grid, unproject


@dataclass
class ModelCfg:
    backbone: object
    intrinsics: object
    extrinsics: object
    use_correspondence_weights: bool


@dataclass
class ModelOutput:
    depths: Tensor
    surfaces: Tensor
    intrinsics: Tensor
    extrinsics: Tensor
    backward_correspondence_weights: Tensor


class Model(nn.Module):
    def __init__(self, cfg, num_frames=None, image_shape=None):
        super().__init__()
        self.cfg = cfg
        self.backbone = get_backbone(cfg.backbone, num_frames, image_shape)
        self.intrinsics = get_intrinsics(cfg.intrinsics)
        self.extrinsics = get_extrinsics(cfg.extrinsics, num_frames)

    def forward(self, batch, flows, global_step):
        parts = self.backbone.forward(batch, flows)
        if not self.cfg.use_correspondence_weights:
            parts.weights = torch.ones_like(parts.weights)
        k = self.intrinsics.forward(batch, flows, parts, global_step)
        grid, _ = sample_image_grid(batch.videos.shape[-2:], device=batch.videos.device)
        surfaces = unproject(grid, parts.depths, k[:, :, None, None])
        poses = self.extrinsics.forward(batch, flows, parts, surfaces)
        return ModelOutput(parts.depths, surfaces, k, poses, parts.weights)
