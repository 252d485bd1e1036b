"""Stand-in: the flow-consistency loss (host arithmetic: the oracle's, through the mapping registry)."""
from dataclasses import dataclass

from ..model.projection import compute_backward_flow, compute_forward_flow, sample_image_grid
from . import Loss, LossCfgCommon
from .mapping import get_mapping


@dataclass
class LossFlowCfg(LossCfgCommon):
    name: str
    mapping: object


class LossFlow(Loss):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.mapping = get_mapping(cfg.mapping)

    def compute_unweighted_loss(self, batch, flows, tracks, model_output, global_step):
        hw = batch.videos.shape[-2:]
        grid, _ = sample_image_grid(hw, batch.videos.device)
        ahead = compute_forward_flow(model_output.surfaces, model_output.extrinsics, model_output.intrinsics) - grid
        back = compute_backward_flow(model_output.surfaces, model_output.extrinsics, model_output.intrinsics) - grid
        total = (self.mapping.forward(ahead, flows.forward, hw) * flows.forward_mask).sum()
        total = total + (self.mapping.forward(back, flows.backward, hw) * flows.backward_mask).sum()
        return total / ((flows.forward_mask.sum() + flows.backward_mask.sum()) or 1)
