"""Stand-in: the point-track loss over all segments (one global masked mean)."""
from dataclasses import dataclass

from ..model.projection import compute_track_flow
from . import Loss, LossCfgCommon
from .mapping import get_mapping


@dataclass
class LossTrackingCfg(LossCfgCommon):
    name: str
    mapping: object


class LossTracking(Loss):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.mapping = get_mapping(cfg.mapping)

    def compute_unweighted_loss(self, batch, flows, tracks, model_output, global_step):
        hw = batch.videos.shape[-2:]
        total, count = 0, 0
        for segment in tracks:
            window = slice(segment.start_frame, segment.start_frame + segment.xy.shape[1])
            where, valid = compute_track_flow(model_output.surfaces[:, window], model_output.extrinsics[:, window],
                                              model_output.intrinsics[:, window], segment)
            total = total + (self.mapping.forward(where, segment.xy[:, None], hw) * valid).sum()
            count = count + valid.sum()
        return total / (count or 1)
