"""Drop-in for flowmap/loss/loss_tracking.py."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

from torch import Tensor

from ..model.projection import compute_track_flow
from .loss import Loss, LossCfgCommon, or_one
from .mapping import MappingCfg, get_mapping


@dataclass
class LossTrackingCfg(LossCfgCommon):
    name: Literal["tracking"]
    mapping: MappingCfg


class LossTracking(Loss[LossTrackingCfg]):
    """flowmap/loss/loss_tracking.py:23-61: all-pairs track reprojection error over every
    segment, one global masked mean."""

    def __init__(self, cfg: LossTrackingCfg) -> None:
        super().__init__(cfg)
        self.mapping = get_mapping(cfg.mapping)

    def compute_unweighted_loss(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        # Tracks must be available for the tracking loss (loss_tracking.py:37).
        assert tracks is not None

        _, _, _, h, w = batch.videos.shape

        loss_sum = 0
        valid_sum = 0

        for segment_tracks in tracks:
            _, f, _, _ = segment_tracks.xy.shape
            s = segment_tracks.start_frame

            xy_target, visibility = compute_track_flow(
                model_output.surfaces[:, s : s + f],
                model_output.extrinsics[:, s : s + f],
                model_output.intrinsics[:, s : s + f],
                segment_tracks,
            )
            xy_target_gt = segment_tracks.xy[:, None]  # "b ft p xy -> b () ft p xy"

            loss = self.mapping.forward(xy_target, xy_target_gt, (h, w)) * visibility

            loss_sum = loss_sum + loss.sum()
            valid_sum = valid_sum + visibility.sum()

        return loss_sum / or_one(valid_sum)
