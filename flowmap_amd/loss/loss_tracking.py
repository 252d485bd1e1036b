"""Drop-in for flowmap/loss/loss_tracking.py."""

from __future__ import annotations

import weakref
from dataclasses import dataclass
from typing import Literal, Optional

from torch import Tensor

import torch

from .. import _ops
from ..model.projection import LazySurfaces, _dense_extrinsics, compute_track_flow
from .loss import Loss, LossCfgCommon, or_one
from .mapping import MappingCfg, get_mapping


@dataclass
class LossTrackingCfg(LossCfgCommon):
    name: Literal["tracking"]
    mapping: MappingCfg


class LossTracking(Loss[LossTrackingCfg]):
    """flowmap/loss/loss_tracking.py:23-61: all-pairs track reprojection error over every
    segment, one global masked mean.

    Fast path (LazySurfaces of the model's own depths/intrinsics, batch 1): all segments
    in a handful of HIP launches straight from depth (fm_track_*); the (f,f,P,2) all-pairs
    tensors are never materialised.  General path: the reference's per-segment
    composition of compute_track_flow -> mapping -> masked sums.
    """

    reference_name = "LossTracking"

    # let ProcrustesFit.backward apply our sparse depth scatter to its final buffer
    defer_depth_scatter: bool = True

    def __init__(self, cfg: LossTrackingCfg) -> None:
        super().__init__(cfg)
        self.mapping = get_mapping(cfg.mapping)

    @staticmethod
    def _fusable(model_output, tracks) -> bool:
        s = model_output.surfaces
        return (
            isinstance(s, LazySurfaces) and s.depths is model_output.depths and s.depths.shape[0] == 1 and tracks is not None
            and len(tracks) > 0
        )

    def _fused(self, tracks, model_output, weight: float, look_ahead: bool = False) -> Tensor:
        """``look_ahead``: called by the flow loss of the same step just before its own pass (LossFlow._look_ahead) — the loss is evaluated
        now, its depth gradient left at the static taps for that pass to absorb, and the value kept on the step's depth tensor until this
        loss is asked for it with the same track list."""
        s: LazySurfaces = model_output.surfaces
        kept = s.depths.__dict__.get("_fm_tracking_ahead")
        if not look_ahead and kept is not None:
            hit = kept.pop((id(self), float(weight)), None)
            if hit is not None and hit[0] is tracks:
                return hit[1]
        packed = _ops.pack_tracks(tracks, s.depths.device)
        loss = _ops.TrackLossFused.apply(
            s.depths, model_output.intrinsics, _dense_extrinsics(model_output.extrinsics), packed, weight, _ops.MAPPING_KINDS[self.mapping.kind],
            self.mapping.delta, self.defer_depth_scatter, offer_taps=look_ahead,
        )
        if look_ahead:
            s.depths.__dict__.setdefault("_fm_tracking_ahead", {})[(id(self), float(weight))] = (tracks, loss)
        elif s.depths.__dict__.get("_fm_flow_ran") and self.defer_depth_scatter:
            # this loss followed a fused flow loss on the same depth tensor: from the next step on that flow loss evaluates it ahead of its pass
            _ops._root(s.depths).__dict__["_fm_tracking_follows_flow"] = (weakref.ref(self), float(weight))
        return loss

    def forward(self, batch, flows, tracks, model_output, global_step: int) -> Tensor:
        if global_step < self.cfg.enable_after and tracks is not None and self._fusable(model_output, tracks):
            _ops.announce_track_pixels(model_output.surfaces.depths, tracks)
        return super().forward(batch, flows, tracks, model_output, global_step)

    def compute_weighted_loss(self, batch, flows, tracks, model_output, global_step: int, weight: float) -> Tensor:
        assert tracks is not None
        if self._fusable(model_output, tracks):
            return self._fused(tracks, model_output, weight)
        return weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)

    def compute_unweighted_loss(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        # Tracks must be available for the tracking loss (loss_tracking.py:37).
        assert tracks is not None
        if self._fusable(model_output, tracks):
            return self._fused(tracks, model_output, 1.0)
        if len(tracks) == 0:
            return torch.zeros((), dtype=torch.float32, device=batch.videos.device)

        image_shape = tuple(batch.videos.shape[-2:])
        numerators, counts = [], []
        for segment in tracks:
            # the frames this segment's tracks live on (loss_tracking.py:44-52)
            window = slice(segment.start_frame, segment.start_frame + segment.xy.shape[1])
            predicted, visible = compute_track_flow(
                model_output.surfaces[:, window], model_output.extrinsics[:, window], model_output.intrinsics[:, window], segment
            )
            # every source frame is compared with the track position in every target frame
            residual = self.mapping.forward(predicted, segment.xy[:, None], image_shape)
            numerators.append((residual * visible).sum())
            counts.append(visible.sum())
        # ONE ratio over all segments (loss_tracking.py:58-61), `or 1` evaluated on the device
        return torch.stack(numerators).sum() / or_one(torch.stack(counts).sum())
