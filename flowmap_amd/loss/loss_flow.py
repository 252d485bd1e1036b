"""Drop-in for flowmap/loss/loss_flow.py."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

import torch
from torch import Tensor

from .. import _ops
from ..model.projection import LazySurfaces, _dense_extrinsics, compute_backward_flow, compute_forward_flow, sample_image_grid
from .loss import Loss, LossCfgCommon, or_one
from .mapping import MappingCfg, get_mapping


@dataclass
class LossFlowCfg(LossCfgCommon):
    name: Literal["flow"]
    mapping: MappingCfg


class LossFlow(Loss[LossFlowCfg]):
    """flowmap/loss/loss_flow.py:26-70.

    Fast path (surfaces are a LazySurfaces of the model's own depths/intrinsics): ONE
    HIP kernel evaluates both flow directions straight from depth and writes every
    gradient in the same pass (fm_flow_loss_fused); nothing of size (b,f,h,w,3) or
    (b,f-1,h,w,2) is ever materialised.  General path (explicit surfaces tensor): the
    reference's composition of compute_*_flow -> mapping -> masked mean, each step a HIP
    kernel with its own backward.
    """

    reference_name = "LossFlow"

    # tuning knob of the fused kernel (items per thread); None -> library default
    items_per_thread: Optional[int] = None
    # park the dense depth gradient on the Procrustes node instead of returning it twice
    carry_depth_grad: bool = True
    # frame sharding: maps the local Σmask (fp64 device tensor) to the global one
    valid_sum_reducer = None
    # take the relative poses align_surfaces attached to the extrinsics instead of inverting the chain
    use_fitted_poses: bool = True

    def __init__(self, cfg: LossFlowCfg) -> None:
        super().__init__(cfg)
        self.mapping = get_mapping(cfg.mapping)

    # -- fused ----------------------------------------------------------------------------
    @staticmethod
    def _fusable(model_output) -> bool:
        s = model_output.surfaces
        return isinstance(s, LazySurfaces) and s.depths is model_output.depths

    # The tap exchange: when, in the previous step, a fused tracking loss followed this loss on the same depth parameter,
    # evaluate it AHEAD of the flow pass — same arguments, same value, returned when that loss is called — so that the pass can absorb its
    # depth gradient at the static taps instead of the tracking loss read-modify-writing cold lines of dL/ddepth afterwards.
    look_ahead: bool = True

    def _look_ahead(self, tracks, model_output, global_step: int) -> None:
        depths = model_output.surfaces.depths
        note = _ops._root(depths).__dict__.get("_fm_tracking_follows_flow")
        if note is None or not (torch.is_grad_enabled() and depths.requires_grad) or not _ops.options.tap_exchange:
            return
        follower, weight = note[0](), note[1]
        if follower is None or global_step < follower.cfg.enable_after or not follower._fusable(model_output, tracks):
            return
        if _ops.tap_plan_of(depths) is None or (id(follower), weight) in depths.__dict__.get("_fm_tracking_ahead", {}):
            return
        follower._fused(tracks, model_output, weight, look_ahead=True)

    def _fused(self, flows, model_output, weight: float, tracks=None, global_step: int = 0) -> Tensor:
        s: LazySurfaces = model_output.surfaces
        if tracks is not None and self.look_ahead and self.carry_depth_grad:
            self._look_ahead(tracks, model_output, global_step)
        s.depths.__dict__["_fm_flow_ran"] = True
        direct = getattr(model_output.extrinsics, "_fm_relative_poses", None) if self.use_fitted_poses else None
        if direct is not None and direct[0].shape[:2] == (s.depths.shape[0], s.depths.shape[1] - 1):
            rel_fwd, rel_bwd = direct  # straight from the Procrustes fit (align_surfaces)
        else:
            rel_fwd, rel_bwd = _ops.RelativePoses.apply(_dense_extrinsics(model_output.extrinsics))
        norm = _ops.flow_valid_norm(flows.forward_mask, flows.backward_mask, weight, self.valid_sum_reducer)
        packed = _ops.packed_flow_inputs(flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)  # cached
        return _ops.FlowLossFused.apply(
            s.depths, model_output.intrinsics, rel_fwd, rel_bwd, flows.forward, flows.backward, flows.forward_mask,
            flows.backward_mask, norm, _ops.MAPPING_KINDS[self.mapping.kind], self.mapping.delta, self.carry_depth_grad,
            self.items_per_thread or 0, packed,
        )

    def compute_weighted_loss(self, batch, flows, tracks, model_output, global_step: int, weight: float) -> Tensor:
        if self._fusable(model_output):
            return self._fused(flows, model_output, weight, tracks, global_step)
        return weight * self.compute_unweighted_loss(batch, flows, tracks, model_output, global_step)

    # -- general --------------------------------------------------------------------------
    def compute_unweighted_loss(self, batch, flows, tracks: Optional[list], model_output, global_step: int) -> Tensor:
        if self._fusable(model_output):
            return self._fused(flows, model_output, 1.0, tracks, global_step)

        _, _, _, h, w = batch.videos.shape
        device = batch.videos.device
        xy, _ = sample_image_grid((h, w), device)

        # forward flow term (loss_flow.py:46-56)
        xy_flowed_forward = compute_forward_flow(model_output.surfaces, model_output.extrinsics, model_output.intrinsics)
        forward_loss = self.mapping.forward(xy_flowed_forward - xy, flows.forward, (h, w))
        loss_sum = (forward_loss * flows.forward_mask).sum()
        valid_sum = flows.forward_mask.sum()

        # backward flow term (loss_flow.py:58-68)
        xy_flowed_backward = compute_backward_flow(model_output.surfaces, model_output.extrinsics, model_output.intrinsics)
        backward_loss = self.mapping.forward(xy_flowed_backward - xy, flows.backward, (h, w))
        loss_sum = loss_sum + (backward_loss * flows.backward_mask).sum()
        valid_sum = valid_sum + flows.backward_mask.sum()

        return loss_sum / or_one(valid_sum)
