"""FlowPredictor (flowmap/flow/flow_predictor.py:27-102) with everything after the optical-flow
network fused into one launch per temporal direction.

A concrete predictor still supplies ``forward(videos) -> (b, f-1, h, w, 2)`` (RAFT etc. are
out of scope here); ``compute_bidirectional_flow`` keeps the reference's signature and
result.  The static helpers are kept for callers that use them on their own.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Generic, Tuple, TypeVar

import torch.nn.functional as F
from torch import Tensor, nn

from .. import _ops
from ..types import Flows

T = TypeVar("T")


def split_videos(videos: Tensor):
    """flowmap/flow/common.py:6-22: (source, target, b, f) with the batch dims flattened."""
    b, f, c, h, w = videos.shape
    return videos[:, :-1].reshape(b * (f - 1), c, h, w), videos[:, 1:].reshape(b * (f - 1), c, h, w), b, f


class FlowPredictor(nn.Module, ABC, Generic[T]):
    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, videos: Tensor) -> Tensor:
        """videos (batch, frame, 3, height, width) -> flow (batch, frame-1, height, width, 2)."""

    @staticmethod
    def rescale_flow(flow: Tensor, shape: Tuple[int, int]) -> Tensor:
        """flow_predictor.py:39-47.  Plain resize (not on the fused path)."""
        b, f, h, w, _ = flow.shape
        flat = flow.permute(0, 1, 4, 2, 3).reshape(b * f, 2, h, w)
        out = F.interpolate(flat, shape, mode="bilinear", align_corners=False)
        return out.reshape(b, f, 2, *shape).permute(0, 1, 3, 4, 2)

    @staticmethod
    def rescale_mask(mask: Tensor, shape: Tuple[int, int]) -> Tensor:
        """flow_predictor.py:49-57."""
        b, f, h, w = mask.shape
        out = F.interpolate(mask.reshape(b * f, 1, h, w), shape, mode="bilinear", align_corners=False)
        return out.reshape(b, f, *shape)

    @staticmethod
    def compute_consistency_mask(videos: Tensor, flow: Tensor) -> Tensor:
        """flow_predictor.py:59-80, one kernel (fm_consistency_mask)."""
        return _ops.consistency_mask(videos, flow)

    def compute_bidirectional_flow(self, batch, flow_shape: Tuple[int, int]) -> Flows:
        """flow_predictor.py:82-102.  The network runs on the video and on its time-flipped
        copy exactly as in the reference; mask, resize and the flips back are one launch each."""
        videos = batch.videos
        forward, forward_mask = _ops.flow_postprocess(videos, self.forward(videos), flow_shape, reverse=False)
        backward_raw = self.forward(videos.flip(dims=(1,)))
        backward, backward_mask = _ops.flow_postprocess(videos, backward_raw, flow_shape, reverse=True)
        return Flows(forward, backward, forward_mask, backward_mask)
