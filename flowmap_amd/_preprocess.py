"""One-off data preparation on the GPU: flow / mask post-processing (flowmap/flow/flow_predictor.py:39-102) and the
one-pass resize + crop (flowmap/misc/cropping.py).  No gradients: flows, masks and videos are constants of the optimisation."""

from __future__ import annotations

import ctypes
import warnings
import weakref
from typing import Optional

import torch
from torch import Tensor

from ._lib import call, check_device, ptr, stream_for, torch_ops  # noqa: F401
from ._base import AUX_STRIDE, PAIR_GRAD_STRIDE, STAT_STRIDE, TRACK_TILE, _f32c, _guard  # noqa: F401

# --------------------------------------------------------------------------------------
# Flow post-processing (no gradients: flows and masks are constants of the optimisation)
# --------------------------------------------------------------------------------------


def _check_video_flow(videos: Tensor, flow: Tensor):
    check_device(videos, flow)
    if videos.dim() != 5 or videos.shape[2] != 3:
        raise RuntimeError("flowmap_amd: videos must be (batch, frame, 3, height, width)")
    b, f, _, h, w = videos.shape
    if f < 2 or tuple(flow.shape) != (b, f - 1, h, w, 2):
        raise RuntimeError("flowmap_amd: flow must be (batch, frame-1, height, width, 2) at the video's resolution")
    return b, f, h, w


def consistency_mask(videos: Tensor, flow: Tensor) -> Tensor:
    """FlowPredictor.compute_consistency_mask (flowmap/flow/flow_predictor.py:59-80)."""
    b, f, h, w = _check_video_flow(videos, flow)
    with torch.no_grad():
        videos, flow = _f32c(videos, "videos"), _f32c(flow, "flow")
        mask = torch.empty((b, f - 1, h, w), dtype=torch.float32, device=videos.device)
        with _guard(videos.device):
            call("fm_consistency_mask", ptr(videos), ptr(flow), b, f, h, w, ptr(mask), stream_for(videos))
    return mask


def flow_postprocess(videos: Tensor, raw_flow: Tensor, shape, reverse: bool):
    """Consistency mask + rescale_flow + rescale_mask (+ the flips back when ``reverse``) of
    compute_bidirectional_flow (flow_predictor.py:82-102).  -> (flow (b,f-1,*shape,2), mask)."""
    b, f, h, w = _check_video_flow(videos, raw_flow)
    oh, ow = int(shape[0]), int(shape[1])
    with torch.no_grad():
        videos, raw_flow = _f32c(videos, "videos"), _f32c(raw_flow, "flow")
        out_flow = torch.empty((b, f - 1, oh, ow, 2), dtype=torch.float32, device=videos.device)
        out_mask = torch.empty((b, f - 1, oh, ow), dtype=torch.float32, device=videos.device)
        with _guard(videos.device):
            call("fm_flow_postprocess", ptr(videos), ptr(raw_flow), b, f, h, w, oh, ow, 1 if reverse else 0, ptr(out_flow),
                 ptr(out_mask), stream_for(videos))
    return out_flow, out_mask


def resize_crop(images: Tensor, resized_shape, crop_shape) -> Tensor:
    """``center_crop_images(F.interpolate(images, resized_shape, bilinear), crop_shape)``
    (flowmap/misc/cropping.py:19-51) for (..., H, W) images in one launch; no gradients (it is
    data preparation)."""
    check_device(images)
    *lead, h, w = images.shape
    rh, rw = int(resized_shape[0]), int(resized_shape[1])
    oh, ow = int(crop_shape[0]), int(crop_shape[1])
    if oh > rh or ow > rw or min(oh, ow, rh, rw) < 1:
        raise RuntimeError("flowmap_amd: the crop must fit inside the resized image")
    planes = 1
    for d in lead:
        planes *= int(d)
    with torch.no_grad():
        images = _f32c(images, "images")
        out = torch.empty((*lead, oh, ow), dtype=torch.float32, device=images.device)
        done = 0
        with _guard(images.device):
            while done < planes:  # the launch takes at most 65535 planes
                chunk = min(planes - done, 65535)
                call("fm_resize_crop", images.data_ptr() + done * h * w * 4, chunk, h, w, rh, rw, (rh - oh) // 2, (rw - ow) // 2, oh, ow,
                     out.data_ptr() + done * oh * ow * 4, stream_for(images))
                done += chunk
    return out
