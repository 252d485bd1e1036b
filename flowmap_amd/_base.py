"""Shared plumbing of the operator modules (flowmap_amd/_ops.py and its parts): the C ABI's layout constants, the device flags a backward
raises when a loss reached it scaled, dtype / contiguity checks, the fm_layout description of frame windows, the device guard and the
"derived data lives on the tensor it was derived from" helper.  No operator lives here."""

from __future__ import annotations

import ctypes
import warnings
import weakref
from typing import Optional

import torch
from torch import Tensor

from ._lib import call, check_device, ptr, stream_for, torch_ops  # noqa: F401

MAPPING_KINDS = {"huber": 0, "l1": 1, "l2": 2}

FLOW_ACC_STRIDE = 20
STAT_STRIDE = 16
AUX_STRIDE = 40
PAIR_GRAD_STRIDE = 20
DENSE_CONST_STRIDE = 40  # FM_DENSE_CONST_STRIDE
TRACK_TILE = 6  # FM_TRACK_TILE (include/flowmap_hip.h; tests/test_abi.py checks they agree)


# Device flags (one int32 each) that a backward raises when a loss reached it SCALED although something had already used its unscaled
# gradient (FusedAdam.fuse_depth_update, FrameShard.enable_early_halo).  Their owners read them every so often; a step replayed as a
# hipGraph runs no Python, so GraphedStep reads every live flag outside its replays.  Weak: a flag dies with its owner.
_unit_flags = weakref.WeakSet()


def register_unit_flag(flag: Tensor) -> Tensor:
    _unit_flags.add(flag)
    return flag


def check_unit_flags(what: str) -> None:
    """Raise (and clear) if any registered flag is up.  Synchronises: callers space their calls out."""
    for flag in list(_unit_flags):
        if int(flag.item()) != 0:
            flag.zero_()
            raise RuntimeError(f"flowmap_amd: {what}: a loss reached backward() with an upstream gradient other than 1 although its unscaled gradient had "
                               "already been used (FusedAdam.fuse_depth_update applied it inside the flow pass / FrameShard.enable_early_halo sent it before "
                               "backward): the affected steps are wrong.  Switch those options off for a scaled or averaged loss.")


def _f32c(t: Tensor, what: str) -> Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"flowmap_amd: {what} must be float32 (got {t.dtype})")
    if t.is_contiguous():
        return t
    if t.numel() >= 1 << 24:  # a copy the size of a pass over the step's tensors: say so (SURVEY.md §8b "Ownership")
        warnings.warn(f"flowmap_amd: {what} is a non-contiguous view of {t.numel() * 4 >> 20} MB and is copied on every call; pass a contiguous tensor")
    return t.contiguous()


class FmLayout(ctypes.Structure):
    """include/flowmap_hip.h: fm_layout — element strides between the frames / batch entries of an image stack ({0, 0} = dense)."""

    _fields_ = [("frame_stride", ctypes.c_long), ("batch_stride", ctypes.c_long)]


def frame_window_layout(t: Tensor) -> Optional[tuple]:
    """(frame_stride, batch_stride) in elements when ``t`` (batch, frame, ...) can be read in place — every frame dense, i.e. a
    contiguous tensor or a frame window ``x[:, s:s+f]`` / batch slice of one — else None (the caller copies)."""
    if t.is_contiguous():
        return (0, 0)
    if t.dim() < 3:
        return None
    per_frame = 1
    for d in range(t.dim() - 1, 1, -1):
        if t.shape[d] != 1 and t.stride(d) != per_frame:
            return None
        per_frame *= t.shape[d]
    if t.shape[1] != 1 and t.stride(1) < per_frame:
        return None
    frame_stride = per_frame if t.shape[1] == 1 else t.stride(1)
    # batch entries must not overlap (the launchers refuse it): a batch-expanded stack (stride 0 over the batch) is copied by the caller
    if t.shape[0] != 1 and t.stride(0) < frame_stride * (t.shape[1] - 1) + per_frame:
        return None
    return (frame_stride, frame_stride * t.shape[1] if t.shape[0] == 1 else t.stride(0))


def _layout_array(*tensors):
    """(ctypes array of fm_layout, any of them a real view?) for the `_views` entry points; None when a tensor cannot be read in place."""
    arr = (FmLayout * len(tensors))()
    any_view = False
    for i, t in enumerate(tensors):
        lay = frame_window_layout(t)
        if lay is None:
            return None, False
        arr[i].frame_stride, arr[i].batch_stride = lay
        any_view = any_view or lay != (0, 0)
    return arr, any_view


class _guard:
    """Select the tensor's GPU for the launches inside (no-op for the host test double)."""

    def __init__(self, dev: torch.device):
        self.ctx = torch.cuda.device(dev) if dev.type == "cuda" else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _derived(owner: Tensor, name: str, key, build):
    """``build()`` once per (owner tensor object, key): the value is kept on the tensor itself."""
    slot = owner.__dict__.get(name)
    if slot is not None and slot[0] == key:
        return slot[1]
    value = build()
    owner.__dict__[name] = (key, value)
    return value
