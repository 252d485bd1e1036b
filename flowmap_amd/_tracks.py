"""The tracking loss's static data: ``PackedTracks`` (a track list packed once for the fused kernels, with its scatter plan) and ``TapPlan``
(the static tap set on one depth tensor shape and the compact tap image that travels between the fused flow pass and the tracking
loss).  The operators that use them are in _ops.py (TrackLossFused, FlowLossFused)."""

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
# Fused tracking loss
# --------------------------------------------------------------------------------------


class PackedTracks:
    """All track segments (flowmap/tracking/track_predictor.py:13-20) packed into the flat
    arrays fm_track_* expects.  Tracks are constants of the optimisation: packed once.
    ``own = (first, end)``: frame sharding — only frames first <= frame < end act as SOURCES on
    this rank (the targets of a segment can lie on any rank; they need poses, not depth)."""

    def __init__(self, tracks, device, own=None):
        xy, vis, seg, blocks, tiles = [], [], [], [], []
        offset = 0
        owned = (lambda frame: True) if own is None else (lambda frame: own[0] <= frame < own[1])
        for s_idx, t in enumerate(tracks):
            b, f, p, _ = t.xy.shape
            if b != 1:
                raise RuntimeError("flowmap_amd: the fused tracking loss supports batch size 1 (as the reference asserts)")
            start = int(t.start_frame)
            xy.append(t.xy[0].reshape(f * p, 2).to(device=device, dtype=torch.float32))
            vis.append(t.visibility[0].reshape(f * p).to(device=device, dtype=torch.uint8))
            seg.append([start, f, p, offset])
            blocks.extend([s_idx, fr] for fr in range(f) if owned(start + fr))
            tiles.extend([s_idx, fr] for fr in range(0, f, TRACK_TILE) if any(owned(start + q) for q in range(fr, min(fr + TRACK_TILE, f))))
            offset += f * p
        self.total = offset
        self.partial = own is not None  # some (segment, frame) entries are not sources here: flags start at 0
        self.xy = torch.cat(xy).contiguous()
        self.vis = torch.cat(vis).contiguous()
        self.seg = torch.tensor(seg, dtype=torch.int32).to(device)
        # frame-major launch order for the per-(segment, frame) kernels (track_points, track_scatter):
        # the ~8 segments that cover a frame gather from / scatter into the SAME depth image back to
        # back, so their 4-tap accesses share DRAM pages and L2 lines instead of sweeping 41 images
        blocks.sort(key=lambda sf: (seg[sf[0]][0] + sf[1], sf[0]))
        self.blocks = torch.tensor(blocks, dtype=torch.int32).reshape(-1, 2).to(device)
        self.nblocks = len(blocks)
        self.tiles = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 2).to(device)  # (segment, first source frame) per register tile
        self.ntiles = len(tiles)
        self.pmax = max(s_[2] for s_ in seg)
        self.fmax = max(s_[1] for s_ in seg)
        self.last_frame = max(s_[0] + s_[1] for s_ in seg)
        self.counts = [self.nblocks, self.ntiles, self.pmax, self.fmax, self.total, int(self.partial), self.last_frame,
                       0 if own is None else int(own[0]), -1 if own is None else int(own[1])]  # ..., source frames owned here [first, end)
        self._plans: dict = {}
        self._tap_slots: dict = {}
        self._tap_plans: dict = {}

    def scatter_plan(self, height: int, width: int):
        """Where the tracking gradient lands in dL/ddepth, planned once per image shape (tracks are
        constants): (pixels int64 ascending, first int32, source point of each entry int32, weights)
        for fm_depth_gather.  Built with one launch + a sort; None when nothing is scattered."""
        key = (int(height), int(width))
        if key not in self._plans:
            plan = None
            if self.nblocks > 0:
                dev = self.xy.device
                keys = torch.full((self.total * 4,), -1, dtype=torch.int64, device=dev)
                weights = torch.empty((self.total * 4,), dtype=torch.float32, device=dev)
                with _guard(dev):
                    call("fm_track_scatter_plan", ptr(self.xy), ptr(self.vis), ptr(self.seg), ptr(self.blocks), self.nblocks, self.pmax,
                         key[0], key[1], ptr(keys), ptr(weights), stream_for(self.xy))
                used = torch.nonzero(keys >= 0).reshape(-1)
                if used.numel() > 0:
                    sorted_keys, order = torch.sort(keys[used], stable=True)
                    entries = used[order]
                    pixels, counts = torch.unique_consecutive(sorted_keys, return_counts=True)
                    first = torch.zeros((pixels.numel() + 1,), dtype=torch.int32, device=dev)
                    first[1:] = torch.cumsum(counts, 0).to(torch.int32)
                    plan = (pixels.contiguous(), first, (entries // 4).to(torch.int32).contiguous(), weights[entries].contiguous())
                    # where each tap of each track point sits in `pixels` (its rank), -1 for a tap that contributes nothing: the tap
                    # exchange's view of the same plan (fm_track_loss_fused_fwd_taps)
                    # (bit 30: the pixel has more than one entry, i.e. several track points share it — fm_track_loss_fused_fwd_taps)
                    # (the slot encoding keeps 29 rank bits: bit 29 = read the depth image, TapPlan.slots_reading_around)
                    assert plan[0].numel() < 1 << 29
                    slots = torch.full((self.total * 4,), -1, dtype=torch.int32, device=dev)
                    ranks = torch.searchsorted(plan[0], keys[used])
                    shared = counts > 1
                    slots[used] = (ranks + shared[ranks].to(torch.int64) * (1 << 30)).to(torch.int32)
                    self._tap_slots[key] = (slots.contiguous(), torch.nonzero(shared).reshape(-1).to(torch.int32).contiguous())
            self._plans[key] = plan
        return self._plans[key]

    def tap_plan(self, frames: int, height: int, width: int):
        """The static tap set of this track list as the fused flow pass wants it (include/flowmap_hip.h: fm_flow_taps), built once per
        video shape: a TapPlan with the sorted tap pixels, the rank of the first tap of every 64-quad chunk of a frame of a (1, frames,
        height, width) depth tensor, each tap's pixel index inside its frame, and the (total, 4) slot of every tap of every track point.  None when the layout
        does not apply (width or pixel count not a multiple of 4, nothing scattered, a segment past the last frame)."""
        key = (int(frames), int(height), int(width))
        if key not in self._tap_plans:
            plan = self.scatter_plan(height, width) if self.nblocks > 0 and not self.partial else None
            n = int(height) * int(width)
            built = None
            if plan is not None and width % 4 == 0 and self.last_frame <= frames and plan[0].numel() < 2**29:
                pixels = plan[0]
                dev = pixels.device
                quads, chunks = n // 4, (n // 4 + 63) // 64
                # rank of the first tap at or after quad 64·c of frame f: taps with key < f·n + 256·c; one more entry at the end: M
                starts = (torch.arange(frames, dtype=torch.int64, device=dev)[:, None] * n
                          + torch.arange(chunks, dtype=torch.int64, device=dev)[None, :] * 256).reshape(-1)
                chunk_base = torch.cat([torch.searchsorted(pixels, starts), torch.tensor([pixels.numel()], dtype=torch.int64, device=dev)]).to(torch.int32).contiguous()
                pixel_in_frame = (pixels % n).to(torch.int32).contiguous()
                built = TapPlan(self, key, plan, chunk_base, pixel_in_frame, *self._tap_slots[(int(height), int(width))])
            self._tap_plans[key] = built
        return self._tap_plans[key]


class TapPlan:
    """The tracking loss's static tap set on one depth tensor shape, and the compact tap image that travels between the fused flow pass
    and the tracking loss (csrc/fm_flow.hip: TAPS; csrc/fm_track.hip: track_sample_many).  ``image`` (M floats) holds the depth value
    at every tap as the last flow pass left it; it may be sampled from only while the depth parameter has not moved since
    (``image_valid_for``: same storage, same version counter)."""

    def __init__(self, packed, key, plan, chunk_base, pixel_in_frame, slots, shared_ranks):
        self.packed, self.key, self.plan = packed, key, plan
        self.pixels, self.chunk_base, self.pixel_in_frame, self.slots, self.shared_ranks = plan[0], chunk_base, pixel_in_frame, slots, shared_ranks
        # (one value of padding: the tracking loss reads the two taps of an image row with one 8-byte load)
        self.image = torch.zeros((plan[0].numel() + 1,), dtype=torch.float32, device=plan[0].device)[: plan[0].numel()]
        self._tag = None  # what the image was left for: (the parameter object — weakly —, its storage object's identity, data_ptr, version)
        # raised by the flow pass when a tap depth it reads differs from the image value the tracking loss of the same step sampled: the
        # parameter was edited behind its version counter (`param.data.clamp_()` ...).  Read at the first sampled step and every 64th.
        self.stale_flag = torch.zeros((1,), dtype=torch.int32, device=plan[0].device)
        self.sampled_now = False  # the tracking loss of the current step sampled from the image: the coming flow pass verifies it
        self.samples = 0
        self.image_slots = slots  # the slot table to sample the CURRENT image with (an in-pass Adam update leaves one with holes: slots_reading_around)
        self.pending_in_pass = False  # the image was left by an in-pass Adam update whose step() has not finished: FusedAdam.step tags it

    def tag(self, root: Tensor) -> None:
        # (nothing here keeps the parameter or its storage alive: a plan outlives models — it hangs on the track tensors)
        self._tag = (weakref.ref(root), root.untyped_storage()._cdata, root.data_ptr(), root._version)
        self.pending_in_pass = False

    def slots_reading_around(self, kept: Optional[Tensor]) -> Tensor:
        """The slot table for sampling from an image an IN-PASS Adam update left: the taps at the pixels that update keeps for the
        element-list update (``kept``: sorted flat indices — the Procrustes samples and taps) are flagged to be read from the depth image
        (bit 29); the image holds their pre-update value.  Built once per kept set."""
        if kept is None or kept.numel() == 0:
            return self.slots
        key = (id(kept), kept._version)
        hit = self.__dict__.get("_around")
        if hit is None or hit[0] != key:
            dense_rank = torch.isin(self.pixels, kept)  # per tap (rank): is its pixel kept?
            slots = self.slots.clone()
            valid = slots >= 0
            ranks = (slots[valid] & 0x1FFFFFFF).to(torch.int64)
            slots[valid] = slots[valid] | (dense_rank[ranks].to(torch.int32) << 29)
            hit = self.__dict__["_around"] = (key, kept, slots.contiguous())
        return hit[2]

    def invalidate(self) -> None:
        self._tag = None

    def image_valid_for(self, root: Tensor) -> bool:
        tag = self._tag
        if tag is None or tag[0]() is not root or tag[1] != root.untyped_storage()._cdata or tag[2:] != (root.data_ptr(), root._version):
            return False
        # a step replayed as a hipGraph runs no Python: whether depth moved between replays could not be checked
        return not (root.is_cuda and torch.cuda.is_current_stream_capturing())

    def note_sampled(self) -> None:
        self.sampled_now = True
        self.samples += 1

    def check_stale(self) -> None:
        """(synchronises) Raise if a flow pass found the image stale although the version counter said otherwise."""
        if int(self.stale_flag.item()) != 0:
            self.stale_flag.zero_()
            self.invalidate()
            raise RuntimeError(
                "flowmap_amd: the depth parameter was modified without its version counter moving (an edit through `.data`, a raw pointer): the "
                "tracking loss sampled tap depths the last flow pass had left behind, and they were stale — the tracking loss and its gradients "
                "of the affected steps are wrong.  Edit parameters in place under torch.no_grad() (as optimisers do), or set "
                "flowmap_amd._ops.options.tap_image = False.")
