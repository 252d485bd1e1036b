"""The differentiable building blocks the reference-shaped call surface (flowmap_amd.model.projection,
flowmap_amd.loss) is assembled from.

Per-step operators — K from the focal length, the Procrustes pose fit, the pose chain, the fused flow and
tracking losses, Adam — are C++ ``torch::autograd::Function``s in libflowmap_torch.so
(csrc/fm_torch.cpp, registered with TORCH_LIBRARY as ``torch.ops.flowmap_amd.*``); the classes of the same
names below are thin facades that gather what those operators take explicitly.  Function-level
operators on explicit point sets (Unproject, Reproject, BilinearSample, ...) and the data-preparation
kernels are ``torch.autograd.Function``s over the same C ABI through ctypes.  Nothing here synchronises
the device or falls back to eager PyTorch math.

State: there is no module-level cache.  Everything derived from a constant input lives ON that
input's tensor object (a Python attribute, validated against the tensor's version counter) and dies
with it: K^-1 on K, the packed flows / valid sums on the flow and mask tensors, the static scatter
plans on the backward-flow tensor, the packed tracks on the first segment's coordinates.  What one
step hands from operator to operator travels in explicit objects: ``DepthSink`` (one per depth tensor
object, i.e. per step) and ``GradArena`` (one per weight parameter).
"""

from __future__ import annotations

import ctypes
import warnings
import weakref
from typing import Optional

import torch
from torch import Tensor

from ._lib import call, check_device, ptr, stream_for, torch_ops
from .config import options

from ._base import (AUX_STRIDE, DENSE_CONST_STRIDE, FLOW_ACC_STRIDE, MAPPING_KINDS, PAIR_GRAD_STRIDE, STAT_STRIDE, TRACK_TILE, FmLayout,  # noqa: F401
                    _derived, _f32c, _guard, _layout_array, _unit_flags, check_unit_flags, frame_window_layout, register_unit_flag)


# --------------------------------------------------------------------------------------
# Intrinsics
# --------------------------------------------------------------------------------------


def intrinsics_inverse(k: Tensor) -> Tensor:
    """K^-1 for a (..., 3, 3) stack (no autograd; callers chain the backward).  The result stays on the
    K tensor object it was computed for — views of it included: ``K[:, :, None, None]`` or ``K[:, 1:]`` look
    it up on their base, each window of memory beside the others — so the consumers of one step invert
    each K once and constant K sets once per run; an in-place edit of K (version counter) or another
    tensor object misses."""
    k = _f32c(k, "intrinsics")
    root = k if k._base is None else k._base
    slots = root.__dict__.setdefault("_fm_kinv", {})
    hit = slots.get((k.data_ptr(), k.numel()))
    if hit is not None and hit[0] == k._version:
        return hit[1].view(k.shape)
    kinv = torch_ops().intrinsics_inverse(k)
    if len(slots) >= 4:
        slots.clear()
    slots[(k.data_ptr(), k.numel())] = (k._version, kinv)
    return kinv


def _attach_inverse(k: Tensor, kinv: Tensor) -> None:
    k.__dict__.setdefault("_fm_kinv", {})[(k.data_ptr(), k.numel())] = (k._version, kinv)


def focal_intrinsics(focal: Tensor, repeat_shape, image_shape) -> Tensor:
    """focal_lengths_to_intrinsics spread over the frames (intrinsics/common.py:6-20 as used by
    intrinsics_regressed.py:34-41): K (*focal.shape, *repeat_shape, 3, 3) from normalised focal lengths in
    one launch that also computes K^-1, left on K for the step's consumers (intrinsics_inverse)."""
    k, kinv = torch_ops().focal_intrinsics(focal, [int(d) for d in repeat_shape], int(image_shape[0]), int(image_shape[1]))
    _attach_inverse(k, kinv)
    return k


# --------------------------------------------------------------------------------------
# What one step hands between its operators
# --------------------------------------------------------------------------------------


def depth_sink(depth: Tensor):
    """The DepthSink of this depth tensor OBJECT (csrc/fm_torch.cpp): the model makes a fresh
    ``depth[None]`` view per step, so this is one sink per step, reachable by every consumer that is
    handed the same tensor — the Procrustes fit, the fused losses, the softmin sweep's LeadingFrames."""
    sink = depth.__dict__.get("_fm_sink")
    if sink is None:
        sink = depth.__dict__["_fm_sink"] = torch.classes.flowmap_amd.DepthSink()
    return sink


def grad_arena(weights: Tensor):
    """The persistent dL/dweights storage of the PARAMETER behind ``weights`` (GradArena), or None when
    the weights are not a view of a leaf that wants gradients."""
    root = weights if weights._base is None else weights._base
    if not (root.is_leaf and root.requires_grad):
        return None
    arena = root.__dict__.get("_fm_arena")
    if arena is None:
        arena = root.__dict__["_fm_arena"] = torch.classes.flowmap_amd.GradArena()
    return arena


def _root(t: Tensor) -> Tensor:
    return t if t._base is None else t._base


def _note_sparse_grad(weights: Tensor, arena, indices: Tensor, pixels_per_pair: int) -> None:
    """dL/dweights of a planned fit lives in the arena and is zero outside the slots ``indices`` selects in every pair
    (projection.py:226-249 gathers P weights per pair): the parameter carries (arena, flat slot list) so that FusedAdam
    can update those elements only — exactly torch.optim.Adam as long as the moments are zero everywhere else."""
    root = _root(weights)
    if weights.numel() != root.numel() or indices.dim() != 1:
        return

    # images the softmin sweep has added gradient into (random pixels, new every step: LeadingFrames): wholly listed.
    # The list never shrinks — after the hand-over to regressed intrinsics their moments are still non-zero.
    leading = int(root.__dict__.get("_fm_leading_count", 0))

    def build():
        pairs = root.numel() // int(pixels_per_pair)  # one index list (projection.py:227: Int64[" point"]) for every pair
        offsets = torch.arange(pairs, dtype=torch.int64, device=indices.device)[:, None] * int(pixels_per_pair)
        slots = (indices.reshape(1, -1) + offsets).reshape(-1)
        if leading > 0:
            slots = torch.cat([torch.arange(min(leading, pairs) * int(pixels_per_pair), dtype=torch.int64, device=indices.device), slots])
        return indices, torch.unique(slots).contiguous()

    elements = _derived(root, "_fm_sparse_elements", (id(indices), indices._version, int(pixels_per_pair), leading), build)[1]
    root.__dict__["_fm_sparse_grad"] = (arena, elements)


def note_touched(depth: Tensor, consumer: str, pixels: Optional[Tensor]) -> None:
    """A consumer of ``depth`` other than the fused flow loss (the Procrustes fit, the tracking loss) records the
    STATIC set of elements it reads / adds gradient to (flat indices into ``depth``) on the parameter behind it.
    FusedAdam.fuse_depth_update needs the union: those elements are left to the element-list update."""
    root = _root(depth)
    registry = root.__dict__.setdefault("_fm_touched", {})
    if registry.get(consumer) is pixels:
        return
    optimizer = root.__dict__.get("_fm_fused_adam")
    if optimizer is not None and pixels is not None and optimizer.in_pass_pending(root):
        raise RuntimeError(
            f"flowmap_amd: the {consumer} operator reads depth values the flow loss has already updated in this step "
            "(FusedAdam.fuse_depth_update): every consumer of depth must have run once, or announced its pixels with "
            "flowmap_amd._ops.note_touched, before the first fused step")
    if pixels is None:
        registry.pop(consumer, None)
    else:
        registry[consumer] = pixels


def note_leading_frames(depth: Tensor, count: int) -> None:
    """The softmin sweep (intrinsics_softmin.py:85-131) reads `count` leading frames of ``depth`` at pixels it draws anew
    every step: for FusedAdam.fuse_depth_update every element of those frames counts as touched.  count = 0 withdraws."""
    root = _root(depth)
    if root.__dict__.get("_fm_fused_adam") is None or depth.dim() != 4:
        return
    if count == 0:
        if "softmin" in root.__dict__.get("_fm_touched", {}):
            note_touched(depth, "softmin", None)
        return
    total = int(count) * depth.shape[2] * depth.shape[3]
    elements = _derived(root, "_fm_leading_elements", (total, str(depth.device)),
                        lambda: (None, torch.arange(total, dtype=torch.int64, device=depth.device)))[1]
    note_touched(depth, "softmin", elements)


def announce_track_pixels(depth: Tensor, tracks) -> None:
    """The tracking loss while it is still gated off (loss.py:39-41, `enable_after`): its static set of depth taps is
    recorded now, so that an in-pass depth update never has to learn about it in the middle of a step."""
    root = _root(depth)
    if root.__dict__.get("_fm_fused_adam") is None or depth.dim() != 4 or not tracks:
        return
    plan = pack_tracks(tracks, depth.device).scatter_plan(depth.shape[2], depth.shape[3])
    note_touched(depth, "tracking", plan[0])


def touched_elements(depth: Tensor, whole_frames=(), exclude=()):
    """(sorted unique flat indices, per-quad bit mask uint8 (numel/4)) of everything recorded by note_touched, built
    once per combination of recorded sets; None when nothing was recorded or a 4-element quad layout does not apply.
    ``whole_frames`` (frame sharding: the halo frames, indices along dim 1 of a (1, F, H, W) depth): every element of these
    frames is marked in the mask but left OUT of the index list — the caller updates them with a dense pass per frame."""
    root = _root(depth)
    registry = root.__dict__.get("_fm_touched")
    if registry and exclude:
        # ``exclude``: the pixel sets (tensor objects) of consumers whose gradient the flow pass itself absorbs — the ONE tracking loss whose
        # tap gradient the step's DepthSink offers.  By identity, not by consumer name: a second tracking loss on the same depth keeps its pixels.
        registry = {name: v for name, v in registry.items() if not any(v is x for x in exclude)}
    if not registry or depth.numel() % 4 != 0:
        return None
    whole_frames = tuple(sorted(int(f) for f in whole_frames))
    if whole_frames and (depth.dim() != 4 or depth.shape[0] != 1 or (depth.shape[2] * depth.shape[3]) % 4 != 0):
        return None

    def build():
        keys = torch.unique(torch.cat([v.reshape(-1) for v in registry.values()]))
        if whole_frames:
            n = depth.shape[2] * depth.shape[3]
            frame = torch.div(keys, n, rounding_mode="floor")
            keep = torch.ones_like(keys, dtype=torch.bool)
            for f in whole_frames:
                keep &= frame != f
            keys = keys[keep]
        mask = torch.zeros((depth.numel() // 4,), dtype=torch.uint8, device=depth.device)
        quad, bit = torch.div(keys, 4, rounding_mode="floor"), keys % 4
        for e in range(4):
            mask[quad[bit == e]] |= 1 << e
        for f in whole_frames:
            n4 = depth.shape[2] * depth.shape[3] // 4
            mask[f * n4 : (f + 1) * n4] = 15
        return list(registry.values()), keys.contiguous(), mask

    key = tuple((name, id(v), v._version) for name, v in sorted(registry.items())) + (depth.numel(), whole_frames)
    return _derived(root, "_fm_touched_union" + ("_without_" + "_".join(str(id(x)) for x in exclude) if exclude else ""), key, build)[1:]


# (the switches that used to be module variables here — grad arena, one-launch fit, dense-plan choice — are flowmap_amd.config.options)


def _dense_flow_is_rough(bwd_flow: Tensor, h: int, w: int) -> bool:
    """Would the fused dense backward's window (the 32x64 tile displaced by the flow at its centre, +-4 rows / +-8 columns) lose the taps of
    this flow?  Per tile the spread (max - min) of the flow in pixels; a tile is rough when it exceeds 6 rows or 12 columns.  Evaluated once
    per flow tensor with four pooling passes (a few ms at C1 size, one host sync) and kept on the tensor."""

    def build():
        fl = bwd_flow.reshape(-1, h, w, 2).permute(0, 3, 1, 2)  # (pairs, 2, H, W)
        hi = torch.nn.functional.max_pool2d(fl, (32, 64), ceil_mode=True)
        lo = -torch.nn.functional.max_pool2d(-fl, (32, 64), ceil_mode=True)
        spread = hi - lo
        rough = (spread[:, 0] * w > 12.0) | (spread[:, 1] * h > 6.0) | ~torch.isfinite(spread).all(dim=1)
        return bool(rough.float().mean().item() > options.dense_plan_rough_tiles)

    return _derived(bwd_flow, "_fm_dense_rough", (bwd_flow._version, h, w), build)

# which backward path the facades selected (tests)
counters = {"procrustes_planned": 0, "procrustes_dense_planned": 0, "flow_packs": 0, "procrustes_plans_built": 0, "track_tap_samples": 0,
            "flow_tap_passes": 0, "flow_tap_absorbs": 0}


class LeadingFrames:
    """``x[:, :count].contiguous()`` for (b, F, H, W) image stacks (csrc/fm_torch.cpp: LeadingFrames).  The
    softmin sweep reads two of the 150 depth frames; autograd's slice backward would zero-fill a full-size
    tensor and add it densely to the main path's gradient (1.7 GB of traffic at C1).  The node's backward
    runs AFTER the consumers of the intrinsics it helped to produce, so the ``count`` frames are added into
    the buffer the Procrustes fit has already returned for ``x`` (known through x's DepthSink); it falls
    back to the zero-padded tensor whenever that buffer is not known."""

    @staticmethod
    def apply(x: Tensor, count: int) -> Tensor:
        _root(x).__dict__["_fm_leading_count"] = max(int(count), _root(x).__dict__.get("_fm_leading_count", 0))
        return torch_ops().leading_frames(x, int(count), depth_sink(x) if x.dim() == 4 else None)


# --------------------------------------------------------------------------------------
# Pose plumbing
# --------------------------------------------------------------------------------------


class PoseChain:
    """get_extrinsics (flowmap/model/projection.py:187-210)."""

    @staticmethod
    def apply(rel: Tensor) -> Tensor:
        return torch_ops().pose_chain(rel)


class RelativePoses:
    """later(E).inverse() @ earlier(E) and earlier(E).inverse() @ later(E)
    (flowmap/model/projection.py:154,176).  extrinsics (B,F,4,4) -> two (B,F-1,4,4)."""

    @staticmethod
    def apply(ext: Tensor):
        return torch_ops().relative_poses(ext)


class AllPairsPoses(torch.autograd.Function):
    """extrinsics_target.inverse() @ extrinsics_source for all (source, target) pairs
    (flowmap/model/projection.py:288).  (B,f,4,4) -> (B,f,f,4,4) indexed [b, src, tgt]."""

    @staticmethod
    def forward(ctx, ext: Tensor):
        check_device(ext)
        ext = _f32c(ext, "extrinsics")
        b, f = ext.shape[:2]
        rel = torch.empty((b, f, f, 4, 4), dtype=torch.float32, device=ext.device)
        with _guard(ext.device):
            call("fm_allpairs_pose_fwd", ptr(ext), b, f, ptr(rel), stream_for(ext))
        ctx.save_for_backward(ext)
        return rel

    @staticmethod
    def backward(ctx, g_rel: Tensor):
        (ext,) = ctx.saved_tensors
        b, f = ext.shape[:2]
        g_rel = _f32c(g_rel, "grad")
        g_ext = torch.empty_like(ext)
        with _guard(ext.device):
            call("fm_allpairs_pose_bwd", ptr(ext), ptr(g_rel), b, f, ptr(g_ext), stream_for(ext))
        return g_ext


# --------------------------------------------------------------------------------------
# Procrustes fit of adjacent frames
# --------------------------------------------------------------------------------------


def _procrustes_scatter_plan(indices: Tensor, bwd_flow: Tensor, b: int, f: int, h: int, w: int):
    """(pixels, first, vector index per entry, weights, first pixel of every frame) for fm_depth_gather / fm_procrustes_bwd_planned, or None: with constant flows and
    a constant, duplicate-free index set, the pixels the sparse Procrustes gradient touches never change.
    A plan costs a sort, so it is built when the same (indices, flows) come back a second time — per-step
    random indices never qualify.  Kept on the flow tensor, keyed by the index tensor's identity."""
    plans = bwd_flow.__dict__.setdefault("_fm_sparse_plans", {})
    key = (id(indices), indices._version, indices.data_ptr(), indices.numel(), bwd_flow._version, b, f, h, w)
    entry = plans.get(key)
    if entry is None:
        if len(plans) >= 4:
            plans.clear()
        plans[key] = [indices, None, False]  # [the index tensor (kept alive: its id is the key), plan, decided]
        return None
    if not entry[2]:
        entry[2] = True
        points = indices.numel()
        if torch.unique(indices).numel() == points:
            dev = bwd_flow.device
            keys = torch.empty((b * (f - 1) * points * 5,), dtype=torch.int64, device=dev)
            weights = torch.empty((keys.numel(),), dtype=torch.float32, device=dev)
            lay, any_view = _layout_array(bwd_flow)
            with _guard(dev):
                if any_view:
                    call("fm_procrustes_scatter_plan_views", ptr(bwd_flow), ptr(indices), points, b, f, h, w, ptr(keys), ptr(weights),
                         ctypes.addressof(lay), stream_for(bwd_flow))
                else:
                    call("fm_procrustes_scatter_plan", ptr(bwd_flow), ptr(indices), points, b, f, h, w, ptr(keys), ptr(weights), stream_for(bwd_flow))
            used = torch.nonzero(keys >= 0).reshape(-1)
            sorted_keys, order = torch.sort(keys[used], stable=True)
            entries = used[order]
            pixels, counts = torch.unique_consecutive(sorted_keys, return_counts=True)
            first = torch.zeros((pixels.numel() + 1,), dtype=torch.int32, device=dev)
            first[1:] = torch.cumsum(counts, 0).to(torch.int32)
            vectors = (torch.div(entries, 5, rounding_mode="floor") * 2 + (entries % 5 == 4)).to(torch.int32)
            # where every frame's pixels begin (the plan is sorted by frame·H·W + pixel): fm_procrustes_bwd_planned's blocks, one per frame
            bounds = torch.arange(b * f + 1, dtype=torch.int64, device=dev) * (h * w)
            frame_first = torch.searchsorted(pixels, bounds).to(torch.int32)
            # the static taps of every correspondence for the forward fit (fm_procrustes_fit_chain's tap_records): slots 0..3 of the
            # plan as pixel offsets inside the earlier frame (int32 bits, -1 outside) + their bilinear weights
            per = keys.reshape(b * (f - 1), points, 5)
            base = (torch.arange(b * (f - 1), dtype=torch.int64, device=dev) // (f - 1) * f + torch.arange(b * (f - 1), dtype=torch.int64, device=dev) % (f - 1)) * (h * w)
            offsets = torch.where(per[:, :, :4] >= 0, per[:, :, :4] - base[:, None, None], torch.full_like(per[:, :, :4], -1)).to(torch.int32)
            tap_records = torch.cat([offsets.view(torch.float32), weights.reshape(b * (f - 1), points, 5)[:, :, :4]], dim=2).contiguous()
            entry[1] = (pixels.contiguous(), first, vectors.contiguous(), weights[entries].contiguous(), frame_first.contiguous(), tap_records)
            counters["procrustes_plans_built"] += 1
    return entry[1]


def _dense_procrustes_plan(bwd_flow: Tensor, b: int, f: int, h: int, w: int):
    """(first int64 (pairs*tiles + 1), list uint32-as-int32 (entries)) of fm_procrustes_dense_plan: for every tile of
    every pair's earlier frame, the later pixels whose bilinear taps land in it.  The flows are constants
    of the optimisation, so this is built once per flow tensor and kept ON that tensor (it lives and dies
    with it; an in-place edit bumps the version and the plan is rebuilt)."""

    def build():
        tiles = ctypes.c_int(0)
        call("fm_procrustes_dense_tiles", h, w, ctypes.addressof(tiles))
        dev = bwd_flow.device
        slots = b * (f - 1) * tiles.value
        with _guard(dev):
            st = stream_for(bwd_flow)
            counts = torch.zeros((slots,), dtype=torch.int32, device=dev)
            call("fm_procrustes_dense_plan", ptr(bwd_flow), b, f, h, w, ptr(counts), None, None, st)
            first = torch.zeros((slots + 1,), dtype=torch.int64, device=dev)
            torch.cumsum(counts, 0, out=first[1:])
            total = int(first[-1].item())  # one host sync, when the plan is built
            entries = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
            counts.zero_()
            call("fm_procrustes_dense_plan", ptr(bwd_flow), b, f, h, w, ptr(counts), ptr(first), ptr(entries), st)
        return first, entries

    return _derived(bwd_flow, "_fm_dense_plan", (bwd_flow._version, b, f, h, w), build)


class ProcrustesFit:
    """align_surfaces up to (not including) the pose chain (projection.py:213-249) with
    align_rigid (procrustes.py:7-51) inside (csrc/fm_torch.cpp: ProcrustesFit).  Source of xyz is either

      depth (B,F,H,W) + intrinsics (B,F,3,3)   [surfaces never materialised], or
      surfaces (B,F,H,W,3).

    ``weight_sens != 0``: ``weights`` holds LOGITS and w = sigmoid(weight_sens·logit) is
    evaluated at the gathered points only (BackboneExplicitDepth fused into the gather);
    the returned gradient is then w.r.t. the logits.

    Returns the "inverse relative transformations" (B,F-1,4,4): later -> earlier camera, and
    their rigid inverses (earlier -> later camera) for consumers that want both directions
    without going through the pose chain.

    This facade adds what the operator takes explicitly: K^-1, the step's DepthSink, the weight
    parameter's GradArena and the static backward plans (planned gather for a constant sparse index
    set, tile lists for the dense case)."""

    @staticmethod
    def apply(depth, k, surfaces, weights, bwd_flow, indices, weight_sens=0.0, batch_repeat=1):
        t_bwd, t_fwd, _ext = ProcrustesFit.apply_chained(depth, k, surfaces, weights, bwd_flow, indices, weight_sens, batch_repeat, chain=False)
        return t_bwd, t_fwd

    @staticmethod
    def apply_chained(depth, k, surfaces, weights, bwd_flow, indices, weight_sens=0.0, batch_repeat=1, chain=True, want_extrinsics=True):
        """-> (t_bwd, t_fwd, extrinsics or None).  ``chain``: let the fit's own launch chain the poses into the
        extrinsics (get_extrinsics, projection.py:187-210) when it can — a sparse index set, no repeat; the
        persistent, self-cleaning workspace this needs is kept on the flow tensor.  ``want_extrinsics = False``: the one-launch fit
        without the chain — the caller holds the relative poses and chains them if and when something asks (LazyExtrinsics)."""
        from_depth = surfaces is None
        rep = int(batch_repeat)
        kinv = sink = wsink = arena = None
        sparse = (None, None, None, None, None, None)
        dense = (None, None)
        if from_depth:
            if depth is None or k is None:
                raise RuntimeError("flowmap_amd: the Procrustes fit needs depth + intrinsics or surfaces")
            kinv = intrinsics_inverse(k)
            wants_grad = torch.is_grad_enabled() and depth.requires_grad and rep == 1
            static_flow = bwd_flow.dtype == torch.float32 and frame_window_layout(bwd_flow) is not None and depth.dim() == 4 and bwd_flow.dim() == 5
            if rep == 1:
                sink = depth_sink(depth)
            if wants_grad and static_flow:
                b, f, h, w = depth.shape
                if indices is None:
                    note_touched(depth, "procrustes", None)  # every pixel is a correspondence: nothing is left to an in-pass update
                    planned = _dense_flow_is_rough(bwd_flow, h, w) if options.dense_plan is None else bool(options.dense_plan)
                    if planned and h <= 65535 and w <= 65535 and bwd_flow.is_contiguous():
                        dense = _dense_procrustes_plan(bwd_flow, b, f, h, w)
                        counters["procrustes_dense_planned"] += 1
                elif indices.dtype == torch.int64 and indices.is_contiguous():
                    plan = _procrustes_scatter_plan(indices, bwd_flow, b, f, h, w)
                    note_touched(depth, "procrustes", None if plan is None else plan[0])
                    if plan is not None:
                        sparse = plan
                        counters["procrustes_planned"] += 1
                        if options.grad_arena and torch.is_tensor(weights) and weights.requires_grad:
                            arena = grad_arena(weights)
                            _note_sparse_grad(weights, arena, indices, h * w)
            wsink = weights.__dict__.get("_fm_sink")  # exists when the softmin sweep took LeadingFrames of the weights
        work = None
        if chain and options.fit_chain and rep == 1 and indices is not None and bwd_flow.dim() == 5:
            pairs = bwd_flow.shape[0] * bwd_flow.shape[1]
            work = _derived(bwd_flow, "_fm_fit_work", (pairs, str(bwd_flow.device)),
                            lambda: torch.zeros((pairs * STAT_STRIDE + (pairs + 2) // 2 + 1,), dtype=torch.float64, device=bwd_flow.device))
        t_bwd, t_fwd, ext = torch_ops().procrustes_fit(depth, k, kinv, surfaces, weights, bwd_flow, indices, float(weight_sens), rep, sink, wsink,
                                                       arena, *sparse, *dense, work, bool(want_extrinsics))
        return t_bwd, t_fwd, (ext if ext.numel() > 0 else None)


# --------------------------------------------------------------------------------------
# Fused flow loss
# --------------------------------------------------------------------------------------


def flow_kernel_timing(enable: bool) -> None:
    """bench.py: record HIP events on the launch stream around every launch of the fused flow kernel."""
    torch_ops().flow_timing_enable(bool(enable))


def flow_kernel_times(tracking: bool = False):
    """Milliseconds of every fused flow kernel launch (``tracking``: of every fm_track_loss_fwd call, i.e.
    track_pairs + its reduction) since the last call (synchronise first)."""
    return list(torch_ops().flow_timing_collect(bool(tracking)))


def flow_valid_norm(mask_fwd: Tensor, mask_bwd: Tensor, weight: float, reducer=None) -> Tensor:
    """Device tensor [weight/(V or 1), (V or 1)] with V = Σmask_fwd + Σmask_bwd
    (loss_flow.py:56,66,70).  Masks are constants of the optimisation, so the result is kept on the
    forward-mask tensor (per backward mask, weight and reducer) and costs nothing after the first step.
    ``reducer`` (frame sharding) maps the local fp64 sum to the global one."""
    released = mask_fwd.__dict__.get("_fm_released_norms")
    if released is not None:  # release_flow_originals: the masks are gone, the normalisers computed from them are not
        hit = released.get((float(weight), id(reducer)))
        if hit is None:
            raise RuntimeError("flowmap_amd: the flow masks were released (release_flow_originals) before a loss with this weight / reducer "
                               "had computed its normaliser from them")
        return hit

    def build():
        vsum = torch.empty((1,), dtype=torch.float64, device=mask_fwd.device)
        norm = torch.empty((2,), dtype=torch.float32, device=mask_fwd.device)
        lay, any_view = _layout_array(mask_fwd, mask_bwd) if mask_fwd.dim() == 4 else (None, False)
        mf, mb = (mask_fwd, mask_bwd) if (lay is not None or mask_fwd.dim() != 4) else (mask_fwd.contiguous(), mask_bwd.contiguous())
        if mask_fwd.dim() != 4:
            mf, mb = mask_fwd.contiguous(), mask_bwd.contiguous()
        with _guard(mask_fwd.device):
            if any_view:  # frame windows of larger tensors: summed in place
                b_, pairs_, h_, w_ = mask_fwd.shape
                call("fm_flow_valid_norm_views", ptr(mf), ptr(mb), b_, pairs_, h_ * w_, float(weight), ptr(vsum), ptr(norm),
                     ctypes.addressof(lay), stream_for(mask_fwd))
            else:
                call("fm_flow_valid_norm", ptr(mf), ptr(mb), mf.numel(), float(weight), ptr(vsum), ptr(norm), stream_for(mask_fwd))
        if reducer is not None:
            total = reducer(vsum)
            veff = torch.where(total == 0, torch.ones_like(total), total)
            norm = torch.cat([float(weight) / veff, veff]).to(torch.float32)
        mask_fwd.__dict__.setdefault("_fm_norms_seen", {})[(float(weight), id(reducer))] = norm
        return (mask_bwd, reducer, norm)  # the other mask and the reducer are kept alive: their ids are part of the key

    key = (mask_fwd._version, id(mask_bwd), mask_bwd._version, mask_bwd.data_ptr(), tuple(mask_fwd.shape), float(weight), id(reducer))
    return _derived(mask_fwd, "_fm_norm", key, build)[2]




def packed_flow_inputs(flow_fwd: Tensor, flow_bwd: Tensor, mask_fwd: Tensor, mask_bwd: Tensor, eager: bool = False) -> Optional[Tensor]:
    """Flows + masks in the layout of fm_flow_pack_inputs, or None when it does not apply
    (width not a multiple of 4, unexpected shapes / dtypes) — or not YET: the re-layout is a full read + write pass over
    the flows and masks (+3.3 GB resident at C1), worth it only for inputs that come back.  In an overfit loop they are
    constants (flow_predictor.py:82-102 runs once per video) and the SECOND step that brings the same four tensors packs
    them; in the reference's pretraining loop every step brings a new ``Flows`` (model_wrapper_pretrain.py:46-71) and
    nothing is ever packed — the fused kernel streams the caller's tensors (its un-packed instance).  ``eager``: pack at
    first sight.  Kept on the forward-flow tensor and validated against the identity and version of all four tensors."""
    if not options.packed_inputs:
        return None
    released = flow_fwd.__dict__.get("_fm_released_packed")
    if released is not None:
        return released
    srcs = (flow_fwd, flow_bwd, mask_fwd, mask_bwd)
    if mask_fwd.dim() != 4 or mask_fwd.shape[-1] % 4 != 0 or tuple(mask_bwd.shape) != tuple(mask_fwd.shape):
        return None
    if tuple(flow_fwd.shape) != (*mask_fwd.shape, 2) or tuple(flow_bwd.shape) != (*mask_fwd.shape, 2):
        return None
    if any(t.dtype != torch.float32 or t.data_ptr() % 16 != 0 for t in srcs):
        return None
    lay, any_view = _layout_array(*srcs)
    if lay is None or any(lay[i].frame_stride % 4 or lay[i].batch_stride % 4 for i in range(4)):
        return None
    key = tuple((id(t), t._version, t.data_ptr()) for t in srcs) + (tuple(mask_fwd.shape),)
    if not (eager or options.pack_on_first_sight):
        slot = flow_fwd.__dict__.get("_fm_packed")
        if (slot is None or slot[0] != key) and flow_fwd.__dict__.get("_fm_packed_seen") != key:
            flow_fwd.__dict__["_fm_packed_seen"] = key  # first sighting: remember, stream directly
            return None

    def build():
        b, pairs, h, w = mask_fwd.shape
        chunks = (h * w // 4 + 63) // 64
        packed = torch.empty((b * (pairs + 1), chunks, 6, 64, 4), dtype=torch.float32, device=mask_fwd.device)
        with _guard(mask_fwd.device):
            if any_view:
                call("fm_flow_pack_inputs_views", ptr(flow_fwd), ptr(flow_bwd), ptr(mask_fwd), ptr(mask_bwd), b, pairs + 1, h, w, ptr(packed),
                     ctypes.addressof(lay), stream_for(mask_fwd))
            else:
                call("fm_flow_pack_inputs", ptr(flow_fwd), ptr(flow_bwd), ptr(mask_fwd), ptr(mask_bwd), b, pairs + 1, h, w, ptr(packed),
                     stream_for(mask_fwd))
        counters["flow_packs"] += 1
        return (srcs[1:], packed)  # the three other tensors are kept alive: their ids are part of the key

    return _derived(flow_fwd, "_fm_packed", key, build)[1]


def release_flow_originals(flows) -> int:
    """After the flows and masks of an overfit loop have been packed (the second step), the fused flow loss reads only the
    packed copy: the forward flow and the two masks (2.2 of the 3.3 GB at C1; the backward flow stays — the Procrustes fit
    samples it) can be given back.  Replaces ``flows.forward``, ``flows.forward_mask`` and ``flows.backward_mask`` by
    storage-free placeholders of the same shape that carry the packed copy and the loss normalisers computed so far.
    Returns the bytes released.  The placeholders are only good for the fused flow loss: ``flows`` no longer holds the data."""
    ff, fb, mf, mb = flows.forward, flows.backward, flows.forward_mask, flows.backward_mask
    slot = ff.__dict__.get("_fm_packed")
    if slot is None:
        raise RuntimeError("flowmap_amd: nothing to release — these flows have not been packed yet (the fused flow loss packs them at its second step)")
    packed = slot[1][1]
    freed = (ff.numel() + mf.numel() + mb.numel()) * 4

    def placeholder(t: Tensor) -> Tensor:
        return torch.zeros((1,), dtype=t.dtype, device=t.device).expand(t.shape)

    new_ff, new_mf, new_mb = placeholder(ff), placeholder(mf), placeholder(mb)
    new_ff.__dict__["_fm_released_packed"] = packed
    new_mf.__dict__["_fm_released_norms"] = dict(mf.__dict__.get("_fm_norms_seen", {}))
    for name in ("_fm_flow_acc",):
        if name in mf.__dict__:
            new_mf.__dict__[name] = mf.__dict__[name]
    flows.forward, flows.forward_mask, flows.backward_mask = new_ff, new_mf, new_mb
    return freed


class FlowLossFused:
    """weight · LossFlow.compute_unweighted_loss (flowmap/loss/loss_flow.py:31-70,
    flowmap/loss/loss.py:47) evaluated from depth + intrinsics + relative poses, with the
    analytic gradient of every input produced in the same HBM pass (csrc/fm_torch.cpp: FlowLossFused).
    ``carry``: let the dense dL/ddepth travel through the step's DepthSink to the Procrustes fit's
    node (which returns it once, summed with the sparse parts) when the poses come from that fit."""

    @staticmethod
    def apply(depth, k, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, kind, delta, carry, items, packed=None):
        kinv = intrinsics_inverse(k)
        sink = depth_sink(depth) if carry else None
        # the per-(frame, direction) fp64 sums: a workspace kept on the mask tensor, zeroed once (the finalize launch leaves it zero)
        size = depth.shape[0] * depth.shape[1] * 2 * FLOW_ACC_STRIDE if depth.dim() == 4 else 0
        acc = _derived(mask_fwd, "_fm_flow_acc", (size, str(depth.device)), lambda: torch.zeros((size,), dtype=torch.float64, device=depth.device))
        adam, ticket = (None, None, None, 0, [], None), None
        optimizer = _root(depth).__dict__.get("_fm_fused_adam")
        # the tap exchange with the tracking loss (its static taps were registered with the parameter by TrackLossFused): this pass leaves the
        # depth at every tap in the plan's compact image and absorbs the tracking gradient a look-ahead evaluation left in the sink
        tap_plan = tap_plan_of(depth) if (sink is not None and torch.is_grad_enabled() and depth.requires_grad) else None
        if tap_plan is not None and packed is None and not all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (flow_fwd, flow_bwd, mask_fwd, mask_bwd)):
            tap_plan = None  # (frame windows / unaligned flows: the pass that reads them in place has no tap variant)
        if tap_plan is not None and int(items) > 6:
            tap_plan = None  # (the tap variant of the pass stages at most 6 quads per thread: fm_flow_loss_fused_taps)
        absorbing = tap_plan is not None and sink.offers_taps()
        if optimizer is not None and sink is not None and torch.is_grad_enabled():
            # (a tracking gradient this pass absorbs completes dL/ddepth at the taps: they need not wait for the element-list update)
            offer = optimizer.begin_in_pass(depth, sink, t_fwd, t_bwd, exclude=(tap_plan.pixels,) if absorbing else ())
            if offer is not None:
                adam, ticket = offer
        # frame sharding with an early halo exchange (FrameShard.enable_early_halo): the dense dL/ddepth exists at the end of THIS
        # forward pass — the boundary frames are sent now, under the rest of the step; only a sparse correction follows backward
        early = _root(depth).__dict__.get("_fm_early_halo") if (sink is not None and torch.is_grad_enabled()) else None
        if early is not None and ticket is not None and not early.ghost_halo_enabled():
            early = None  # (the early DENSE exchange is not combined with the in-pass Adam update; the ghost halo is: it reads the boundary frames only)
        if early is not None:
            sink.request_early_dense(early.unit_flag(depth.device))
        # (verified only while the version counter still vouches for the image: after a regular update the image was simply out of date, and
        # whoever sampled it did so before that update)
        # ... and at the first sampled step and every 64th only: the check costs the pass one more load per tap
        verify = (tap_plan is not None and tap_plan.sampled_now and ticket is None and (tap_plan.samples == 1 or tap_plan.samples % 64 == 0)
                  and tap_plan.image_valid_for(_root(depth)))
        taps = (None, None, None, None) if tap_plan is None else (tap_plan.chunk_base, tap_plan.pixel_in_frame, tap_plan.image, tap_plan.stale_flag if verify else None)
        offered = absorbing
        loss = torch_ops().flow_loss(depth, k, kinv, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, packed, int(kind), float(delta),
                                     sink, int(items), acc, *adam, *taps, bool(absorbing and ticket is not None))
        if tap_plan is not None:
            counters["flow_tap_passes"] += 1
            counters["flow_tap_absorbs"] += int(offered and sink.tap_absorbed())
            if ticket is None and loss.requires_grad:
                tap_plan.image_slots = tap_plan.slots
                tap_plan.tag(_root(depth))  # depth as this pass read it: good until the parameter moves
            elif ticket is not None:
                # an in-pass Adam update: the image holds the updated depth except at the pixels other operators keep (the ticket's element
                # list), which step() updates afterwards — it becomes valid, read around those, when step() has finished (FusedAdam.step)
                tap_plan.invalidate()
                tap_plan.image_slots = tap_plan.slots_reading_around(ticket[2])
                tap_plan.pending_in_pass = True
            else:
                tap_plan.invalidate()
            tap_plan.sampled_now = False
            if verify:
                capturing = depth.is_cuda and torch.cuda.is_current_stream_capturing()
                if not capturing:
                    tap_plan.check_stale()
        if ticket is not None:  # the operator accepted the arguments and launched: only now does the optimiser's state advance
            optimizer.commit_in_pass(ticket)
        if early is not None:
            dense = sink.take_early_dense()
            if dense is not None:
                early.start_early_halo(dense, _root(depth), (t_fwd, t_bwd, k, kinv, norm, int(kind), float(delta)))
        return as_root_loss(loss)


def softmin_intrinsics(depth, weights, bwd_flow, indices, candidate_k, rel, weight_sens, frames):
    """-> (K (b,frames,3,3) blended over the candidates, softmin weights (b,n)): SoftminScore followed by the tail of
    IntrinsicsSoftmin.forward (intrinsics_softmin.py:105-141), csrc/fm_torch.cpp: SoftminIntrinsics.  K⁻¹ comes out of
    the same launch and is left where intrinsics_inverse() finds it."""
    k, soft, kinv = torch_ops().softmin_intrinsics(depth, weights, bwd_flow, indices, candidate_k, rel, float(weight_sens), int(frames))
    _attach_inverse(k, kinv)
    return k, soft


# hipGraph capture (flowmap_amd.graph.GraphedStep): a captured launch cannot carry a host value that
# changes between replays, so while this is on random_subset keeps its seed in device memory.
graph_capturable = False
_rng_states: dict = {}


def random_subset(n: int, count: int, device, seed: Optional[int] = None) -> Tensor:
    """``count`` distinct pseudo-random indices of [0, n) in pseudo-random order (int64) — what
    ``torch.randperm(n, device=device)[:count]`` is used for — from one launch of fm_random_subset.
    ``seed`` defaults to a draw from torch's CPU generator, so ``torch.manual_seed`` reproduces it.
    With ``graph_capturable`` the seed is a per-device state tensor that every call advances."""
    if not 1 <= count <= n:
        raise RuntimeError("flowmap_amd: random_subset needs 1 <= count <= n")
    device = torch.device(device)
    if seed is None and graph_capturable:
        state = _rng_states.get(str(device))
        if state is None:  # created OUTSIDE any capture (GraphedStep warms the step up first)
            first = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
            state = _rng_states[str(device)] = torch.tensor([first], dtype=torch.int64).to(device)
        return torch_ops().random_subset(int(n), int(count), device, 0, state)
    if seed is None:
        seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    return torch_ops().random_subset(int(n), int(count), device, int(seed), None)


# The one-off data preparation (flow / mask post-processing, resize + crop), the function-level operators on explicit point sets and the
# tracking loss's static data live in modules of their own; their names stay reachable here.
from ._functions import AlignRigid, BilinearSample, Reproject, RobustMapping, Unproject  # noqa: E402,F401
from ._preprocess import consistency_mask, flow_postprocess, resize_crop  # noqa: E402,F401
from ._tracks import PackedTracks, TapPlan  # noqa: E402,F401


# (tap image / tap exchange / unit seed / calling-thread backward: flowmap_amd.config.options)
_unit_seeds: dict = {}


class RootLoss(Tensor):
    """A scalar loss that seeds its own ``backward()``.  ``loss.backward()`` normally makes autograd fill a fresh ``ones_like(loss)`` (a
    launch) and hand it down, and a fused loss then has to find out on the device that its upstream gradient is 1 before it releases the
    gradients it already holds (``fm_scale_if_needed``: another launch, which does nothing).  Here ``backward()`` without an explicit
    gradient passes a ones tensor that was made once per device and registered with the operators (``register_unit_seed``): a node that
    receives exactly that tensor — same memory, never written — knows its upstream gradient on the host.

    Like ``nn.Parameter`` the class switches ``__torch_function__`` off: operators and attribute reads on a RootLoss cost what they cost on
    a plain tensor (with the default ``__torch_function__`` every ``.dim()`` / ``.dtype`` / ``+`` went through Python: 0.1 ms per step where
    the host is the bottleneck) and RETURN plain tensors — ``detach()``, ``torch.stack`` of logged values, whatever a caller computes from
    the loss.  Only the arithmetic a training step does with its losses keeps the class, through the operators defined here: sums
    (``sum(losses)``, ``a + b``), differences, a weight or a divisor (which reach the fused node as an ordinary gradient and are handled
    as before).  ``torch.autograd.grad`` / an explicit ``gradient=`` / ``create_graph=True`` take autograd's usual path."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if gradient is None and not create_graph and options.unit_seed and self.dim() == 0 and self.dtype == torch.float32:
            gradient = unit_seed(self.device)
        if options.backward_on_calling_thread and not create_graph:
            # autograd hands the nodes of GPU tensors to a worker thread per device and waits for it: two thread wake-ups around a backward
            # pass whose nodes only ENQUEUE a handful of kernels — where the host is the bottleneck (the reference's default resolution)
            # they are a measurable part of the step.  The nodes of a fused loss run just as well on the calling thread.
            with torch.autograd.set_multithreading_enabled(False):
                return torch.autograd.backward(self, gradient, retain_graph, create_graph, inputs=inputs)
        return torch.autograd.backward(self, gradient, retain_graph, create_graph, inputs=inputs)

    def __add__(self, other):
        return as_root_loss(Tensor.__add__(self, other))

    def __radd__(self, other):  # (sum(losses) starts from 0 + loss)
        return as_root_loss(Tensor.__radd__(self, other))

    def __sub__(self, other):
        return as_root_loss(Tensor.__sub__(self, other))

    def __rsub__(self, other):
        return as_root_loss(Tensor.__rsub__(self, other))

    def __mul__(self, other):
        return as_root_loss(Tensor.__mul__(self, other))

    def __rmul__(self, other):
        return as_root_loss(Tensor.__rmul__(self, other))

    def __truediv__(self, other):
        # (a trainer normalises the loss by its accumulation factor — Lightning: `closure_loss / accumulate_grad_batches`, 1 by default — and calls
        # backward() on the quotient: dividing by the number 1 is the loss itself, and the fused nodes keep their host-known seed)
        if type(other) in (int, float) and other == 1:
            return self
        return as_root_loss(Tensor.__truediv__(self, other))

    def __neg__(self):
        return as_root_loss(Tensor.__neg__(self))

    def __reduce_ex__(self, protocol):  # saved / deep-copied as the plain tensor it is (torch.load(weights_only=True) knows no RootLoss)
        return self.as_subclass(Tensor).__reduce_ex__(protocol)

    def __format__(self, format_spec):  # (torch.Tensor.__format__ formats the VALUE of a 0-d tensor only when type(self) is Tensor)
        if self.dim() == 0 and not self.is_meta:
            return self.item().__format__(format_spec)
        return object.__format__(self, format_spec)


def unit_seed(device) -> Tensor:
    seed = _unit_seeds.get(device)
    if seed is not None and seed._version == 0:
        return seed
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    seed = _unit_seeds.get(device)
    if seed is None or seed._version != 0:
        seed = torch.ones((), dtype=torch.float32, device=device)
        torch_ops().register_unit_seed(seed)
        _unit_seeds[device] = seed
    return seed


def as_root_loss(loss):
    if type(loss) is not Tensor or not options.unit_seed or not loss.requires_grad or loss.dim() != 0:
        return loss  # (also NotImplemented from a reflected operator)
    if loss.device not in _unit_seeds:
        unit_seed(loss.device)  # made outside any later graph capture
    # the SAME tensor object, re-typed (both classes are plain Python subclasses of torch._C.TensorBase with one layout): `as_subclass` would
    # make an alias — a second tensor and an AliasBackward node for the engine to walk in every backward
    try:
        loss.__class__ = RootLoss
        return loss
    except TypeError:
        return loss.as_subclass(RootLoss)


def _whole_parameter(depth: Tensor) -> Optional[Tensor]:
    """The leaf parameter ``depth`` (1, F, H, W) is a whole, dense view of — or None."""
    root = _root(depth)
    if depth.dim() != 4 or depth.shape[0] != 1 or not depth.is_contiguous() or depth.dtype != torch.float32:
        return None
    if depth.data_ptr() != root.data_ptr() or depth.numel() != root.numel() or not root.is_contiguous():
        return None
    return root


def tap_plan_of(depth: Tensor) -> Optional[TapPlan]:
    """The TapPlan a tracking loss registered for the parameter behind ``depth`` (matching its shape), if any."""
    if not options.tap_exchange or depth.numel() * 4 < options.tap_exchange_min_bytes:
        return None
    root = _whole_parameter(depth)
    plan = root.__dict__.get("_fm_tap_plan") if root is not None else None
    if plan is None or plan.key != tuple(int(d) for d in depth.shape[1:]) or plan.chunk_base.device != depth.device or depth.data_ptr() % 16 != 0:
        return None
    return plan


def _local_pixels(self, height: int, width: int, frame0: int):
    """The scatter plan's touched pixels as flat indices into a depth WINDOW that starts at frame `frame0` (frame sharding) — one
    tensor object per window, so that consumers comparing identities (FusedAdam.fuse_depth_update, FrameShard.enable_early_halo)
    see a constant set."""
    plan = self.scatter_plan(height, width)
    if plan is None:
        return None
    if frame0 == 0:
        return plan[0]
    key = (int(height), int(width), int(frame0))
    cache = self.__dict__.setdefault("_local", {})
    if key not in cache:
        cache[key] = plan[0] - int(frame0) * int(height) * int(width)
    return cache[key]


PackedTracks.local_pixels = _local_pixels


def pack_tracks(tracks, device, own=None) -> PackedTracks:
    """The packed form of a track list, built once: kept on the first segment's coordinate tensor and
    validated against every segment's identity and version."""
    # (the same list object step after step: compare what could have changed — the segments' tensors, their version counters, their start
    # frames — against the last call's snapshot without building the full key: 6 us instead of 20 per step where the host is the bottleneck)
    first = tracks[0].xy
    last = first.__dict__.get("_fm_packed_tracks_last")
    if last is not None and last[0] is tracks and last[1] == (device, own) and len(tracks) == len(last[2]):
        for t, (xy, xy_version, vis, vis_version, start) in zip(tracks, last[2]):
            if t.xy is not xy or xy._version != xy_version or t.visibility is not vis or vis._version != vis_version or t.start_frame != start:
                break
        else:
            return last[3]
    key = tuple((id(t.xy), t.xy._version, id(t.visibility), t.visibility._version, int(t.start_frame), tuple(t.xy.shape)) for t in tracks)
    key += (str(device), own)
    packed = _derived(first, "_fm_packed_tracks", key, lambda: (list(tracks), PackedTracks(tracks, device, own)))[1]
    first.__dict__["_fm_packed_tracks_last"] = (tracks, (device, own), [(t.xy, t.xy._version, t.visibility, t.visibility._version, t.start_frame) for t in tracks], packed)
    return packed


class TrackLossFused:
    """weight · LossTracking.compute_unweighted_loss (flowmap/loss/loss_tracking.py:28-61,
    flowmap/loss/loss.py:47) over all segments, from depth + intrinsics + extrinsics
    (csrc/fm_torch.cpp: TrackLossFused).

    Frame sharding (flowmap_amd/sharding.py): ``depth`` holds the rank's frames from ``frame0`` on,
    ``k`` / ``ext`` the whole video; ``packed`` was built with this rank's ``own`` source range;
    ``reducer`` sums the fp64 pair [Σρ, count] over the ranks.  The result is then the GLOBAL loss,
    the gradients this rank's share of it (autograd / FrameShard.sync sum them)."""

    @staticmethod
    def apply(depth, k, ext, packed: PackedTracks, weight, kind, delta, defer, frame0=0, reducer=None, fit_from=None, offer_taps=False):
        """``offer_taps``: the caller guarantees that the fused flow loss of this step runs on the same depth tensor right after this call
        (LossFlow's look-ahead): dL/ddepth is compacted at the static taps and left in the step's DepthSink for that pass to absorb."""
        check_device(depth, k, ext, packed.xy)
        kinv = intrinsics_inverse(k)
        needs_depth = torch.is_grad_enabled() and depth.requires_grad
        plan = packed.scatter_plan(depth.shape[2], depth.shape[3]) if needs_depth and depth.dim() == 4 else None  # built at the first step
        if plan is not None:
            pixels = packed.local_pixels(depth.shape[2], depth.shape[3], int(frame0))
            known = _root(depth).__dict__.get("_fm_touched", {}).get("tracking")
            # (a second tracking loss with its own track set on the same depth registers beside the first, not over it)
            note_touched(depth, "tracking" if (known is None or known is pixels) else f"tracking@{id(packed)}", pixels)
        sink = depth_sink(depth) if defer else None
        # the tap exchange: whole video local, gradients on — register the static tap set with the parameter (the flow pass then leaves the tap
        # depths in its compact image) and sample from that image while the parameter has not moved since
        taps = (None, None, None)
        root = _whole_parameter(depth) if (options.tap_exchange and plan is not None and reducer is None and int(frame0) == 0 and defer
                                           and depth.numel() * 4 >= options.tap_exchange_min_bytes) else None
        if root is not None and root.__dict__.get("_fm_tap_exchange_off"):
            root = None
        if root is not None and root.is_leaf and ext.shape[1] == depth.shape[1]:
            tap_plan = packed.tap_plan(depth.shape[1], depth.shape[2], depth.shape[3])
            registered = root.__dict__.get("_fm_tap_plan")
            owner = root.__dict__.get("_fm_tap_owner_step")
            if tap_plan is not None and registered is not None and registered is not tap_plan and owner is not None and owner() is depth:
                # TWO tracking losses with their own track sets on one depth tensor in one step: the exchange is built around ONE static tap set
                # per parameter (one compact image, one absorbed gradient) — it is switched off for this parameter for good; both run as in round 3
                root.__dict__["_fm_tap_exchange_off"] = True
                root.__dict__.pop("_fm_tap_plan", None)
                root.__dict__.pop("_fm_tracking_follows_flow", None)
                tap_plan, offer_taps = None, False
            if tap_plan is not None:
                root.__dict__["_fm_tap_plan"] = tap_plan
                root.__dict__["_fm_tap_owner_step"] = weakref.ref(depth)  # (the step's depth tensor: a second registration within the same step is a second loss)
                taps = (tap_plan.slots, None, tap_plan.shared_ranks)
                if options.tap_image and tap_plan.image_valid_for(root):
                    taps = (tap_plan.image_slots, tap_plan.image, tap_plan.shared_ranks)
                    tap_plan.note_sampled()
                    counters["track_tap_samples"] += 1
                else:
                    tap_plan.sampled_now = False
        else:
            offer_taps = False
        loss, scale, totals = torch_ops().track_loss(depth, k, kinv, ext, packed.xy, packed.vis, packed.seg, packed.blocks, packed.tiles,
                                                     packed.counts, float(weight), int(kind), float(delta), sink, int(frame0),
                                                     *(plan if plan is not None else (None, None, None, None)), fit_from, *taps, bool(offer_taps))
        if reducer is not None:
            # the operator's gradients follow the `scale` tensor they find at backward time: overwrite it with the
            # global normaliser and report the global value through the local node
            total = reducer(totals)
            den = torch.where(total[1] == 0, torch.ones_like(total[1]), total[1])  # `valid_sum or 1` (loss_tracking.py:61)
            scale.copy_(torch.stack([float(weight) / den, total[1]]).to(torch.float32))
            value = (float(weight) * total[0] / den).to(torch.float32)
            loss = loss + (value - loss).detach()
        return as_root_loss(loss)
