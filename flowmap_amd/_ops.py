"""torch.autograd.Function wrappers around the C ABI (include/flowmap_hip.h).

These are the differentiable building blocks the reference-shaped call surface
(flowmap_amd.model.projection, flowmap_amd.loss) is assembled from.  Every Function
launches hand-written HIP kernels on the current stream through ctypes; none of them
synchronises the device or falls back to eager PyTorch math.
"""

from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
from torch import Tensor

from ._lib import call, check_device, ptr, stream_for

MAPPING_KINDS = {"huber": 0, "l1": 1, "l2": 2}

FLOW_ACC_STRIDE = 20
STAT_STRIDE = 16
AUX_STRIDE = 32
PAIR_GRAD_STRIDE = 20
DENSE_CONST_STRIDE = 40  # FM_DENSE_CONST_STRIDE
TRACK_TILE = 6  # FM_TRACK_TILE (include/flowmap_hip.h; tests/test_abi.py checks they agree)


def _f32c(t: Tensor, what: str) -> Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"flowmap_amd: {what} must be float32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


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


# K⁻¹ of the few K tensors alive in a step (the model's K shared by the extrinsics fit and the fused
# losses; the softmin sweep's constant candidate sets): (id(root K), address, numel) -> (weakref, key, K⁻¹)
_kinv_cache: dict = {}
_KINV_CACHE_SLOTS = 4


def _kinv_key(k: Tensor):
    return (k._version, k.data_ptr(), k.numel())


def _kinv_store(root: Tensor, key, kinv: Tensor) -> None:
    for ident in [i for i, entry in _kinv_cache.items() if entry[0]() is None]:  # their K died
        del _kinv_cache[ident]
    while len(_kinv_cache) >= _KINV_CACHE_SLOTS:
        del _kinv_cache[next(iter(_kinv_cache))]  # oldest first
    _kinv_cache[(id(root), *key[1:])] = (weakref.ref(root), key, kinv)


def intrinsics_inverse(k: Tensor) -> Tensor:
    """K⁻¹ for a (..., 3, 3) stack (no autograd; callers chain the backward).  Results are kept for
    the last few K tensors, so the consumers of one step invert each K once and constant K sets are
    inverted once per run.  "The same K" = the same root tensor object (views of it included: they
    share its version counter), same memory, same version — a recycled allocation belongs to a
    different root object and misses."""
    k = _f32c(k, "intrinsics")
    root = k if k._base is None else k._base
    key = _kinv_key(k)
    entry = _kinv_cache.get((id(root), *key[1:]))
    if entry is not None and entry[0]() is root and entry[1] == key:
        return entry[2].view(k.shape)
    out = torch.empty_like(k)
    with _guard(k.device):
        call("fm_intrinsics_inverse", ptr(k), k.numel() // 9, ptr(out), stream_for(k))
    _kinv_store(root, key, out)
    return out


class FocalIntrinsics(torch.autograd.Function):
    """focal_lengths_to_intrinsics spread over the frames (intrinsics/common.py:6-20 as used by
    intrinsics_regressed.py:34-41): focal (*lead) -> K (*lead, *repeat_shape, 3, 3) in one launch that
    also leaves K^-1 behind for the step's consumers; the backward is one reduction."""

    @staticmethod
    def forward(ctx, focal, repeat_shape, image_shape):
        check_device(focal)
        focal = _f32c(focal, "focal lengths")
        h, w = int(image_shape[0]), int(image_shape[1])
        repeat = 1
        for d in repeat_shape:
            repeat *= int(d)
        count = focal.numel()
        k = torch.empty((*focal.shape, *repeat_shape, 3, 3), dtype=torch.float32, device=focal.device)
        kinv = torch.empty_like(k)
        with _guard(focal.device):
            call("fm_focal_intrinsics_fwd", ptr(focal), count, repeat, h, w, ptr(k), ptr(kinv), stream_for(focal))
        ctx.geometry = (count, repeat, h, w, tuple(focal.shape))
        ctx.kinv = kinv  # handed to the cache by focal_intrinsics() below, not needed for backward
        return k

    @staticmethod
    def backward(ctx, g_k):
        count, repeat, h, w, shape = ctx.geometry
        g_k = _f32c(g_k, "grad")
        g_focal = torch.empty(shape, dtype=torch.float32, device=g_k.device)
        with _guard(g_k.device):
            call("fm_focal_intrinsics_bwd", ptr(g_k), count, repeat, h, w, ptr(g_focal), stream_for(g_k))
        return g_focal, None, None


def focal_intrinsics(focal: Tensor, repeat_shape, image_shape) -> Tensor:
    """K (*focal.shape, *repeat_shape, 3, 3) from normalised focal lengths; K^-1 is computed in the
    same launch and parked where intrinsics_inverse() finds it."""
    k = FocalIntrinsics.apply(focal, tuple(repeat_shape), tuple(image_shape))
    _park_inverse(k)
    return k


# --------------------------------------------------------------------------------------
# Pose plumbing
# --------------------------------------------------------------------------------------


class PoseChain(torch.autograd.Function):
    """get_extrinsics (flowmap/model/projection.py:187-210)."""

    @staticmethod
    def forward(ctx, rel: Tensor) -> Tensor:
        check_device(rel)
        rel = _f32c(rel, "relative transformations")
        *batch, steps, _, _ = rel.shape
        nb = 1
        for d in batch:
            nb *= d
        ext = torch.empty((*batch, steps + 1, 4, 4), dtype=torch.float32, device=rel.device)
        with _guard(rel.device):
            call("fm_pose_chain_fwd", ptr(rel), nb, steps, ptr(ext), stream_for(rel))
        ctx.save_for_backward(rel, ext)
        ctx.nb, ctx.steps = nb, steps
        return ext

    @staticmethod
    def backward(ctx, g_ext: Tensor):
        rel, ext = ctx.saved_tensors
        g_ext = _f32c(g_ext, "grad")
        g_rel = torch.empty_like(rel)
        with _guard(rel.device):
            call("fm_pose_chain_bwd", ptr(rel), ptr(ext), ptr(g_ext), ctx.nb, ctx.steps, ptr(g_rel), stream_for(rel))
        return g_rel


class RelativePoses(torch.autograd.Function):
    """later(E).inverse() @ earlier(E) and earlier(E).inverse() @ later(E)
    (flowmap/model/projection.py:154,176).  extrinsics (B,F,4,4) -> two (B,F-1,4,4)."""

    @staticmethod
    def forward(ctx, ext: Tensor):
        check_device(ext)
        ext = _f32c(ext, "extrinsics")
        b, f = ext.shape[:2]
        fwd = torch.empty((b, f - 1, 4, 4), dtype=torch.float32, device=ext.device)
        bwd = torch.empty_like(fwd)
        with _guard(ext.device):
            call("fm_relative_pose_fwd", ptr(ext), b, f, ptr(fwd), ptr(bwd), stream_for(ext))
        ctx.save_for_backward(ext)
        return fwd, bwd

    @staticmethod
    def backward(ctx, g_fwd: Optional[Tensor], g_bwd: Optional[Tensor]):
        (ext,) = ctx.saved_tensors
        b, f = ext.shape[:2]
        g_fwd = None if g_fwd is None else _f32c(g_fwd, "grad")
        g_bwd = None if g_bwd is None else _f32c(g_bwd, "grad")
        g_ext = torch.empty_like(ext)
        with _guard(ext.device):
            call("fm_relative_pose_bwd", ptr(ext), ptr(g_fwd), ptr(g_bwd), b, f, ptr(g_ext), stream_for(ext))
        return g_ext


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


def _find_fit_node(t: Tensor, depth_key, max_depth: int = 4):
    """Walk up the autograd graph from a pose tensor looking for the ProcrustesFit node
    that produced it from the SAME depth tensor (see "carried depth gradient" below)."""
    start = t.grad_fn
    if start is None:
        return None
    frontier = [start]
    for _ in range(max_depth):
        nxt = []
        for node in frontier:
            if getattr(node, "_fm_fit_depth_key", None) == depth_key:
                return node
            nxt.extend(fn for fn, _ in getattr(node, "next_functions", ()) if fn is not None)
        frontier = nxt
        if not frontier:
            break
    return None


# Dense gradient buffers this backward pass has already handed to autograd, by source tensor:
# (data_ptr, version, shape) -> weakref(buffer).  A later node whose own contribution to the same
# tensor is tiny adds it into that buffer in place instead of emitting a second full-size tensor
# for autograd to sum (see LeadingFrames).  Single use; a dead or missing entry means "emit".
_emitted: dict = {}
# which path LeadingFrames.backward / the sparse Procrustes backward took (tests)
counters = {"leading_frames_in_place": 0, "leading_frames_dense": 0, "procrustes_planned": 0}


def _tensor_key(t: Tensor):
    return (t.data_ptr(), t._version, tuple(t.shape), str(t.device))


def _note_emitted(source_key, buffer: Optional[Tensor]) -> None:
    if buffer is not None and source_key is not None:
        _emitted.pop(source_key, None)  # re-insert at the end: the dict is kept in age order
        _emitted[source_key] = weakref.ref(buffer)
        if len(_emitted) > 16:  # entries nobody collected (no LeadingFrames consumer): drop dead ones, then the oldest
            for key in [k for k, ref in _emitted.items() if ref() is None]:
                del _emitted[key]
            while len(_emitted) > 16:
                del _emitted[next(iter(_emitted))]


class LeadingFrames(torch.autograd.Function):
    """``x[:, :count].contiguous()`` for (b, F, H, W) image stacks.  The softmin sweep reads two of
    the 150 depth frames; autograd's slice backward would zero-fill a full-size tensor and add it
    densely to the main path's gradient (1.7 GB of traffic at C1).  Autograd runs this node's
    backward AFTER the nodes that consume the intrinsics it helped to produce, so the dense
    gradient of ``x`` from the flow loss / extrinsics fit is already sitting in autograd's input
    buffer: the ``count`` frames are added into it in place and nothing is returned.  Falls back
    to the zero-padded tensor whenever that buffer is not known."""

    @staticmethod
    def forward(ctx, x: Tensor, count: int):
        if x.dim() != 4 or not 1 <= count <= x.shape[1]:
            raise RuntimeError("flowmap_amd: LeadingFrames expects (batch, frame, height, width) and 1 <= count <= frame")
        ctx.shape, ctx.count, ctx.key = tuple(x.shape), count, _tensor_key(x)
        return x[:, :count].contiguous()

    @staticmethod
    def backward(ctx, g):
        ref = _emitted.pop(ctx.key, None)
        buf = ref() if ref is not None else None
        if buf is not None and tuple(buf.shape) == ctx.shape and buf.dtype == g.dtype and buf.device == g.device:
            buf[:, : ctx.count].add_(g)
            counters["leading_frames_in_place"] += 1
            return None, None
        counters["leading_frames_dense"] += 1
        if ctx.count == ctx.shape[1]:
            return g, None
        full = g.new_zeros(ctx.shape)
        full[:, : ctx.count] = g
        return full, None


# Early zero fill of the sparse fit's dense dL/dweights (549 MB at C1) on a side stream: started when
# the backward pass begins (by the fused flow loss, the first node to run), behind everything the
# forward enqueued, with a bounded number of workgroups — it then overlaps the latency-bound kernels
# of the backward pass instead of standing in line between them.  Off under hipGraph capture.
prefill_weight_grads = os.environ.get("FLOWMAP_PREFILL", "1") != "0"
PREFILL_BLOCKS = int(os.environ.get("FLOWMAP_PREFILL_BLOCKS", "512"))
_side_streams: dict = {}


def _start_weight_grad_prefill(node) -> None:
    weights = node._fm_weights_like
    if (not prefill_weight_grads or graph_capturable or weights is None or not weights.is_cuda or not node.needs_input_grad[3]
            or getattr(node, "_fm_prefilled", None) is not None):
        return
    dev = weights.device
    main = torch.cuda.current_stream(dev)
    side = _side_streams.get(dev.index)
    if side is None:
        side = _side_streams[dev.index] = torch.cuda.Stream(dev)
    side.wait_stream(main)  # not before the forward's kernels (the flow kernel owns the HBM while it runs)
    with torch.cuda.stream(side):
        g_w = torch.empty_like(weights)
        with _guard(dev):
            call("fm_fill_zero", ptr(g_w), g_w.numel(), PREFILL_BLOCKS, side.cuda_stream)
    node._fm_prefilled = (g_w, side)


# Plans of the sparse Procrustes backward (fm_procrustes_scatter_plan): with constant flows and a
# constant, duplicate-free index set, the pixels the gradient touches never change.
_scatter_plans: dict = {}
_SCATTER_PLAN_SLOTS = 4


def _procrustes_scatter_plan(indices: Tensor, bwd_flow: Tensor, b: int, f: int, h: int, w: int):
    """(pixels, first, vector index per entry, weights) for fm_depth_gather, or None.  A plan costs a
    sort, so it is built when the same (indices, flows) tensors come back a second time — per-step
    random indices never qualify — and only for index sets without duplicates."""
    key = (indices.data_ptr(), indices._version, indices.numel(), bwd_flow.data_ptr(), bwd_flow._version, b, f, h, w)
    entry = _scatter_plans.get(key)
    if entry is not None and (entry[0]() is not indices or entry[1]() is not bwd_flow):
        entry = None  # recycled addresses
    if entry is None:
        for stale in [k_ for k_, e_ in _scatter_plans.items() if e_[0]() is None or e_[1]() is None]:
            del _scatter_plans[stale]
        while len(_scatter_plans) >= _SCATTER_PLAN_SLOTS:
            del _scatter_plans[next(iter(_scatter_plans))]
        _scatter_plans[key] = [weakref.ref(indices), weakref.ref(bwd_flow), None, False]  # [.., plan, built]
        return None
    if not entry[3]:
        entry[3] = True
        points = indices.numel()
        if torch.unique(indices).numel() == points:
            dev = bwd_flow.device
            keys = torch.empty((b * (f - 1) * points * 5,), dtype=torch.int64, device=dev)
            weights = torch.empty((keys.numel(),), dtype=torch.float32, device=dev)
            with _guard(dev):
                call("fm_procrustes_scatter_plan", ptr(bwd_flow), ptr(indices), points, b, f, h, w, ptr(keys), ptr(weights), stream_for(bwd_flow))
            used = torch.nonzero(keys >= 0).reshape(-1)
            sorted_keys, order = torch.sort(keys[used], stable=True)
            entries = used[order]
            pixels, counts = torch.unique_consecutive(sorted_keys, return_counts=True)
            first = torch.zeros((pixels.numel() + 1,), dtype=torch.int32, device=dev)
            first[1:] = torch.cumsum(counts, 0).to(torch.int32)
            vectors = (torch.div(entries, 5, rounding_mode="floor") * 2 + (entries % 5 == 4)).to(torch.int32)
            entry[2] = (pixels.contiguous(), first, vectors.contiguous(), weights[entries].contiguous())
    return entry[2]


def _dense_procrustes_plan(bwd_flow: Tensor, b: int, f: int, h: int, w: int):
    """(first int64 (pairs*tiles + 1), list uint32-as-int32 (entries)) of fm_procrustes_dense_plan: for every tile of
    every pair's earlier frame, the later pixels whose bilinear taps land in it.  The flows are constants
    of the optimisation, so this is built once per flow tensor and kept ON that tensor (it lives and dies
    with it; an in-place edit bumps the version and the plan is rebuilt)."""
    key = (bwd_flow._version, b, f, h, w)
    hit = getattr(bwd_flow, "_fm_dense_plan", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    import ctypes

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
    bwd_flow._fm_dense_plan = (key, first, entries)
    return first, entries


class ProcrustesFit(torch.autograd.Function):
    """align_surfaces up to (not including) the pose chain (projection.py:213-249) with
    align_rigid (procrustes.py:7-51) inside.  Source of xyz is either

      depth (B,F,H,W) + intrinsics (B,F,3,3)   [surfaces never materialised], or
      surfaces (B,F,H,W,3).

    ``weight_sens != 0``: ``weights`` holds LOGITS and w = sigmoid(weight_sens·logit) is
    evaluated at the gathered points only (BackboneExplicitDepth fused into the gather);
    the returned gradient is then w.r.t. the logits.

    Returns the "inverse relative transformations" (B,F-1,4,4): later -> earlier camera, and
    their rigid inverses (earlier -> later camera) for consumers that want both directions
    without going through the pose chain.
    """

    @staticmethod
    def forward(ctx, depth, k, surfaces, weights, bwd_flow, indices, weight_sens=0.0, batch_repeat=1):
        from_depth = surfaces is None
        check_device(depth if from_depth else surfaces, weights, bwd_flow, indices)
        weights = _f32c(weights, "weights")
        bwd_flow = _f32c(bwd_flow, "backward flow")
        if bwd_flow.requires_grad:
            raise RuntimeError("flowmap_amd: gradients w.r.t. optical flow are not supported (flows are constants)")
        rep = int(batch_repeat)
        if from_depth:
            depth = _f32c(depth, "depth")
            k = _f32c(k, "intrinsics")
            bd, f, h, w = depth.shape
            b = bd * rep  # pose / intrinsics batch: every image-batch entry serves `rep` candidates
            if tuple(k.shape) != (b, f, 3, 3):
                raise RuntimeError("flowmap_amd: intrinsics shape does not match depth (x batch_repeat)")
            kinv = intrinsics_inverse(k)
            dev = depth.device
        else:
            if rep != 1:
                raise RuntimeError("flowmap_amd: batch_repeat needs depth-sourced surfaces")
            surfaces = _f32c(surfaces, "surfaces")
            bd, f, h, w, _ = surfaces.shape
            b = bd
            kinv = None
            dev = surfaces.device
        if tuple(weights.shape) != (bd, f - 1, h, w) or tuple(bwd_flow.shape) != (bd, f - 1, h, w, 2):
            raise RuntimeError("flowmap_amd: weights/backward-flow shapes do not match the surfaces")
        if indices is not None:
            if indices.dtype != torch.int64:
                raise RuntimeError("flowmap_amd: indices must be int64")
            indices = indices.contiguous()
            points = indices.numel()
        else:
            points = h * w
        pairs = b * (f - 1)
        stats = torch.empty((pairs, STAT_STRIDE), dtype=torch.float64, device=dev)
        t_bwd = torch.empty((b, f - 1, 4, 4), dtype=torch.float32, device=dev)
        t_fwd = torch.empty_like(t_bwd)
        aux = torch.empty((pairs, AUX_STRIDE), dtype=torch.float64, device=dev)
        with _guard(dev):
            st = stream_for(weights)
            call("fm_procrustes_fit", ptr(depth) if from_depth else None, ptr(kinv), ptr(surfaces), ptr(bwd_flow),
                 ptr(weights), float(weight_sens), ptr(indices), points, b, rep, f, h, w, ptr(stats), ptr(t_bwd), ptr(t_fwd), ptr(aux), st)
        ctx.save_for_backward(depth if from_depth else surfaces, kinv, weights, bwd_flow, indices, t_bwd, aux)
        ctx.from_depth, ctx.dims, ctx.points, ctx.weight_sens, ctx.rep = from_depth, (b, f, h, w), points, float(weight_sens), rep
        # Carried depth gradient: when the fused flow loss consumes poses fitted from the very
        # same depth tensor, it parks its dense dL/ddepth here instead of returning it, and
        # this node (which autograd always runs later) scatters its sparse part into that
        # buffer and returns the sum once — no second dense tensor, no dense add.
        ctx._fm_src_key = _tensor_key(depth if from_depth else surfaces)
        ctx._fm_weights_key = _tensor_key(weights)
        ctx._fm_fit_depth_key = (depth.data_ptr(), depth._version, tuple(depth.shape)) if from_depth else None
        ctx._fm_carried = None
        ctx._fm_pending = []  # deferred scatters of other losses into the final dL/ddepth buffer
        # sparse fits accumulate into a zeroed dL/dweights: the fused flow loss may start that fill early
        ctx._fm_weights_like = weights if (indices is not None and rep == 1) else None
        ctx._fm_prefilled = None
        return t_bwd, t_fwd

    @staticmethod
    def backward(ctx, g_t, g_t_fwd):
        src, kinv, weights, bwd_flow, indices, t_bwd, aux = ctx.saved_tensors
        b, f, h, w = ctx.dims
        pairs = b * (f - 1)
        dev = weights.device
        g_t = None if g_t is None else _f32c(g_t, "grad")
        g_t_fwd = None if g_t_fwd is None else _f32c(g_t_fwd, "grad")
        need_src = ctx.needs_input_grad[0] if ctx.from_depth else ctx.needs_input_grad[2]
        need_k = ctx.from_depth and ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[3]
        pair_grad = torch.empty((pairs, PAIR_GRAD_STRIDE), dtype=torch.float64, device=dev)
        g_src = g_k = g_w = fill_stream = None
        carried = ctx._fm_carried
        ctx._fm_carried = None
        pending, ctx._fm_pending = ctx._fm_pending, []
        if need_src:
            g_src = carried if carried is not None else torch.zeros_like(src)
            for scatter in pending:
                scatter(g_src)
        if need_w:
            # (zero-filling this 549 MB buffer at FORWARD time on a side stream was measured and
            # rejected: the fill contends with the fused flow kernel — step 1.126 -> 1.187 ms.  What
            # works is _start_weight_grad_prefill: started when the backward pass begins, bounded
            # footprint, and the P values per pair placed by fm_sparse_store after the join.)
            # the tiled dense kernels (depth-sourced, every pixel a correspondence) STORE every
            # element of dL/dweights; all other paths accumulate atomically into zeros
            dense_tiled = ctx.from_depth and indices is None and ctx.rep == 1 and h <= 65535 and w <= 65535
            prefilled, ctx._fm_prefilled = ctx._fm_prefilled, None
            if prefilled is not None:
                g_w, fill_stream = prefilled  # being zeroed on the side stream; joined below
                g_w.record_stream(torch.cuda.current_stream(dev))
            else:
                g_w = torch.empty_like(weights) if dense_tiled else torch.zeros_like(weights)
        kinv_acc = torch.empty((b * f, 9), dtype=torch.float64, device=dev) if need_k else None  # zeroed by fm_pose_solve_bwd
        # sparse depth-sourced fit with constant indices / flows: the depth gradient is gathered along a
        # plan instead of scattered with atomics (which run at the memory side: 100 us for 0.9 M adds)
        plan = point_grads = point_gw = None
        if ctx.from_depth and indices is not None and ctx.rep == 1:
            plan = _procrustes_scatter_plan(indices, bwd_flow, b, f, h, w)
            if plan is not None:
                point_grads = torch.empty((pairs * ctx.points, 2, 3), dtype=torch.float32, device=dev)
                counters["procrustes_planned"] += 1
                if fill_stream is not None and g_w is not None:  # keep the per-point pass off the buffer still being zeroed
                    point_gw = torch.empty((pairs * ctx.points,), dtype=torch.float32, device=dev)
        if fill_stream is not None and point_gw is None:
            torch.cuda.current_stream(dev).wait_stream(fill_stream)  # the scatter writes into g_w directly
            fill_stream = None
        dense = ctx.from_depth and indices is None and ctx.rep == 1 and h <= 65535 and w <= 65535
        with _guard(dev):
            st = stream_for(weights)
            call("fm_pose_solve_bwd", ptr(g_t), ptr(g_t_fwd), ptr(t_bwd), ptr(aux), pairs, ptr(pair_grad), ptr(kinv_acc),
                 0 if kinv_acc is None else kinv_acc.numel(), st)
            if dense:  # every pixel a correspondence: tiled, planned, no atomics (fm_procrustes_scatter_dense)
                first = entries = None
                if g_src is not None:
                    first, entries = _dense_procrustes_plan(bwd_flow, b, f, h, w)
                consts = torch.empty((pairs, DENSE_CONST_STRIDE), dtype=torch.float64, device=dev)
                call("fm_procrustes_scatter_dense", ptr(src), ptr(kinv), ptr(bwd_flow), ptr(weights), ctx.weight_sens, b, f, h, w, ptr(aux),
                     ptr(pair_grad), ptr(g_src), ptr(g_w), ptr(kinv_acc), ptr(first), ptr(entries), ptr(consts), st)
            else:
                call("fm_procrustes_scatter", ptr(src) if ctx.from_depth else None, ptr(kinv), None if ctx.from_depth else ptr(src),
                     ptr(bwd_flow), ptr(weights), ctx.weight_sens, ptr(indices), ctx.points, b, ctx.rep, f, h, w, ptr(aux), ptr(pair_grad),
                     ptr(g_src) if ctx.from_depth else None, None if ctx.from_depth else ptr(g_src), ptr(g_w), ptr(kinv_acc),
                     ptr(point_grads), ptr(point_gw), st)
            if plan is not None and g_src is not None:
                pixels, first, vectors, weights_e = plan
                call("fm_depth_gather", ptr(point_grads), ptr(pixels), ptr(first), ptr(vectors), ptr(weights_e), pixels.numel(), ptr(kinv),
                     None, None, h, w, 0, ptr(g_src), st)
            if point_gw is not None:
                torch.cuda.current_stream(dev).wait_stream(fill_stream)  # the zero fill is done: place the P values per pair
                call("fm_sparse_store", ptr(point_gw), ptr(indices), ctx.points, pairs, h * w, ptr(g_w), st)
            if need_k:
                g_k = torch.empty_like(kinv)
                call("fm_intrinsics_inverse_bwd", ptr(kinv_acc), ptr(kinv), b * f, ptr(g_k), 0, st)
        _note_emitted(ctx._fm_src_key, g_src)
        _note_emitted(ctx._fm_weights_key, g_w)
        if ctx.from_depth:
            return g_src, g_k, None, g_w, None, None, None, None
        return None, None, g_src, g_w, None, None, None, None


# --------------------------------------------------------------------------------------
# Fused flow loss
# --------------------------------------------------------------------------------------

_norm_cache: dict = {}

# bench.py sets this to a list to collect (start, end) torch.cuda.Event pairs around every
# launch of the fused flow kernel (same stream as the launch).
flow_kernel_events = None


def flow_valid_norm(mask_fwd: Tensor, mask_bwd: Tensor, weight: float, reducer=None) -> Tensor:
    """Device tensor [weight/(V or 1), (V or 1)] with V = Σmask_fwd + Σmask_bwd
    (loss_flow.py:56,66,70).  Masks are constants of the optimisation, so the result is
    cached per (storage, version, weight) and costs nothing after the first step.
    ``reducer`` (frame sharding) maps the local fp64 sum to the global one."""
    key = (mask_fwd.data_ptr(), mask_bwd.data_ptr(), mask_fwd._version, mask_bwd._version, tuple(mask_fwd.shape),
           float(weight), str(mask_fwd.device), id(reducer))
    hit = _norm_cache.get(key)
    if hit is not None:
        norm, ref_f, ref_b = hit
        if ref_f() is mask_fwd and ref_b() is mask_bwd:  # same live tensors, not a recycled address
            return norm
    vsum = torch.empty((1,), dtype=torch.float64, device=mask_fwd.device)
    norm = torch.empty((2,), dtype=torch.float32, device=mask_fwd.device)
    with _guard(mask_fwd.device):
        call("fm_flow_valid_norm", ptr(mask_fwd), ptr(mask_bwd), mask_fwd.numel(), float(weight), ptr(vsum), ptr(norm),
             stream_for(mask_fwd))
    if reducer is not None:
        vsum = reducer(vsum)
        veff = torch.where(vsum == 0, torch.ones_like(vsum), vsum)
        norm = torch.cat([float(weight) / veff, veff]).to(torch.float32)
    if len(_norm_cache) > 8:
        _norm_cache.clear()
    _norm_cache[key] = (norm, weakref.ref(mask_fwd), weakref.ref(mask_bwd))
    return norm


_pack_cache: dict = {}

# The packed copy costs as much HBM as the flows and masks themselves (3.3 GB at C1); set to
# False to stream the caller's tensors directly (tests exercise both kernels).
use_packed_inputs = True


def packed_flow_inputs(flow_fwd: Tensor, flow_bwd: Tensor, mask_fwd: Tensor, mask_bwd: Tensor) -> Optional[Tensor]:
    """Flows + masks in the layout of fm_flow_pack_inputs, or None when it does not apply
    (width not a multiple of 4, unexpected shapes / dtypes).  Like the valid-sum, these are
    constants of an optimisation (flow_predictor.py:82-102 runs once per video), so the
    re-layout runs once per Flows object: cached per (storage, version) of all four tensors
    and validated against the live tensors."""
    if not use_packed_inputs:
        return None
    srcs = (flow_fwd, flow_bwd, mask_fwd, mask_bwd)
    if mask_fwd.dim() != 4 or mask_fwd.shape[-1] % 4 != 0 or tuple(mask_bwd.shape) != tuple(mask_fwd.shape):
        return None
    if tuple(flow_fwd.shape) != (*mask_fwd.shape, 2) or tuple(flow_bwd.shape) != (*mask_fwd.shape, 2):
        return None
    if any(t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 != 0 for t in srcs):
        return None
    key = tuple((t.data_ptr(), t._version) for t in srcs) + (tuple(mask_fwd.shape), str(mask_fwd.device))
    hit = _pack_cache.get(key)
    if hit is not None and all(ref() is t for ref, t in zip(hit[1], srcs)):
        return hit[0]
    b, pairs, h, w = mask_fwd.shape
    chunks = (h * w // 4 + 63) // 64
    packed = torch.empty((b * (pairs + 1), chunks, 6, 64, 4), dtype=torch.float32, device=mask_fwd.device)
    with _guard(mask_fwd.device):
        call("fm_flow_pack_inputs", ptr(flow_fwd), ptr(flow_bwd), ptr(mask_fwd), ptr(mask_bwd), b, pairs + 1, h, w, ptr(packed),
             stream_for(mask_fwd))
    if len(_pack_cache) > 4:
        _pack_cache.clear()
    _pack_cache[key] = (packed, [weakref.ref(t) for t in srcs])
    return packed


class FlowLossFused(torch.autograd.Function):
    """weight · LossFlow.compute_unweighted_loss (flowmap/loss/loss_flow.py:31-70,
    flowmap/loss/loss.py:47) evaluated from depth + intrinsics + relative poses, with the
    analytic gradient of every input produced in the same HBM pass."""

    @staticmethod
    def forward(ctx, depth, k, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm, kind, delta, carry, items, packed=None):
        dev = check_device(depth, k, t_fwd, t_bwd, flow_fwd, flow_bwd, mask_fwd, mask_bwd, norm)
        depth = _f32c(depth, "depth")
        k = _f32c(k, "intrinsics")
        t_fwd = _f32c(t_fwd, "forward poses")
        t_bwd = _f32c(t_bwd, "backward poses")
        flow_fwd, flow_bwd = _f32c(flow_fwd, "forward flow"), _f32c(flow_bwd, "backward flow")
        mask_fwd, mask_bwd = _f32c(mask_fwd, "forward mask"), _f32c(mask_bwd, "backward mask")
        b, f, h, w = depth.shape
        if tuple(flow_fwd.shape) != (b, f - 1, h, w, 2) or tuple(flow_bwd.shape) != (b, f - 1, h, w, 2):
            raise RuntimeError("flowmap_amd: flow shape does not match depth")
        if tuple(mask_fwd.shape) != (b, f - 1, h, w) or tuple(mask_bwd.shape) != (b, f - 1, h, w):
            raise RuntimeError("flowmap_amd: mask shape does not match depth")
        if tuple(k.shape) != (b, f, 3, 3) or tuple(t_fwd.shape) != (b, f - 1, 4, 4) or tuple(t_bwd.shape) != (b, f - 1, 4, 4):
            raise RuntimeError("flowmap_amd: intrinsics / pose shapes do not match depth")
        need = any(ctx.needs_input_grad[:4])
        kinv = intrinsics_inverse(k)
        if packed is not None and (packed.dtype != torch.float32 or not packed.is_contiguous() or w % 4 != 0
                                   or tuple(packed.shape) != (b * f, (h * w // 4 + 63) // 64, 6, 64, 4)):
            raise RuntimeError("flowmap_amd: packed flow inputs do not match the depth shape")
        acc = torch.empty((b * f * 2 * FLOW_ACC_STRIDE,), dtype=torch.float64, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        g_depth = torch.empty_like(depth) if (need and ctx.needs_input_grad[0]) else None
        # the three small gradients share one allocation so one launch rescales them in backward
        small = torch.empty((2 * t_fwd.numel() + k.numel(),), dtype=torch.float32, device=dev)
        g_tf = small[: t_fwd.numel()].view_as(t_fwd)
        g_tb = small[t_fwd.numel() : 2 * t_fwd.numel()].view_as(t_bwd)
        g_k = small[2 * t_fwd.numel() :].view_as(k)
        scale = (h * w) ** 0.5
        events = None
        if flow_kernel_events is not None and depth.is_cuda:
            events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        with _guard(dev):
            st = stream_for(depth)
            if events:
                events[0].record()
            call("fm_flow_loss_fused", ptr(depth), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), ptr(flow_fwd), ptr(flow_bwd),
                 ptr(mask_fwd), ptr(mask_bwd), ptr(packed), ptr(norm) if need else None, b, f, h, w, kind, float(delta), w / scale, h / scale,
                 ptr(g_depth), ptr(acc), int(items), st)
            if events:
                events[1].record()
                flow_kernel_events.append(events)
            call("fm_flow_loss_finalize", ptr(acc), ptr(k), ptr(kinv), ptr(t_fwd), ptr(t_bwd), ptr(norm), b, f, w / scale, h / scale,
                 ptr(loss), ptr(g_tf), ptr(g_tb), ptr(g_k), st)
        ctx.grads = (g_depth, g_k, g_tf, g_tb, small) if need else None
        ctx.fit_node = None
        if carry and g_depth is not None:
            key = (depth.data_ptr(), depth._version, tuple(depth.shape))
            node = _find_fit_node(t_bwd, key)
            if node is not None and node.needs_input_grad[0] and _find_fit_node(t_fwd, key) is node:
                ctx.fit_node = node
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            raise RuntimeError("flowmap_amd: FlowLossFused gradients are single-use; run the forward again")
        g_depth, g_k, g_tf, g_tb, small = ctx.grads
        ctx.grads = None
        if ctx.fit_node is not None:
            _start_weight_grad_prefill(ctx.fit_node)
        g = g.reshape(1).to(torch.float32).contiguous()
        with _guard(g.device):
            st = stream_for(g)
            call("fm_scale_if_needed", ptr(g_depth), 0 if g_depth is None else g_depth.numel(), ptr(small), small.numel(), ptr(g), st)
        node = ctx.fit_node
        ctx.fit_node = None
        if node is not None and g_depth is not None and node._fm_carried is None:
            node._fm_carried = g_depth  # returned (summed) by ProcrustesFit.backward
            g_depth = None
        need = ctx.needs_input_grad
        return (g_depth if (need[0] and g_depth is not None) else None, g_k if need[1] else None, g_tf if need[2] else None,
                g_tb if need[3] else None, None, None, None, None, None, None, None, None, None, None)


class SoftminScore(torch.autograd.Function):
    """IntrinsicsSoftmin's per-candidate weighted L1 flow error (intrinsics_softmin.py:105-121) from
    the images themselves: depth (B,2,H,W) frames 0/1, weights (B,1,H,W) of pair 0 (logits when
    ``weight_sens`` != 0), bwd_flow (B,1,H,W,2), indices (P) distinct pixels, candidate intrinsics
    (N,3,3) [constants], rel (B·N,4,4) the fitted later->earlier poses  ->  error (B,N)."""

    @staticmethod
    def forward(ctx, depth, weights, bwd_flow, indices, k, rel, weight_sens):
        saved, err = _softmin_score_forward(depth, weights, bwd_flow, indices, k, rel, weight_sens)
        ctx.save_for_backward(*saved)
        ctx.sens = float(weight_sens)
        return err.to(torch.float32).reshape(saved[0].shape[0], saved[4].shape[0])

    @staticmethod
    def backward(ctx, g_err):
        g_depth, g_weights, g_rel = _softmin_score_backward(ctx.saved_tensors, ctx.sens, _f32c(g_err, "grad"), ctx.needs_input_grad)
        return g_depth, g_weights, None, None, None, g_rel, None


def _softmin_score_forward(depth, weights, bwd_flow, indices, k, rel, weight_sens):
    """Checks + launch shared by SoftminScore and SoftminIntrinsics -> (tensors to save, err fp64 (b*n))."""
    dev = check_device(depth, weights, bwd_flow, indices, k, rel)
    depth, weights, bwd_flow = _f32c(depth, "depth"), _f32c(weights, "weights"), _f32c(bwd_flow, "backward flow")
    k, rel = _f32c(k, "intrinsics"), _f32c(rel, "poses")
    b, two, h, w = depth.shape
    n = k.shape[0]
    if two != 2 or tuple(weights.shape) != (b, 1, h, w) or tuple(bwd_flow.shape) != (b, 1, h, w, 2):
        raise RuntimeError("flowmap_amd: SoftminScore expects depth (b,2,h,w), weights (b,1,h,w), backward flow (b,1,h,w,2)")
    if tuple(k.shape) != (n, 3, 3) or tuple(rel.shape) != (b * n, 4, 4) or indices.dtype != torch.int64:
        raise RuntimeError("flowmap_amd: SoftminScore expects intrinsics (n,3,3), poses (b*n,4,4), int64 indices")
    if k.requires_grad or bwd_flow.requires_grad:
        raise RuntimeError("flowmap_amd: the softmin candidates and the optical flow are constants")
    indices = indices.contiguous()
    kinv = intrinsics_inverse(k)
    err = torch.empty((b * n,), dtype=torch.float64, device=dev)
    with _guard(dev):
        call("fm_softmin_score_fwd", ptr(depth), ptr(weights), float(weight_sens), ptr(bwd_flow), ptr(indices), indices.numel(),
             ptr(k), ptr(kinv), ptr(rel), b, n, h, w, ptr(err), stream_for(depth))
    return (depth, weights, bwd_flow, indices, k, kinv, rel), err


def _softmin_score_backward(saved, sens, g_err, needs):
    """g_err (b*n) fp32 -> (g_depth, g_weights, g_rel); ``needs`` indexes like SoftminScore's inputs."""
    depth, weights, bwd_flow, indices, k, kinv, rel = saved
    b, _, h, w = depth.shape
    n = k.shape[0]
    g_err = g_err.reshape(b * n)
    g_depth = torch.zeros_like(depth) if needs[0] else None
    g_weights = torch.zeros_like(weights) if needs[1] else None
    acc = torch.empty((b * n, 12), dtype=torch.float64, device=depth.device)
    g_rel = torch.empty_like(rel)
    with _guard(depth.device):
        call("fm_softmin_score_bwd", ptr(depth), ptr(weights), sens, ptr(bwd_flow), ptr(indices), indices.numel(), ptr(k),
             ptr(kinv), ptr(rel), b, n, h, w, ptr(g_err), ptr(g_depth), ptr(g_weights), ptr(acc), ptr(g_rel), stream_for(depth))
    return g_depth, g_weights, g_rel if needs[5] else None


class SoftminIntrinsics(torch.autograd.Function):
    """SoftminScore followed by the tail of IntrinsicsSoftmin.forward (intrinsics_softmin.py:105-141)
    as three launches forward, three backward: the candidates' flow errors stay in their fp64
    accumulator, one wave per batch entry turns them into the softmin weights, the blended K for every
    frame and its inverse.  -> (K (B,frames,3,3), soft (B,N) [not differentiable: the window's input])."""

    @staticmethod
    def forward(ctx, depth, weights, bwd_flow, indices, k, rel, weight_sens, frames):
        saved, err = _softmin_score_forward(depth, weights, bwd_flow, indices, k, rel, weight_sens)
        b, n, frames = saved[0].shape[0], saved[4].shape[0], int(frames)
        dev = err.device
        soft = torch.empty((b, n), dtype=torch.float32, device=dev)
        out = torch.empty((b, frames, 3, 3), dtype=torch.float32, device=dev)
        kinv_out = torch.empty_like(out)
        with _guard(dev):
            call("fm_softmin_blend_fwd", ptr(err), ptr(saved[4]), b, n, frames, ptr(soft), ptr(out), ptr(kinv_out), stream_for(out))
        ctx.save_for_backward(*saved, soft)
        ctx.sens, ctx.frames = float(weight_sens), frames
        ctx.kinv = kinv_out  # handed to the K^-1 cache by softmin_intrinsics()
        ctx.mark_non_differentiable(soft)
        ctx.set_materialize_grads(False)  # no zeros tensor for soft's (absent) gradient
        return out, soft

    @staticmethod
    def backward(ctx, g_k, _g_soft):
        if g_k is None:
            return (None,) * 8
        *saved, soft = ctx.saved_tensors
        b, n = soft.shape
        g_k = _f32c(g_k, "grad")
        g_err = torch.empty((b * n,), dtype=torch.float32, device=g_k.device)
        with _guard(g_k.device):
            call("fm_softmin_blend_bwd", ptr(g_k), ptr(soft), ptr(saved[4]), b, n, ctx.frames, ptr(g_err), stream_for(g_k))
        g_depth, g_weights, g_rel = _softmin_score_backward(saved, ctx.sens, g_err, ctx.needs_input_grad)
        return g_depth, g_weights, None, None, None, g_rel, None, None


def _park_inverse(k: Tensor) -> None:
    """Move the K^-1 a fused producer computed alongside ``k`` into intrinsics_inverse()'s cache."""
    node = k.grad_fn
    kinv = getattr(node, "kinv", None) if node is not None else None
    if kinv is not None:
        node.kinv = None
        _kinv_store(k, _kinv_key(k), kinv)


def softmin_intrinsics(depth, weights, bwd_flow, indices, candidate_k, rel, weight_sens, frames):
    """-> (K (b,frames,3,3) blended over the candidates, softmin weights (b,n))."""
    k, soft = SoftminIntrinsics.apply(depth, weights, bwd_flow, indices, candidate_k, rel, weight_sens, frames)
    _park_inverse(k)
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
    out = torch.empty((count,), dtype=torch.int64, device=device)
    check_device(out)
    if seed is None and graph_capturable:
        state = _rng_states.get(str(out.device))
        if state is None:  # created OUTSIDE any capture (GraphedStep warms the step up first)
            first = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
            state = _rng_states[str(out.device)] = torch.tensor([first], dtype=torch.int64).to(out.device)
        with _guard(device):
            call("fm_random_subset_stateful", ptr(state), int(n), int(count), ptr(out), stream_for(out))
        return out
    if seed is None:
        seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    with _guard(device):
        call("fm_random_subset", int(seed), int(n), int(count), ptr(out), stream_for(out))
    return out


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


# --------------------------------------------------------------------------------------
# Function-level building blocks on explicit point sets
# --------------------------------------------------------------------------------------


class Unproject(torch.autograd.Function):
    """unproject (flowmap/model/projection.py:76-90) for G groups of N points:
    xy (N,2) shared or (G,N,2); z (G,N); k (G,3,3) -> (G,N,3)."""

    @staticmethod
    def forward(ctx, xy, z, k):
        dev = check_device(xy, z, k)
        xy, z, k = _f32c(xy, "coordinates"), _f32c(z, "z"), _f32c(k, "intrinsics")
        if xy.requires_grad:
            raise RuntimeError("flowmap_amd: gradients w.r.t. image coordinates are not supported")
        g, n = z.shape
        shared = xy.dim() == 2
        kinv = intrinsics_inverse(k)
        out = torch.empty((g, n, 3), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_unproject_fwd", ptr(xy), 0 if shared else n * 2, ptr(z), ptr(kinv), g, n, ptr(out), stream_for(z))
        ctx.save_for_backward(xy, z, kinv)
        ctx.shared = shared
        return out

    @staticmethod
    def backward(ctx, g_out):
        xy, z, kinv = ctx.saved_tensors
        g, n = z.shape
        g_out = _f32c(g_out, "grad")
        g_z = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        need_k = ctx.needs_input_grad[2]
        acc = torch.empty((g, 9), dtype=torch.float64, device=z.device) if need_k else None
        g_k = None
        with _guard(z.device):
            st = stream_for(z)
            call("fm_unproject_bwd", ptr(xy), 0 if ctx.shared else n * 2, ptr(z), ptr(kinv), ptr(g_out), g, n, ptr(g_z), ptr(acc), st)
            if need_k:
                g_k = torch.empty_like(kinv)
                call("fm_intrinsics_inverse_bwd", ptr(acc), ptr(kinv), g, ptr(g_k), 0, st)
        return None, g_z, g_k


class Reproject(torch.autograd.Function):
    """reproject_points (flowmap/model/projection.py:116-134): xyz (G,N,3), T (G,4,4),
    K (G,3,3) -> xy (G,N,2)."""

    @staticmethod
    def forward(ctx, xyz, t, k):
        dev = check_device(xyz, t, k)
        xyz, t, k = _f32c(xyz, "points"), _f32c(t, "transformations"), _f32c(k, "intrinsics")
        g, n, _ = xyz.shape
        out = torch.empty((g, n, 2), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_reproject_fwd", ptr(xyz), ptr(t), ptr(k), g, n, ptr(out), stream_for(xyz))
        ctx.save_for_backward(xyz, t, k)
        return out

    @staticmethod
    def backward(ctx, g_xy):
        xyz, t, k = ctx.saved_tensors
        g, n, _ = xyz.shape
        g_xy = _f32c(g_xy, "grad")
        g_xyz = torch.empty_like(xyz) if ctx.needs_input_grad[0] else None
        g_t = torch.empty_like(t)
        g_k = torch.empty_like(k)
        acc = torch.empty((g, 18), dtype=torch.float64, device=xyz.device)
        with _guard(xyz.device):
            call("fm_reproject_bwd", ptr(xyz), ptr(t), ptr(k), ptr(g_xy), g, n, ptr(g_xyz), ptr(g_t), ptr(g_k), ptr(acc),
                 stream_for(xyz))
        return g_xyz, g_t if ctx.needs_input_grad[1] else None, g_k if ctx.needs_input_grad[2] else None


class BilinearSample(torch.autograd.Function):
    """F.grid_sample(bilinear, border, align_corners=False) of a channels-last image
    (G,H,W,C) at normalised coordinates (G,P,2) in (0,1) -> (G,P,C)
    (flowmap/model/projection.py:235-241,266-272)."""

    @staticmethod
    def forward(ctx, img, xy):
        dev = check_device(img, xy)
        img, xy = _f32c(img, "image"), _f32c(xy, "coordinates")
        if xy.requires_grad:
            raise RuntimeError("flowmap_amd: gradients w.r.t. sampling coordinates are not supported")
        g, h, w, c = img.shape
        p = xy.shape[1]
        out = torch.empty((g, p, c), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_bilinear_sample_fwd", ptr(img), ptr(xy), g, h, w, c, p, ptr(out), stream_for(img))
        ctx.save_for_backward(xy)
        ctx.dims = (g, h, w, c, p)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (xy,) = ctx.saved_tensors
        g, h, w, c, p = ctx.dims
        g_out = _f32c(g_out, "grad")
        g_img = torch.zeros((g, h, w, c), dtype=torch.float32, device=xy.device)
        with _guard(xy.device):
            call("fm_bilinear_sample_bwd", ptr(g_out), ptr(xy), g, h, w, c, p, ptr(g_img), stream_for(xy))
        return g_img, None


class RobustMapping(torch.autograd.Function):
    """Mapping.forward (flowmap/loss/mapping/mapping.py:35-43) on (n,2) pairs."""

    @staticmethod
    def forward(ctx, a, b, kind, delta, ax, ay):
        dev = check_device(a, b)
        a, b = _f32c(a, "a"), _f32c(b, "b")
        n = a.shape[0]
        out = torch.empty((n,), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_mapping_fwd", ptr(a), ptr(b), n, kind, float(delta), float(ax), float(ay), ptr(out), stream_for(a))
        ctx.save_for_backward(a, b)
        ctx.cfg = (kind, float(delta), float(ax), float(ay))
        return out

    @staticmethod
    def backward(ctx, g_out):
        a, b = ctx.saved_tensors
        kind, delta, ax, ay = ctx.cfg
        g_out = _f32c(g_out, "grad")
        g_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        g_b = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        with _guard(a.device):
            call("fm_mapping_bwd", ptr(a), ptr(b), ptr(g_out), a.shape[0], kind, delta, ax, ay, ptr(g_a), ptr(g_b), stream_for(a))
        return g_a, g_b, None, None, None, None


class AlignRigid(torch.autograd.Function):
    """align_rigid (flowmap/model/procrustes.py:7-51): p, q (G,P,3), w (G,P) -> (G,4,4)."""

    @staticmethod
    def forward(ctx, p, q, w):
        dev = check_device(p, q, w)
        p, q, w = _f32c(p, "p"), _f32c(q, "q"), _f32c(w, "weights")
        g, n, _ = p.shape
        stats = torch.empty((g, STAT_STRIDE), dtype=torch.float64, device=dev)
        t = torch.empty((g, 4, 4), dtype=torch.float32, device=dev)
        aux = torch.empty((g, AUX_STRIDE), dtype=torch.float64, device=dev)
        with _guard(dev):
            st = stream_for(p)
            call("fm_align_rigid_stats", ptr(p), ptr(q), ptr(w), g, n, ptr(stats), st)
            call("fm_pose_solve", ptr(stats), g, ptr(t), None, ptr(aux), st)
        ctx.save_for_backward(p, q, w, t, aux)
        return t

    @staticmethod
    def backward(ctx, g_t):
        p, q, w, t, aux = ctx.saved_tensors
        g, n, _ = p.shape
        g_t = _f32c(g_t, "grad")
        pair_grad = torch.empty((g, PAIR_GRAD_STRIDE), dtype=torch.float64, device=p.device)
        g_p = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        g_q = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        g_w = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        with _guard(p.device):
            st = stream_for(p)
            call("fm_pose_solve_bwd", ptr(g_t), None, ptr(t), ptr(aux), g, ptr(pair_grad), None, 0, st)
            call("fm_align_rigid_bwd", ptr(p), ptr(q), ptr(w), g, n, ptr(aux), ptr(pair_grad), ptr(g_p), ptr(g_q), ptr(g_w), st)
        return g_p, g_q, g_w


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
        self._plans: dict = {}

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
            self._plans[key] = plan
        return self._plans[key]


_packed_cache: dict = {}


def pack_tracks(tracks, device, own=None) -> PackedTracks:
    key = tuple((t.xy.data_ptr(), t.xy._version, t.visibility.data_ptr(), int(t.start_frame), tuple(t.xy.shape)) for t in tracks) + (str(device), own)
    hit = _packed_cache.get(key)
    if hit is not None:
        packed, refs = hit
        if all(r() is t.xy for r, t in zip(refs, tracks)):  # live tensors, not recycled addresses
            return packed
    if len(_packed_cache) > 4:
        _packed_cache.clear()
    packed = PackedTracks(tracks, device, own)
    _packed_cache[key] = (packed, [weakref.ref(t.xy) for t in tracks])
    return packed


class TrackLossFused(torch.autograd.Function):
    """weight · LossTracking.compute_unweighted_loss (flowmap/loss/loss_tracking.py:28-61,
    flowmap/loss/loss.py:47) over all segments, from depth + intrinsics + extrinsics.

    Frame sharding (flowmap_amd/sharding.py): ``depth`` holds the rank's frames from ``frame0`` on,
    ``k`` / ``ext`` the whole video; ``packed`` was built with this rank's ``own`` source range;
    ``reducer`` sums the fp64 pair [Σρ, count] over the ranks.  The result is then the GLOBAL loss,
    the gradients this rank's share of it (autograd / FrameShard.sync sum them)."""

    @staticmethod
    def forward(ctx, depth, k, ext, packed: PackedTracks, weight, kind, delta, defer, frame0=0, reducer=None, fit_from=None):
        dev = check_device(depth, k, ext, packed.xy)
        depth, k, ext = _f32c(depth, "depth"), _f32c(k, "intrinsics"), _f32c(ext, "extrinsics")
        b, f_local, h, w = depth.shape
        f = ext.shape[1]
        if b != 1:
            raise RuntimeError("flowmap_amd: the fused tracking loss supports batch size 1")
        if tuple(k.shape) != (1, f, 3, 3) or tuple(ext.shape) != (1, f, 4, 4) or frame0 < 0 or frame0 + f_local > f:
            raise RuntimeError("flowmap_amd: intrinsics / extrinsics must cover the whole video and depth a window of it")
        if packed.last_frame > f:
            raise RuntimeError("flowmap_amd: a track segment extends past the last frame")
        kinv = intrinsics_inverse(k)
        ext_inv = torch.empty_like(ext)
        ws = torch.empty((packed.total, 9), dtype=torch.float32, device=dev)
        flag = (torch.zeros if packed.partial else torch.empty)((packed.total,), dtype=torch.uint8, device=dev)
        acc = torch.empty((f * 20,), dtype=torch.float64, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        scale = torch.empty((2,), dtype=torch.float32, device=dev)
        totals = torch.empty((2,), dtype=torch.float64, device=dev)
        need = any(ctx.needs_input_grad[:3])
        # every residual is evaluated once: the (unscaled) gradients come out of the same launch
        gws = torch.empty((packed.total, 3), dtype=torch.float32, device=dev) if need else None
        acc2 = torch.empty((f * 24,), dtype=torch.float64, device=dev) if need else None
        tgt = torch.empty((f, 12), dtype=torch.float32, device=dev)
        partial = torch.empty((max(packed.ntiles, 1) * ((packed.pmax + 63) // 64) * (packed.fmax * 14 + TRACK_TILE * 21),), dtype=torch.float32,
                              device=dev)  # per-wave sums (FM_TRACK_PARTIAL), reduced per frame without atomics
        sc = (h * w) ** 0.5
        with _guard(dev):
            st = stream_for(depth)
            call("fm_extrinsics_inverse", ptr(ext), f, ptr(ext_inv), st)
            if packed.ntiles > 0:
                call("fm_track_points", ptr(depth), int(frame0), ptr(kinv), ptr(ext), ptr(ext_inv), ptr(k), f, ptr(packed.xy), ptr(packed.vis),
                     ptr(packed.seg), ptr(packed.blocks), packed.nblocks, packed.pmax, h, w, ptr(ws), ptr(flag), ptr(tgt), st)
                call("fm_track_loss_fwd", ptr(ws), ptr(flag), ptr(packed.xy), ptr(packed.vis), ptr(packed.seg), ptr(packed.tiles),
                     packed.ntiles, packed.pmax, packed.fmax, ptr(ext), ptr(tgt), f, h, w, kind, float(delta), w / sc, h / sc,
                     float(weight), ptr(partial), ptr(acc), ptr(loss), ptr(scale), ptr(totals), ptr(gws), ptr(acc2), st)
        if packed.ntiles == 0:  # this rank owns no source frame of any segment
            acc.zero_()
            totals.zero_()
            loss.zero_()
            scale.copy_(torch.tensor([float(weight), 0.0], device=dev))
            if need:
                acc2.zero_()
        if reducer is not None:
            totals = reducer(totals)
            den = torch.where(totals[1] == 0, torch.ones_like(totals[1]), totals[1])  # `valid_sum or 1` (loss_tracking.py:61)
            loss = (float(weight) * totals[0] / den).to(torch.float32).reshape(1)
            scale = torch.stack([float(weight) / den, totals[1]]).to(torch.float32)
        ctx.save_for_backward(k, kinv, ext_inv, flag, acc, scale)
        ctx.grads = (gws, acc2) if need else None
        ctx.packed, ctx.dims, ctx.shapes = packed, (f, h, w), (tuple(depth.shape), tuple(k.shape), tuple(ext.shape))
        ctx.frame0 = int(frame0)
        ctx.fit_node = None
        ctx.plan = packed.scatter_plan(h, w) if ctx.needs_input_grad[0] else None  # built at the first step
        if defer and ctx.needs_input_grad[0]:
            node = _find_fit_node(ext if fit_from is None else fit_from, (depth.data_ptr(), depth._version, tuple(depth.shape)))
            if node is not None and node.needs_input_grad[0]:
                ctx.fit_node = node
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            raise RuntimeError("flowmap_amd: TrackLossFused gradients are single-use; run the forward again")
        k, kinv, ext_inv, flag, acc, scale = ctx.saved_tensors
        gws, acc2 = ctx.grads
        ctx.grads = None
        depth_shape, k_shape, ext_shape = ctx.shapes
        pk: PackedTracks = ctx.packed
        f, h, w = ctx.dims
        dev = kinv.device
        g = g.reshape(1).to(torch.float32).contiguous()
        g_ext = torch.empty(ext_shape, dtype=torch.float32, device=dev)
        g_k = torch.empty(k_shape, dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_track_loss_bwd", ptr(acc), ptr(acc2), ptr(scale), ptr(g), ptr(ext_inv), ptr(k), ptr(kinv), f, ptr(g_ext), ptr(g_k),
                 stream_for(kinv))

        frame0 = ctx.frame0

        plan = ctx.plan

        def scatter(buffer: Tensor) -> None:
            if plan is None:
                return
            pixels, first, entries, weights = plan
            with _guard(dev):  # the planned gather: no atomics (fm_track_scatter is the unplanned form)
                call("fm_depth_gather", ptr(gws), ptr(pixels), ptr(first), ptr(entries), ptr(weights), pixels.numel(), ptr(kinv),
                     ptr(scale), ptr(g), h, w, frame0, ptr(buffer), stream_for(buffer))

        g_depth = None
        if ctx.needs_input_grad[0]:
            node = ctx.fit_node
            ctx.fit_node = None
            if node is not None:
                node._fm_pending.append(scatter)  # lands in the buffer ProcrustesFit.backward returns
            else:
                g_depth = torch.zeros(depth_shape, dtype=torch.float32, device=dev)
                scatter(g_depth)
        need = ctx.needs_input_grad
        return g_depth, g_k if need[1] else None, g_ext if need[2] else None, None, None, None, None, None, None, None, None
