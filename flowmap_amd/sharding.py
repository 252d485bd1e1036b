"""Frame-pair sharding of the hot path across the GPUs of one node.

The reference has no sharding (every DDP rank optimises the whole video,
flowmap/overfit.py:94-108); this is new design.  Flow-loss terms of pair i touch only
frames i, i+1, so rank r owns a contiguous range of pairs [a_r, b_r) and the frames
[a_r, b_r] — the last frame is a one-frame HALO shared with rank r+1.  No collective
sits in the data path; per step there is

  * ONE packed all-reduce (sum) of [loss, dL/dfocal] — a few floats, latency-bound;
  * ONE neighbour exchange of the halo frame's dL/ddepth (N floats each way), because
    both owners of that frame hold a copy of its depth parameter;
  * at set-up, one all-reduce of the constant valid-mask sum so every shard normalises
    by the GLOBAL Σmask (loss_flow.py:70);
  * with the tracking loss (track windows of <= 41 frames straddle shard borders): one
    all-gather of the local poses (F·16 floats), one all-reduce of [Σρ, count], and in backward
    one all-reduce of the pose gradients (``FrameShard.tracking_loss``).

Everything goes through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor


def shard_pairs(num_pairs: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal pair ranges [start, end) for each rank (earlier ranks take
    the remainder).  Ranks beyond the number of pairs get empty ranges."""
    base, extra = divmod(num_pairs, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def shard_frames(pair_range: Tuple[int, int]) -> Tuple[int, int]:
    """Frames [first, last] (inclusive) a rank must hold for its pair range."""
    a, b = pair_range
    return a, b  # pairs [a, b) touch frames a .. b


class FrameShard:
    """Per-rank communication of the sharded optimisation step."""

    def __init__(self, rank: int = 0, world: int = 1, dist=None, group=None):
        self.rank, self.world, self.dist, self.group = rank, world, dist, group

    @property
    def active(self) -> bool:
        return self.world > 1 and self.dist is not None

    # -- set-up ---------------------------------------------------------------------------
    def reduce_valid_sum(self, vsum: Tensor) -> Tensor:
        if self.active:
            self.dist.all_reduce(vsum, op=self.dist.ReduceOp.SUM, group=self.group)
        return vsum

    def prepare_flow_loss(self, loss_fn, flows) -> None:
        """Make ``loss_fn`` (flowmap_amd.loss.LossFlow) normalise by the global Σmask."""
        if self.active:
            loss_fn.valid_sum_reducer = self.reduce_valid_sum

    # -- per step -------------------------------------------------------------------------
    def sync(self, loss: Tensor, shared_param: Optional[Tensor], depth_param: Optional[Tensor],
             already_global: Optional[Tensor] = None) -> Tensor:
        """All-reduce the scalar loss and the shared (intrinsics) gradient in one packed
        buffer; sum the halo frame's depth gradient with the neighbours.  Returns the
        global loss (detached).  ``loss`` is this rank's share (the flow term);
        ``already_global`` (the value ``tracking_loss`` returns) is added after the reduction.
        No-op for world == 1."""
        extra = 0.0 if already_global is None else already_global.detach()
        if not self.active:
            return loss.detach() if already_global is None else loss.detach() + extra
        dist = self.dist
        parts = [loss.detach().reshape(1).to(torch.float32)]
        if shared_param is not None and shared_param.grad is not None:
            parts.append(shared_param.grad.reshape(-1))
        packed = torch.cat(parts)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group)
        if len(parts) > 1:
            shared_param.grad.copy_(packed[1:].reshape(shared_param.grad.shape))
        if depth_param is not None and depth_param.grad is not None:
            self.exchange_halo(depth_param.grad)
        return packed[0] + extra

    def exchange_halo(self, depth_grad: Tensor) -> None:
        """depth_grad (F_local, H, W): the LAST local frame is rank+1's FIRST local frame.
        Both copies end up with the sum of the two partial gradients."""
        dist = self.dist
        ops, recv_prev, recv_next = [], None, None
        if self.rank > 0:
            recv_prev = torch.empty_like(depth_grad[0])
            ops.append(dist.P2POp(dist.isend, depth_grad[0].contiguous(), self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_prev, self.rank - 1, self.group))
        if self.rank < self.world - 1:
            recv_next = torch.empty_like(depth_grad[-1])
            ops.append(dist.P2POp(dist.isend, depth_grad[-1].contiguous(), self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, recv_next, self.rank + 1, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if recv_prev is not None:
            depth_grad[0].add_(recv_prev)
        if recv_next is not None:
            depth_grad[-1].add_(recv_next)


class _GatherPoses(torch.autograd.Function):
    """All-gather of every rank's local camera-to-world poses (F_local, 4, 4) into
    (world, F_max, 4, 4).  Each rank differentiates ITS loss terms w.r.t. all slots; the true
    gradient of slot r is the sum over ranks, so backward is one all-reduce and a slice."""

    @staticmethod
    def forward(ctx, local: Tensor, shard: "FrameShard", counts):
        fmax = max(counts)
        padded = torch.zeros((fmax, 4, 4), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        out = [torch.empty_like(padded) for _ in range(shard.world)]
        shard.dist.all_gather(out, padded, group=shard.group)
        ctx.shard, ctx.count = shard, local.shape[0]
        return torch.stack(out)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ctx.shard.dist.all_reduce(g, op=ctx.shard.dist.ReduceOp.SUM, group=ctx.shard.group)
        return g[ctx.shard.rank, : ctx.count], None, None


def _frame_layout(total_pairs: int, world: int):
    """Per rank: (first frame, last frame inclusive) and the frames it OWNS as tracking sources
    [first, end): the halo frame belongs to the next rank, the very last frame to the last rank."""
    ranges = [shard_frames(r) for r in shard_pairs(total_pairs, world)]
    own = [(lo, hi if r < world - 1 else hi + 1) for r, (lo, hi) in enumerate(ranges)]
    return ranges, own


def _global_extrinsics(shard: "FrameShard", local_ext: Tensor, total_pairs: int) -> Tensor:
    """(1, F_local, 4, 4) poses relative to the shard's first frame -> (1, F, 4, 4) poses of the
    whole video relative to frame 0 (get_extrinsics, projection.py:187-210, across shards): the
    pose of a shard's last (halo) frame is the transform to the next shard's first frame."""
    ranges, _ = _frame_layout(total_pairs, shard.world)
    counts = [hi - lo + 1 for lo, hi in ranges]
    gathered = _GatherPoses.apply(local_ext[0], shard, counts)  # (world, F_max, 4, 4)
    prefix = torch.eye(4, dtype=local_ext.dtype, device=local_ext.device)
    blocks = []
    for r, n in enumerate(counts):
        block = gathered[r, :n]
        last = r == shard.world - 1
        blocks.append(prefix @ (block if last else block[:-1]))
        prefix = prefix @ block[-1]
    return torch.cat(blocks)[None]


def _tracking_loss(shard: "FrameShard", loss_fn, tracks, model_output, total_pairs: int, global_step: int = 0) -> Tensor:
    """LossTracking over a frame-sharded video.  ``tracks`` carry GLOBAL frame
    indices; ``model_output`` is this rank's (depths, intrinsics, extrinsics of its own frames).
    Every rank evaluates the (source, target) pairs whose SOURCE frame it owns — it has that
    frame's depth — against targets anywhere in the segment, for which only poses and intrinsics are
    needed: the local poses are all-gathered and chained (_global_extrinsics), the [Σρ, count]
    pair is all-reduced, and the pose gradients travel back through the gather's backward.
    Returns the GLOBAL weighted loss; its gradients are this rank's share."""
    from . import _ops
    from .model.projection import LazySurfaces

    if global_step < loss_fn.cfg.enable_after:
        return torch.zeros((), dtype=torch.float32, device=model_output.depths.device)
    depths = model_output.surfaces.depths if isinstance(model_output.surfaces, LazySurfaces) else model_output.depths
    ranges, owns = _frame_layout(total_pairs, shard.world)
    lo, _ = ranges[shard.rank]
    frames = total_pairs + 1
    ext = _global_extrinsics(shard, model_output.extrinsics, total_pairs)
    k = model_output.intrinsics
    if k.shape[1] != frames:  # intrinsics are shared by all frames (regressed / softmin): extend to the whole video
        k = k[:, :1].expand(1, frames, 3, 3).contiguous()
    packed = _ops.pack_tracks(tracks, depths.device, own=owns[shard.rank])

    def reducer(totals: Tensor) -> Tensor:
        shard.dist.all_reduce(totals, op=shard.dist.ReduceOp.SUM, group=shard.group)
        return totals

    return _ops.TrackLossFused.apply(depths, k, ext, packed, loss_fn.cfg.weight, _ops.MAPPING_KINDS[loss_fn.mapping.kind],
                                     loss_fn.mapping.delta, loss_fn.defer_depth_scatter, lo, reducer, model_output.extrinsics)


FrameShard.global_extrinsics = lambda self, local_ext, total_pairs: _global_extrinsics(self, local_ext, total_pairs)
FrameShard.tracking_loss = lambda self, loss_fn, tracks, model_output, total_pairs, global_step=0: _tracking_loss(
    self, loss_fn, tracks, model_output, total_pairs, global_step)
