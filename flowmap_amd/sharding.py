"""Frame-pair sharding of the hot path across the GPUs of one node.

The reference has no sharding (every DDP rank optimises the whole video,
flowmap/overfit.py:94-108); this is new design.  Flow-loss terms of pair i touch only
frames i, i+1, so rank r owns a contiguous range of pairs [a_r, b_r) and the frames
[a_r, b_r] — the last frame is a one-frame HALO shared with rank r+1.  No collective
sits in the data path; per step there is

  * ONE packed all-reduce (sum) of [loss, the gradients of every shared parameter] — a few floats for
    regressed intrinsics (what the reference's DDP all-reduces for them, overfit.py:94-108);
  * ONE neighbour exchange of the halo frame's dL/ddepth (N floats each way), because
    both owners of that frame hold a copy of its depth parameter; it is posted from a gradient hook the
    moment that gradient is final and overlaps the all-reduce;
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


class _FromRankZero(torch.autograd.Function):
    """Broadcast of a small tensor rank 0 computed (the softmin sweep's K and weights) with the matching
    backward: every rank's gradient w.r.t. the broadcast value is summed onto rank 0, where the sweep's own
    backward continues.  Other ranks pass a dummy of the same shape (its gradient is discarded)."""

    @staticmethod
    def forward(ctx, value: Tensor, shard: "FrameShard"):
        out = value.detach().clone().contiguous()
        shard.dist.broadcast(out, src=0, group=shard.group)
        ctx.shard = shard
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.shard.dist.reduce(g, dst=0, op=ctx.shard.dist.ReduceOp.SUM, group=ctx.shard.group)
        return (g if ctx.shard.rank == 0 else torch.zeros_like(g)), None


class FrameShard:
    """Per-rank communication of the sharded optimisation step."""

    def __init__(self, rank: int = 0, world: int = 1, dist=None, group=None):
        self.rank, self.world, self.dist, self.group = rank, world, dist, group
        self._halo = None  # (requests, recv_prev, recv_next, gradient) of the exchange in flight

    @property
    def active(self) -> bool:
        return self.world > 1 and self.dist is not None

    @staticmethod
    def owned_sources(total_pairs: int, world: int, rank: int) -> Tuple[int, int]:
        """Frames [first, end) whose tracking terms (as SOURCE frame) this rank evaluates."""
        return _frame_layout(total_pairs, world)[1][rank]

    # -- set-up ---------------------------------------------------------------------------
    def reduce_valid_sum(self, vsum: Tensor) -> Tensor:
        if self.active:
            self.dist.all_reduce(vsum, op=self.dist.ReduceOp.SUM, group=self.group)
        return vsum

    def prepare_flow_loss(self, loss_fn, flows) -> None:
        """Make ``loss_fn`` (flowmap_amd.loss.LossFlow) normalise by the global Σmask."""
        if self.active:
            loss_fn.valid_sum_reducer = self.reduce_valid_sum

    def prepare_model(self, model) -> None:
        """Shared (not frame-local) parts of the model under sharding.

        * The softmin intrinsics sweep (the reference's default for its first 1000 steps,
          intrinsics_softmin.py:85-131) fits frames (0, 1) OF THE VIDEO: they live on rank 0.  Run on every
          rank's local frames it would give every rank a different K, softmin window and hand-over focal
          length, and nothing downstream would ever reconcile them.  So rank 0 runs the sweep and broadcasts
          [K, softmin weights]; the gradient w.r.t. K is reduced back onto rank 0.  After the hand-over the
          regressed focal length is an ordinary shared parameter (sync() all-reduces its gradient).
        * The halo exchange of dL/ddepth starts from a gradient hook, as soon as that gradient is final."""
        if not self.active:
            return
        from .model.intrinsics_softmin import IntrinsicsSoftmin

        intr = getattr(model, "intrinsics", None)
        if isinstance(intr, IntrinsicsSoftmin):
            intr.shard = self
        depth = getattr(getattr(model, "backbone", None), "depth", None)
        if depth is not None and depth.requires_grad:
            depth.register_post_accumulate_grad_hook(lambda param: self.start_halo_exchange(param.grad))
            # the halo frames' gradient is complete only after the exchange: no in-pass optimiser update on a shard
            depth.__dict__["_fm_sharded"] = True

    def softmin_from_rank0(self, sweep, batch: int, candidates: int, frames: int, device):
        """``sweep() -> (K (b,frames,3,3), softmin weights (b,n))`` evaluated on rank 0 only -> the same pair on
        every rank (K differentiable on rank 0)."""
        if self.rank == 0:
            k, soft = sweep()
            packed = torch.cat([k[:, :1].reshape(-1), soft.reshape(-1).detach()])
        else:
            packed = torch.zeros((batch * 9 + batch * candidates,), dtype=torch.float32, device=device, requires_grad=True)
        packed = _FromRankZero.apply(packed, self)
        k = packed[: batch * 9].reshape(batch, 1, 3, 3).expand(batch, frames, 3, 3).contiguous()
        return k, packed[batch * 9 :].reshape(batch, candidates).detach()

    # -- per step -------------------------------------------------------------------------
    def sync(self, loss: Tensor, shared_params, depth_param: Optional[Tensor], already_global: Optional[Tensor] = None) -> Tensor:
        """All-reduce the scalar loss and the gradients of every SHARED parameter (intrinsics, a shared
        backbone: what the reference's DDP all-reduces, overfit.py:94-108) in ONE packed buffer; sum the
        halo frame's depth gradient with the neighbours (the exchange usually started from the gradient hook
        and overlaps the all-reduce).  Returns the global loss (detached).  ``loss`` is this rank's share (the
        flow term); ``already_global`` (the value ``tracking_loss`` returns) is added after the reduction.
        ``shared_params``: a parameter, a list of parameters, or None.  No-op for world == 1."""
        extra = 0.0 if already_global is None else already_global.detach()
        if not self.active:
            return loss.detach() if already_global is None else loss.detach() + extra
        dist = self.dist
        if shared_params is None:
            shared_params = []
        elif torch.is_tensor(shared_params):
            shared_params = [shared_params]
        with_grad = [p for p in shared_params if p.grad is not None]
        if depth_param is not None and depth_param.grad is not None and self._halo is None:
            self.start_halo_exchange(depth_param.grad)  # (no hook registered: start it now)
        packed = torch.cat([loss.detach().reshape(1).to(torch.float32)] + [p.grad.reshape(-1).to(torch.float32) for p in with_grad])
        work = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.finish_halo_exchange()
        work.wait()
        offset = 1
        for p in with_grad:
            n = p.grad.numel()
            p.grad.copy_(packed[offset : offset + n].reshape(p.grad.shape))
            offset += n
        return packed[0] + extra

    def start_halo_exchange(self, depth_grad: Tensor) -> None:
        """depth_grad (F_local, H, W): the LAST local frame is rank+1's FIRST local frame; post the sends and
        receives of the two boundary frames (asynchronous)."""
        if not self.active or self._halo is not None:
            return
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
        self._halo = (dist.batch_isend_irecv(ops) if ops else [], recv_prev, recv_next, depth_grad)

    def finish_halo_exchange(self) -> None:
        """Wait for the exchange and add the neighbours' parts: both copies of a shared frame end up with
        the sum of the two partial gradients."""
        if self._halo is None:
            return
        requests, recv_prev, recv_next, depth_grad = self._halo
        self._halo = None
        for req in requests:
            req.wait()
        if recv_prev is not None:
            depth_grad[0].add_(recv_prev)
        if recv_next is not None:
            depth_grad[-1].add_(recv_next)

    def exchange_halo(self, depth_grad: Tensor) -> None:
        self.start_halo_exchange(depth_grad)
        self.finish_halo_exchange()


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
