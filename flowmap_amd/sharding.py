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
    by the GLOBAL Σmask (loss_flow.py:70).

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
    def sync(self, loss: Tensor, shared_param: Optional[Tensor], depth_param: Optional[Tensor]) -> Tensor:
        """All-reduce the scalar loss and the shared (intrinsics) gradient in one packed
        buffer; sum the halo frame's depth gradient with the neighbours.  Returns the
        global loss (detached).  No-op for world == 1."""
        if not self.active:
            return loss.detach()
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
        return packed[0]

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
