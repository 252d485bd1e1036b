"""Frame-pair sharding of the hot path across the GPUs of one node.

The reference has no sharding (every DDP rank optimises the whole video,
flowmap/overfit.py:94-108); this is new design.  Flow-loss terms of pair i touch only
frames i, i+1, so rank r owns a contiguous range of pairs [a_r, b_r) and the frames
[a_r, b_r] — the last frame is a one-frame HALO shared with rank r+1.  No collective
sits in the data path; per step there is

  * ONE packed all-reduce (sum) of [loss, the gradients of every shared parameter] — a few floats for
    regressed intrinsics (what the reference's DDP all-reduces for them, overfit.py:94-108); a shared module with
    millions of parameters goes through ``SharedGradientBuckets`` (25 MB buckets, reduced while backward still runs);
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
    """Per-rank communication of the sharded optimisation step.

    Everything the step issues is asynchronous on the device and allocation-free after the first step — the packed
    reduction buffer and the halo receive buffers are persistent — so the whole sharded step (collectives included: RCCL
    work is stream-ordered) can be captured in a hipGraph (flowmap_amd.GraphedStep): no host wait, no per-step
    ``torch.cat`` into a fresh tensor, no copy of the reduced values back into ``.grad`` (the parameters' gradients are
    re-pointed at views of the reduced buffer).

    ``proxy``: one rank's share of a `world`-GPU run on a single GPU (bench.py --share K): the process group has ONE member,
    so every collective executes as a one-rank RCCL operation, and the halo exchange — which has no peer — is stood in for by
    the local work it causes (a copy of the two boundary frames out, an accumulate of two received frames in)."""

    def __init__(self, rank: int = 0, world: int = 1, dist=None, group=None, proxy: bool = False, force_collectives: bool = False):
        self.rank, self.world, self.dist, self.group, self.proxy = rank, world, dist, group, proxy
        # a one-rank job normally issues no collective at all; True: it issues every one of them on its one-member group (the driver's GPU
        # boxes have one GPU: tests/test_gpu_rccl.py runs each collective call site of the sharded step on a real RCCL communicator this way)
        self.force_collectives = force_collectives
        self._halo = None  # (requests, recv_prev, recv_next, gradient) of the exchange in flight
        self._halo_by_hook = False  # that exchange was posted by the gradient hook (sync() then only completes it)
        self.defer_halo = False  # True: the gradient hook does not post the exchange (GraphedShardedStep: sync() posts it after the replay)
        self._halo_buffers = {}  # (shape, device) -> persistent receive (and proxy send) buffers
        self._packed = None  # (key, buffer, views): [loss, gradients of the shared parameters] reduced in place every step
        self._early = None  # enable_early_halo(): pixel lists, buffers and the requests of the early (dense) part in flight
        self._unit_flags = {}
        self._syncs = 0
        self.ghost_evaluations = 0  # steps whose shared frames' dense parts were evaluated here (enable_ghost_halo) instead of received

    @property
    def active(self) -> bool:
        return (self.world > 1 or self.force_collectives) and self.dist is not None

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
        * The halo exchange of dL/ddepth starts from a gradient hook, as soon as that gradient is final.
        * The depth parameter is told which of its frames are shared with a neighbour (``_fm_halo_frames``): an in-pass
          optimiser update (FusedAdam.fuse_depth_update) leaves those frames to a dense update after the exchange."""
        if not self.active:
            return
        from .model.intrinsics_softmin import IntrinsicsSoftmin

        intr = getattr(model, "intrinsics", None)
        if isinstance(intr, IntrinsicsSoftmin) and not self.proxy:
            intr.shard = self
        depth = getattr(getattr(model, "backbone", None), "depth", None)
        if depth is not None and depth.requires_grad:
            depth.register_post_accumulate_grad_hook(lambda param: None if self.defer_halo else self.start_halo_exchange(param.grad))
            halo = ([0] if self.rank > 0 else []) + ([depth.shape[0] - 1] if self.rank < self.world - 1 else [])
            depth.__dict__["_fm_halo_frames"] = tuple(halo)

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
    def _packed_buffer(self, loss: Tensor, params):
        """The persistent [loss, shared gradients] buffer and one view per parameter, rebuilt only when the parameter set
        changes (which parameters carry a gradient can change once: the regressed focal length after the softmin hand-over)."""
        key = (loss.device, tuple((id(p), tuple(p.shape)) for p in params))
        if self._packed is None or self._packed[0] != key:
            total = 1 + sum(p.numel() for p in params)
            buffer = torch.zeros((total,), dtype=torch.float32, device=loss.device)
            views, offset = [], 1
            for p in params:
                views.append(buffer[offset : offset + p.numel()].view(p.shape))
                offset += p.numel()
            self._packed = (key, buffer, views, list(params))
        return self._packed[1], self._packed[2]

    def sync(self, loss: Tensor, shared_params, depth_param: Optional[Tensor], already_global: Optional[Tensor] = None) -> Tensor:
        """All-reduce the scalar loss and the gradients of every SHARED parameter (intrinsics, a small shared module:
        what the reference's DDP all-reduces, overfit.py:94-108) in ONE packed buffer; sum the halo frame's depth
        gradient with the neighbours (the exchange usually started from the gradient hook and overlaps the
        all-reduce).  Returns the global loss (detached).  ``loss`` is this rank's share (the flow term);
        ``already_global`` (the value ``tracking_loss`` returns) is added after the reduction.
        ``shared_params``: a parameter, a list of parameters, or None.  No-op for world == 1.

        Afterwards every shared parameter's ``.grad`` IS a view of the reduced buffer (no copy back); the buffer is
        rewritten by the next call — a reference to a gradient kept across steps sees the next step's values (clone it), and so
        does gradient accumulation across ``sync()`` calls without ``zero_grad``: the slot then holds [reduced previous + local
        new] and the sum over ranks counts the previous part `world` times (accumulate locally and call ``sync()`` once instead).
        ``zero_grad(set_to_none=False)`` is fine: the zeroed view is accumulated into in place and reduced where it is.
        A shared module with millions of parameters should use ``SharedGradientBuckets`` instead (bucketed, overlapped with backward)."""
        extra = 0.0 if already_global is None else already_global.detach()
        if not self.active:
            return loss.detach() if already_global is None else loss.detach() + extra
        dist = self.dist
        if shared_params is None:
            shared_params = []
        elif torch.is_tensor(shared_params):
            shared_params = [shared_params]
        with_grad = [p for p in shared_params if p.grad is not None]
        if depth_param is not None and depth_param.grad is not None:
            self.start_halo_exchange(depth_param.grad, from_sync=True)  # (no hook registered: start it now; a no-op when the hook has posted it)
        packed, views = self._packed_buffer(loss, with_grad)
        with torch.no_grad():
            # a gradient that is STILL the view of its slot (zero_grad(set_to_none=False) zeroed it in place and backward accumulated into
            # it) is already where the reduction reads it: torch.cat(..., out=packed) would be handed overlapping input and output
            aliased = [p.grad.data_ptr() == view.data_ptr() and p.grad.shape == view.shape and p.grad.is_contiguous() and p.grad.dtype == torch.float32
                       for p, view in zip(with_grad, views)]
            if not any(aliased):
                pieces = [loss.detach().reshape(1).to(torch.float32)] + [p.grad.reshape(-1).to(torch.float32) for p in with_grad]
                torch.cat(pieces, out=packed)  # one launch, into the persistent buffer
            else:
                packed[:1].copy_(loss.detach().reshape(1))
                fresh = [(view, p.grad) for p, view, same in zip(with_grad, views, aliased) if not same]
                if fresh:
                    torch._foreach_copy_([v for v, _ in fresh], [g.to(torch.float32) for _, g in fresh])
        work = dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.finish_halo_exchange()
        work.wait()  # (RCCL: a stream dependency, not a host wait; gloo: blocks)
        for p, view in zip(with_grad, views):
            p.grad = view
        return packed[0] + extra

    def _halo_buffer(self, like: Tensor, which: str) -> Tensor:
        key = (which, tuple(like.shape), like.device, like.dtype)
        buf = self._halo_buffers.get(key)
        if buf is None:
            buf = self._halo_buffers[key] = torch.zeros_like(like)
        return buf

    # -- early halo exchange ---------------------------------------------------------------
    # The boundary frame's dL/ddepth is (dense flow-loss part) + (a few thousand pixels the Procrustes fit / the tracks add to).  The
    # dense part — 3.7 MB per boundary and direction at 720p, ~50 us on an xGMI link — exists when the flow loss's FORWARD pass
    # ends; the sparse part when backward ends.  With enable_early_halo() the dense part is sent right after the flow pass, so
    # the link works under finalize + backward, and after backward only the values at the locally touched pixels travel
    # (~20 KB: latency).  Both neighbours add [dense + sparse] of the other: the same sums as the one-shot exchange.
    def unit_flag(self, device) -> Tensor:
        flag = self._unit_flags.get(device)
        if flag is None:
            from ._ops import register_unit_flag

            flag = self._unit_flags[device] = register_unit_flag(torch.zeros((1,), dtype=torch.int32, device=device))
        return flag

    @staticmethod
    def _touched_key(depth_param: Tensor):
        registry = depth_param.__dict__.get("_fm_touched") or {}
        return tuple((name, id(v), v._version) for name, v in sorted(registry.items()))

    def enable_early_halo(self, depth_param: Tensor) -> bool:
        """COLLECTIVE over neighbours — call it on every rank at the same point of the loop, once the static set of pixels other
        operators add gradient to is known (the Procrustes fit's plan exists from the second step on; a tracking loss has run or
        announced its pixels).  Neighbours exchange the pixel lists of their copies of the shared frames.  -> False (nothing
        changes) when the set is not known yet.  A later change of the set (a loss switched on) raises in the next step: call
        this again, on every rank, after that step."""
        if not self.active:
            return False
        registry = depth_param.__dict__.get("_fm_touched") or {}
        ready = torch.tensor([1 if "procrustes" in registry else 0], dtype=torch.int64, device=depth_param.device)
        if not self.proxy:
            self.dist.all_reduce(ready, op=self.dist.ReduceOp.MIN, group=self.group)  # all or nobody
        if int(ready.item()) == 0:
            return False
        frames, h, w = depth_param.shape
        n = h * w
        keys = torch.unique(torch.cat([v.reshape(-1) for v in registry.values()]))
        mine = {"prev": keys[keys < n].contiguous() if self.rank > 0 else None,
                "next": (keys[keys >= (frames - 1) * n] - (frames - 1) * n).contiguous() if self.rank < self.world - 1 else None}
        theirs = {"prev": None, "next": None}
        if self.proxy:
            theirs = dict(mine)
        else:
            dist = self.dist
            for side, peer in (("prev", self.rank - 1), ("next", self.rank + 1)):  # sizes, then lists (set-up only: blocking)
                if mine[side] is None:
                    continue
                size_out = torch.tensor([mine[side].numel()], dtype=torch.int64, device=depth_param.device)
                size_in = torch.zeros_like(size_out)
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, size_out, peer, self.group), dist.P2POp(dist.irecv, size_in, peer, self.group)]):
                    req.wait()
                theirs[side] = torch.empty((int(size_in.item()),), dtype=torch.int64, device=depth_param.device)
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine[side], peer, self.group), dist.P2POp(dist.irecv, theirs[side], peer, self.group)]):
                    req.wait()
        dev, dt = depth_param.device, depth_param.dtype
        frame = lambda: torch.zeros((h, w), dtype=dt, device=dev)  # noqa: E731
        self._early = {
            "key": self._touched_key(depth_param), "param": depth_param, "mine": mine, "theirs": theirs, "inflight": None,
            "send": {s_: frame() for s_ in mine if mine[s_] is not None}, "recv": {s_: frame() for s_ in mine if mine[s_] is not None},
            "delta_out": {s_: torch.zeros((mine[s_].numel(),), dtype=dt, device=dev) for s_ in mine if mine[s_] is not None},
            "delta_in": {s_: torch.zeros((theirs[s_].numel(),), dtype=dt, device=dev) for s_ in mine if mine[s_] is not None},
        }
        depth_param.__dict__["_fm_early_halo"] = self
        return True

    def ghost_halo_enabled(self) -> bool:
        return self._early is not None and self._early.get("ghost") is not None

    def set_proxy_ghost_flows(self, flows) -> None:
        """One-GPU proxy of the ghost halo: the shard's own first / last pair stand in for the neighbouring pairs (same sizes, same work)."""
        e = self._early
        if e is None or e.get("ghost") is None:
            return
        e["ghost"]["proxy_flows"] = {"prev": (flows.backward[0, 0].contiguous(), flows.backward_mask[0, 0].contiguous()),
                                     "next": (flows.forward[0, -1].contiguous(), flows.forward_mask[0, -1].contiguous())}

    def take_stashed_early(self):
        """GraphedShardedStep: the dense gradient the flow loss produced inside a capture / warm-up step (None without early halo)."""
        if self._early is None:
            return None
        dense = self._early.pop("stashed", None)
        return dense

    def disable_early_halo(self) -> None:
        if self._early is not None:
            self._early["param"].__dict__.pop("_fm_early_halo", None)
            self._early = None

    # -- ghost halo -------------------------------------------------------------------------
    # The dense part of a shared frame's dL/ddepth that the NEIGHBOUR computes is one direction of the flow loss of one pair with the
    # shared frame as its source: a function of that frame's depth (held on both ranks), the pair's constant flow and mask, K, and the
    # pair's pose.  With enable_ghost_halo() a rank is handed those constants once and receives the 64-byte pose each step — sent as soon
    # as the neighbour's fit has produced it, long before it is needed — and EVALUATES the term (fm_flow_ghost_terms, ~6 us for two
    # 720p frames) instead of receiving 3.7 MB per boundary and direction; the sparse rest (Procrustes / track pixels) travels after
    # backward exactly as with the early exchange.  Intrinsics must be shared by all frames (regressed / softmin), as they are here.
    def enable_ghost_halo(self, depth_param: Tensor, flows_prev=None, flows_next=None) -> bool:
        """COLLECTIVE over neighbours, like ``enable_early_halo`` (which it builds on: same pixel lists, same sparse round).
        ``flows_prev`` = (backward flow (H, W, 2), backward mask (H, W)) of the pair BEFORE this rank's first frame — rank−1's last pair —,
        ``flows_next`` = (forward flow, forward mask) of the pair AFTER its last frame — rank+1's first pair; None on the video's ends (and in
        the one-GPU proxy, where the rank's own boundary pairs stand in).  -> False when the static pixel set is not known yet."""
        if not self.enable_early_halo(depth_param):
            return False
        e = self._early
        for side, given in (("prev", flows_prev), ("next", flows_next)):
            if side in e["send"] and given is None and not self.proxy:
                raise ValueError(f"flowmap_amd.FrameShard.enable_ghost_halo: this rank shares its {'first' if side == 'prev' else 'last'} frame with a "
                                 f"neighbour: flows_{side} (the neighbouring pair's flow and mask) is needed")
        dev = depth_param.device
        pose = lambda: torch.eye(4, dtype=torch.float32, device=dev)  # noqa: E731
        # what the ghost terms are evaluated from, copied out of the step's own tensors by ONE launch per step (start_early_halo) into storage
        # that outlives the step — a step replayed as hipGraphs keeps its intermediates in the graphs' private pool, and nothing here
        # holds on to those: [my first pair's forward pose | my last pair's backward pose | (proxy stand-ins: first backward, last forward) | K | K⁻¹]
        pack = torch.zeros((4 * 16 + 2 * 9,), dtype=torch.float32, device=dev)
        e["ghost"] = {
            "pack": pack, "pose_out": {"prev": pack[0:16].view(4, 4), "next": pack[16:32].view(4, 4)},
            "proxy_in": {"prev": pack[32:48].view(4, 4), "next": pack[48:64].view(4, 4)}, "k": pack[64:73].view(3, 3), "kinv": pack[73:82].view(3, 3),
            "flows": {"prev": None if flows_prev is None else tuple(t.to(dev, torch.float32).contiguous() for t in flows_prev),
                      "next": None if flows_next is None else tuple(t.to(dev, torch.float32).contiguous() for t in flows_next)},
            "pose_in": {s_: pose() for s_ in e["send"]}, "context": None, "begun": False,
            # the dense gradient at the shared frames' touched pixels as the flow loss left it: what the sparse round subtracts (fm_halo_ghost_begin)
            "base": {s_: torch.zeros((e["mine"][s_].numel(),), dtype=torch.float32, device=dev) for s_ in e["send"]},
        }
        # (the whole-frame copies of the early exchange are not needed: nothing dense travels)
        e["send"] = {s_: None for s_ in e["send"]}
        e["recv"] = {s_: None for s_ in e["recv"]}
        return True

    def _exchange(self, pairs):
        """[(send buffer, receive buffer, peer)] -> requests (asynchronous); the proxy has no peer: nothing travels."""
        if self.proxy or not pairs:
            return []
        dist = self.dist
        ops = []
        for out, into, peer in pairs:
            ops.append(dist.P2POp(dist.isend, out, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, into, peer, self.group))
        return dist.batch_isend_irecv(ops)

    def start_early_halo(self, dense_grad: Tensor, depth_param: Tensor, context=None) -> None:
        """Called by the fused flow loss at the end of its forward pass with its dense dL/ddepth (1, F, H, W): copy the boundary
        frames (backward will add the sparse parts to the very same memory) and send them — or, with the ghost halo, send the boundary
        pairs' poses instead.  ``context``: what the ghost terms are evaluated from later — (t_fwd, t_bwd, k, kinv, norm, kind, delta) of
        the flow loss that calls."""
        e = self._early
        if e is None or e["param"] is not depth_param:
            return
        ghost = e.get("ghost")
        if ghost is not None and context is not None:
            # ONE launch (fm_halo_ghost_begin; inside a capture it becomes a node of the forward graph): the boundary pairs' poses, K and K⁻¹
            # into the persistent pack, and the dense gradient's values at the touched pixels of the shared frames — the baseline of the
            # sparse round — instead of copies of the two frames
            from ._lib import call, ptr, stream_for
            from ._ops import _guard

            t_fwd, t_bwd, k, kinv, norm, kind, delta = context
            ok = all(t.dtype == torch.float32 and t.is_contiguous() for t in (t_fwd, t_bwd, k, kinv, dense_grad)) and t_fwd.shape[0] == 1
            if not ok:
                raise RuntimeError("flowmap_amd.FrameShard (ghost halo): the flow loss's poses / intrinsics / dense gradient must be contiguous float32, batch 1")
            frames, h, w = dense_grad.shape[1:]
            count = lambda side: e["mine"][side].numel() if side in e["send"] else 0  # noqa: E731
            with _guard(dense_grad.device):
                call("fm_halo_ghost_begin", ptr(dense_grad), h * w, frames, ptr(e["mine"]["prev"]), count("prev"), ptr(ghost["base"].get("prev")),
                     ptr(e["mine"]["next"]), count("next"), ptr(ghost["base"].get("next")), ptr(t_fwd.detach()), ptr(t_bwd.detach()), t_fwd.shape[1],
                     ptr(k.detach()), ptr(kinv.detach()), ptr(ghost["pack"]), stream_for(dense_grad))
            ghost["context"] = (norm, kind, delta)
            ghost["begun"] = True
        if e.get("stash_only", False):  # GraphedShardedStep's warm-up and captures: nothing is sent; the step sends it between its two replays
            e["stashed"] = dense_grad
            return
        if self._touched_key(depth_param) != e["key"]:
            raise RuntimeError("flowmap_amd.FrameShard: the set of depth pixels other operators touch has changed since enable_early_halo() "
                               "(a loss was switched on?): call enable_early_halo() again, on every rank")
        if e["inflight"] is not None:  # a forward that never reached sync(): complete it first (every rank does)
            for req in e["inflight"]:
                req.wait()
        from ._lib import call, ptr, stream_for
        from ._ops import _guard

        frames, h, w = dense_grad.shape[1:]
        if ghost is None:
            with _guard(dense_grad.device):  # both boundary frames in one launch (fm_halo_copy)
                call("fm_halo_copy", ptr(dense_grad), h * w, frames, ptr(e["send"].get("prev")), ptr(e["send"].get("next")), stream_for(dense_grad))
        elif not ghost.get("begun", False):  # (the baseline and the poses come from the flow loss's own call, above — in a graphed step: inside its capture)
            raise RuntimeError("flowmap_amd.FrameShard: the ghost halo needs the flow loss's poses (a fused LossFlow on this shard's depth parameter)")
        if ghost is not None:
            # the neighbour evaluates my side of the shared frame itself: it needs the pose of MY boundary pair — towards rank−1 my first
            # pair's camera a -> a+1 (its forward term of that pair), towards rank+1 my last pair's camera b -> b−1 (its backward term)
            if ghost["context"] is None:
                raise RuntimeError("flowmap_amd.FrameShard: the ghost halo needs the flow loss's poses (a fused LossFlow on this shard's depth parameter)")
            if self.proxy:  # no peer: the rank's own boundary poses stand in for the neighbours'
                ghost["pose_in"] = ghost["proxy_in"]
            pairs = [(ghost["pose_out"][side], ghost["pose_in"][side], peer) for side, peer in (("prev", self.rank - 1), ("next", self.rank + 1)) if side in e["send"]]
        else:
            pairs = [(e["send"][side], e["recv"][side], peer) for side, peer in (("prev", self.rank - 1), ("next", self.rank + 1)) if side in e["send"]]
        e["inflight"] = self._exchange(pairs)

    def _start_sparse_halo(self, depth_grad: Tensor) -> None:
        """After backward: what the boundary frames gained since the early copy — non-zero only at the locally touched pixels."""
        from ._lib import call, ptr, stream_for
        from ._ops import _guard

        e = self._early
        frames, h, w = depth_grad.shape
        cnt = lambda side: e["mine"][side].numel() if side in e["send"] else 0  # noqa: E731
        ghost = e.get("ghost")
        with _guard(depth_grad.device):  # both sides' deltas in one launch (fm_halo_delta; against the compact baseline with the ghost halo)
            if ghost is not None:
                call("fm_halo_delta_sparse", ptr(depth_grad), h * w, frames, ptr(ghost["base"].get("prev")), ptr(e["mine"]["prev"]), cnt("prev"),
                     ptr(e["delta_out"].get("prev")), ptr(ghost["base"].get("next")), ptr(e["mine"]["next"]), cnt("next"), ptr(e["delta_out"].get("next")),
                     stream_for(depth_grad))
            else:
                call("fm_halo_delta", ptr(depth_grad), h * w, frames, ptr(e["send"].get("prev")), ptr(e["mine"]["prev"]), cnt("prev"), ptr(e["delta_out"].get("prev")),
                     ptr(e["send"].get("next")), ptr(e["mine"]["next"]), cnt("next"), ptr(e["delta_out"].get("next")), stream_for(depth_grad))
        pairs = [(e["delta_out"][side], e["delta_in"][side], peer) for side, peer in (("prev", self.rank - 1), ("next", self.rank + 1)) if side in e["send"]]
        self._halo = (self._exchange(pairs), None, None, depth_grad, depth_grad._version, True)

    def _finish_sparse_halo(self, depth_grad: Tensor) -> None:
        e = self._early
        for req in e["inflight"] or []:
            req.wait()
        e["inflight"] = None
        from ._lib import call, ptr, stream_for
        from ._ops import _guard

        frames, h, w = depth_grad.shape
        cnt = lambda side: e["theirs"][side].numel() if side in e["send"] else 0  # noqa: E731
        ghost = e.get("ghost")
        with _guard(depth_grad.device):  # the neighbours' dense parts, then their sparse parts: one launch each for both boundaries
            if ghost is not None:
                norm, kind, delta = ghost["context"]
                depth = e["param"].detach()
                flows = dict(ghost["flows"])
                if self.proxy:  # (stand-ins: the rank's own boundary pairs)
                    pf = ghost.get("proxy_flows")
                    if pf is None:
                        raise RuntimeError("flowmap_amd.FrameShard (proxy): set_proxy_ghost_flows() has not been called")
                    flows = pf
                first, last = flows.get("prev") if "prev" in e["send"] else None, flows.get("next") if "next" in e["send"] else None
                scale = (h * w) ** 0.5
                self.ghost_evaluations += 1
                call("fm_flow_ghost_terms",
                     ptr(depth[0]) if first else None, ptr(ghost["pose_in"].get("prev")) if first else None, ptr(first[0]) if first else None,
                     ptr(first[1]) if first else None, ptr(depth_grad[0]) if first else None,
                     ptr(depth[-1]) if last else None, ptr(ghost["pose_in"].get("next")) if last else None, ptr(last[0]) if last else None,
                     ptr(last[1]) if last else None, ptr(depth_grad[-1]) if last else None,
                     ptr(ghost["kinv"]), ptr(ghost["k"]), ptr(norm), None, h, w, int(kind), float(delta), w / scale, h / scale, stream_for(depth_grad))
            else:
                call("fm_halo_add", ptr(depth_grad), h * w, frames, ptr(e["recv"].get("prev")), ptr(e["recv"].get("next")), stream_for(depth_grad))
            call("fm_halo_scatter", ptr(depth_grad), h * w, frames, ptr(e["theirs"]["prev"]), ptr(e["delta_in"].get("prev")), cnt("prev"),
                 ptr(e["theirs"]["next"]), ptr(e["delta_in"].get("next")), cnt("next"), stream_for(depth_grad))
        self._syncs += 1
        capturing = depth_grad.is_cuda and torch.cuda.is_current_stream_capturing()  # (a host read cannot sit inside a hipGraph capture)
        # (read at the first exchange — a loop that scales its loss does so from its first step — and every 64th after it; a step replayed as a
        # hipGraph never reaches this Python: GraphedStep / GraphedShardedStep read the flag outside the replay, flowmap_amd/graph.py)
        if (self._syncs == 1 or self._syncs % 64 == 0) and not capturing and int(self.unit_flag(depth_grad.device).item()) != 0:
            self.unit_flag(depth_grad.device).zero_()  # reported: a later, correct loop starts clean
            raise RuntimeError("flowmap_amd.FrameShard: with enable_early_halo() the flow loss must reach backward() unscaled (the boundary frames' "
                               "gradient was sent before backward ran); disable_early_halo() for a scaled loss")

    def start_halo_exchange(self, depth_grad: Tensor, from_sync: bool = False) -> None:
        """depth_grad (F_local, H, W): the LAST local frame is rank+1's FIRST local frame; post the sends and
        receives of the two boundary frames (asynchronous).  ``from_sync``: the call sync() makes after backward — a no-op when
        the gradient hook of the same backward has already posted the exchange (an explicit per-exchange flag, not an inference
        from the gradient's version counter)."""
        if not self.active:
            return
        if self._halo is not None:
            if from_sync and self._halo_by_hook and self._halo[3] is depth_grad:
                # the gradient hook of THIS backward posted it; sync() only asks again.  An in-place edit of the gradient in between
                # (clip_grad_norm_, a scaling, a second backward without sync) would make the posted frame stale and the neighbour's part
                # be added to something else: refuse instead of exchanging twice
                if self._halo[4] != depth_grad._version:
                    raise RuntimeError("flowmap_amd.FrameShard: dL/ddepth was modified in place between backward and sync() while its halo exchange "
                                       "was in flight; modify gradients after sync(), or set shard.defer_halo = True so that sync() posts the exchange")
                return
            self.finish_halo_exchange()  # a backward that never reached sync(): complete it (every rank does) before the next one
        self._halo_by_hook = not from_sync
        if self._early is not None and self._early["inflight"] is not None:  # the dense part left after the flow pass: only the sparse rest now
            self._start_sparse_halo(depth_grad)
            return
        dist = self.dist
        ops, recv_prev, recv_next = [], None, None
        if self.rank > 0:
            recv_prev = self._halo_buffer(depth_grad[0], "prev")
            if self.proxy:
                self._halo_buffer(depth_grad[0], "send_prev").copy_(depth_grad[0])  # what the link would read
            else:
                ops.append(dist.P2POp(dist.isend, depth_grad[0].contiguous(), self.rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, recv_prev, self.rank - 1, self.group))
        if self.rank < self.world - 1:
            recv_next = self._halo_buffer(depth_grad[-1], "next")
            if self.proxy:
                self._halo_buffer(depth_grad[-1], "send_next").copy_(depth_grad[-1])
            else:
                ops.append(dist.P2POp(dist.isend, depth_grad[-1].contiguous(), self.rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, recv_next, self.rank + 1, self.group))
        self._halo = (dist.batch_isend_irecv(ops) if ops else [], recv_prev, recv_next, depth_grad, depth_grad._version, False)

    def finish_halo_exchange(self) -> None:
        """Wait for the exchange and add the neighbours' parts: both copies of a shared frame end up with
        the sum of the two partial gradients.  (Proxy: the receive buffers hold zeros — same launches, unchanged values.)"""
        if self._halo is None:
            return
        requests, recv_prev, recv_next, depth_grad, _, sparse = self._halo
        self._halo = None
        for req in requests:
            req.wait()
        if sparse:
            self._finish_sparse_halo(depth_grad)
            return
        if recv_prev is not None or recv_next is not None:
            if depth_grad.dim() == 3 and depth_grad.is_contiguous() and depth_grad.dtype == torch.float32 and (depth_grad.shape[0] > 1 or recv_prev is None or recv_next is None):
                from ._lib import call, ptr, stream_for
                from ._ops import _guard

                with _guard(depth_grad.device):  # both boundary frames in one launch
                    call("fm_halo_add", ptr(depth_grad), depth_grad.shape[1] * depth_grad.shape[2], depth_grad.shape[0], ptr(recv_prev), ptr(recv_next),
                         stream_for(depth_grad))
            else:
                if recv_prev is not None:
                    depth_grad[0].add_(recv_prev)
                if recv_next is not None:
                    depth_grad[-1].add_(recv_next)

    def exchange_halo(self, depth_grad: Tensor) -> None:
        self.start_halo_exchange(depth_grad)
        self.finish_halo_exchange()


class SharedGradientBuckets:
    """Gradient all-reduce of a SHARED module with many parameters under frame sharding — north_star's "RCCL all-reduce of
    the intrinsics/shared-backbone gradients"; in the reference the shared backbone is BackboneMidas (~10⁷ parameters,
    backbone_midas.py:42-127) and the mechanism Lightning's DDP strategy (overfit.py:94-108).

    Parameters are laid out, in REVERSE registration order (the order backward produces their gradients in, as DDP
    assumes), into flat persistent buckets of about ``bucket_mb`` (25 MB, DDP's default; over xGMI a ring all-reduce of a
    25 MB bucket on 8 GPUs moves 2·7/8·25 MB per link ≈ 0.3 ms at ≈150 GB/s, well above the ≈20 µs launch latency).
    A post-accumulate-grad hook copies a parameter's gradient into its slot and re-points ``.grad`` at the slot (one copy; the
    optimiser then reads the bucket); when the last slot of a bucket is filled its all-reduce is launched asynchronously, so
    the reduction of the late layers' buckets overlaps the backward of the early layers.  ``finish()`` (call it where
    ``FrameShard.sync`` is called) waits for every bucket and, for ``average=True``, divides by the world size —
    frame-pair sharding wants the SUM (every rank holds different loss terms), so the default is False.

    A parameter that gets no gradient in a step leaves its bucket incomplete: ``finish()`` zero-fills the missing slots and
    reduces the bucket then (no overlap for that bucket, same result)."""

    def __init__(self, shard: FrameShard, params, bucket_mb: float = 25.0, average: bool = False):
        self.shard, self.average = shard, average
        self.params = [p for p in params if p.requires_grad]
        limit = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets = []  # [buffer, [(param, view)], filled count, work]
        current, size = [], 0
        for p in reversed(self.params):
            if current and size + p.numel() > limit:
                self._close(current, size)
                current, size = [], 0
            current.append(p)
            size += p.numel()
        if current:
            self._close(current, size)
        self._where = {}
        for bi, bucket in enumerate(self.buckets):
            for p, view in bucket[1]:
                self._where[id(p)] = (bi, view)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _close(self, params, size):
        first = params[0]
        buffer = torch.zeros((size,), dtype=first.dtype, device=first.device)
        slots, offset = [], 0
        for p in params:
            slots.append((p, buffer[offset : offset + p.numel()].view(p.shape)))
            offset += p.numel()
        self.buckets.append([buffer, slots, 0, None])

    def _launch(self, bucket):
        dist = self.shard.dist
        bucket[3] = dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM, group=self.shard.group, async_op=True)

    def _on_grad(self, param):
        bi, view = self._where[id(param)]
        bucket = self.buckets[bi]
        if param.grad is not view:
            with torch.no_grad():
                view.copy_(param.grad)
            param.grad = view
        bucket[2] += 1
        if bucket[2] == len(bucket[1]) and self.shard.active:
            self._launch(bucket)

    def finish(self) -> None:
        for bucket in self.buckets:
            if bucket[3] is None and self.shard.active:
                with torch.no_grad():
                    for p, view in bucket[1]:
                        if p.grad is None:
                            view.zero_()
                        elif p.grad is not view:  # (a gradient that arrived without the hook firing)
                            view.copy_(p.grad)
                            p.grad = view
                self._launch(bucket)
            if bucket[3] is not None:
                bucket[3].wait()
                bucket[3] = None
            if self.average and self.shard.active:
                bucket[0].div_(self.shard.world)
            bucket[2] = 0

    def remove(self) -> None:
        for h in self._handles:
            h.remove()


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
    from .model.projection import LazySurfaces, _dense_extrinsics

    if global_step < loss_fn.cfg.enable_after:
        return torch.zeros((), dtype=torch.float32, device=model_output.depths.device)
    depths = model_output.surfaces.depths if isinstance(model_output.surfaces, LazySurfaces) else model_output.depths
    ranges, owns = _frame_layout(total_pairs, shard.world)
    lo, _ = ranges[shard.rank]
    frames = total_pairs + 1
    local_ext = _dense_extrinsics(model_output.extrinsics)  # (a LazyExtrinsics of a flow-only step so far: the tracking loss reads the chain)
    ext = _global_extrinsics(shard, local_ext, total_pairs)
    k = model_output.intrinsics
    if k.shape[1] != frames:  # intrinsics are shared by all frames (regressed / softmin): extend to the whole video
        k = k[:, :1].expand(1, frames, 3, 3).contiguous()
    packed = _ops.pack_tracks(tracks, depths.device, own=owns[shard.rank])

    def reducer(totals: Tensor) -> Tensor:
        shard.dist.all_reduce(totals, op=shard.dist.ReduceOp.SUM, group=shard.group)
        return totals

    return _ops.TrackLossFused.apply(depths, k, ext, packed, loss_fn.cfg.weight, _ops.MAPPING_KINDS[loss_fn.mapping.kind],
                                     loss_fn.mapping.delta, loss_fn.defer_depth_scatter, lo, reducer, local_ext)


FrameShard.global_extrinsics = lambda self, local_ext, total_pairs: _global_extrinsics(self, local_ext, total_pairs)
FrameShard.tracking_loss = lambda self, loss_fn, tracks, model_output, total_pairs, global_step=0: _tracking_loss(
    self, loss_fn, tracks, model_output, total_pairs, global_step)
