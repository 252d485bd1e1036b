"""hipGraph capture of one optimisation step.

``GraphedStep`` captures the whole step — zeroed gradients, forward, losses, backward and,
optionally, the optimiser — into one hipGraph (``torch.cuda.CUDAGraph``) and replays it; every
kernel of this package launches on torch's current stream and allocates through torch, so it is
captured like any torch op.  What it buys is HOST time: a replay costs the CPU a few microseconds
instead of the 0.45-0.6 ms it takes to enqueue the launches of a step.  Where the host is the
bottleneck that is the step time — the reference's default resolution (150 frames of 180x240, flow +
tracking: 0.35 ms of kernels, 0.45-0.58 ms eager, 0.354 ms replayed; round 5), a rank
of an 8-GPU strong-scaling run, a slow host; at 720p the kernels are HBM-bound and the host's time hides
behind them.  ``flowmap_amd.install(graph=True)`` (flowmap_amd/training.py) gives the same replay to a
trainer that owns the loop: it wraps ``ModelWrapperOverfit.training_step``.

A frame-sharded step (flowmap_amd.sharding.FrameShard) is captured with its collectives: RCCL work is stream-ordered,
``FrameShard.sync`` keeps its buffers across steps and never waits on the host, so the all-reduce, the halo exchange and the
pose all-gather become nodes of the same graph.  That is where replaying pays: a rank of an 8-GPU strong-scaling run owns 19
frame pairs, its kernels take ~0.2 ms, and ~15 launches + 3 collectives of host time no longer hide behind them.

Requirements on ``fn``: static input tensors (parameters, flows, tracks: true for an overfit loop),
no host synchronisation, and for an optimiser inside it ``FusedAdam(..., capturable=True)``.
While a GraphedStep exists the softmin sweep draws its random pixels from a device-side state
(``_ops.graph_capturable``), so each replay still samples afresh.
"""

from __future__ import annotations

from typing import Callable

import torch
import torch.distributed

from . import _ops


class GraphedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 3, device=None, capture_error_mode=None) -> None:
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        if capture_error_mode is None:
            # the process group's watchdog thread queries events of earlier, un-captured collectives: legal beside a capture only
            # when the capture does not claim every thread
            capture_error_mode = "thread_local" if distributed else "global"
        self._previous = _ops.graph_capturable
        _ops.graph_capturable = True
        _ops.flow_kernel_timing(False)  # event records do not belong in a graph
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):  # fills every cache (packed constants, valid sums, indices, RNG state)
                for _ in range(max(2, warmup)):  # the second step builds the scatter plans (a sort, host-synchronising)
                    fn()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            if distributed:
                import time

                time.sleep(0.2)  # let the watchdog retire the warm-up collectives before the capture starts
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
                self.output = fn()
        finally:
            pass
        self.replays = 0
        self.verify_unit_upstream_every = 64  # replays between (synchronising) reads of the "scaled loss" device flags; 0 = never

    def __call__(self):
        """Replay the captured step; returns the tensors ``fn`` returned (refreshed in place)."""
        self.graph.replay()
        self.replays += 1
        # the Python that reads the "loss was scaled" flags (FusedAdam.step, FrameShard.sync) is not run by a replay: read them here,
        # after the first replay (a loop that scales its loss does so from the start) and every 64th
        every = self.verify_unit_upstream_every
        if every and (self.replays == 1 or self.replays % every == 0):
            _ops.check_unit_flags("a step replayed as a hipGraph (GraphedStep)")
        return self.output

    def close(self) -> None:
        _ops.graph_capturable = self._previous


class GraphedShardedStep:
    """A frame-sharded step with its COMPUTE replayed as hipGraphs and its collectives issued eagerly around them.

    ``forward() -> local loss`` runs zero_grad + the model + the flow loss of this rank's shard, ``loss.backward()`` the rest; neither
    contains a collective (the halo exchange's gradient hook is held back while this object exists: ``FrameShard.defer_halo``).
    They are captured as one graph — or, with ``FrameShard.enable_early_halo``, as TWO graphs sharing one memory pool (the pattern of
    torch.cuda.make_graphed_callables), so that the boundary frames' dense gradient, which exists when the forward graph ends, is sent
    between the two replays and travels under the backward graph (measured on the one-GPU proxy: the second replay and the early
    exchange's local copies / index operations cost ~40 us per step, about what they hide of a ~60 us transfer at 720p — it pays at
    1080p and beyond); ``FrameShard.sync`` then reduces [loss, shared
    gradients] and finishes the halo exchange as usual.  Per step the host enqueues two graph launches and the collectives while
    the GPU is still busy: what a rank of an 8-GPU strong-scaling run needs (its ~15 kernels take ~0.2 ms, eagerly enqueueing them
    ~0.45 ms), without RCCL inside a captured graph — the safe default for multi-rank runs; ``GraphedStep`` over the whole step
    (collectives captured too) saves the remaining host time.

    The gradient tensors the captured backward writes live in the graphs' memory pool and are the same every replay; ``sync``
    re-points the shared parameters' ``.grad`` at its reduction buffer, so they are restored before every sync."""

    def __init__(self, forward: Callable[[], object], shard, shared_params, depth_param, warmup: int = 3, device=None) -> None:
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.shard, self.shared, self.depth_param = shard, list(shared_params), depth_param
        shard.defer_halo = True
        if shard._early is not None:
            shard._early["stash_only"] = True
        self._previous = _ops.graph_capturable
        _ops.graph_capturable = True
        _ops.flow_kernel_timing(False)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(max(2, warmup)):
                forward().backward()
                shard.take_stashed_early()  # (warm-up: nothing is sent)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        import time

        time.sleep(0.2)  # let the process group's watchdog retire earlier collectives before the capture starts
        self.forward_graph, self.backward_graph = torch.cuda.CUDAGraph(), None
        if shard._early is None:  # nothing travels between forward and backward: ONE graph (a second replay costs ~15 us of launch latency)
            with torch.cuda.graph(self.forward_graph, capture_error_mode="thread_local"):
                self.loss = forward()
                self.loss.backward()
            self.early_dense = None
        else:
            self.backward_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.forward_graph, capture_error_mode="thread_local"):
                self.loss = forward()
            self.early_dense = shard.take_stashed_early()  # the flow loss's dense dL/ddepth (graph memory: the same tensor every replay), or None
            with torch.cuda.graph(self.backward_graph, pool=self.forward_graph.pool(), capture_error_mode="thread_local"):
                self.loss.backward()
        self.grads = [p.grad for p in self.shared]
        self.depth_grad = None if depth_param is None else depth_param.grad
        if shard._early is not None:
            shard._early["stash_only"] = False

    def __call__(self, already_global=None):
        self.forward_graph.replay()
        if self.early_dense is not None:
            self.shard.start_early_halo(self.early_dense, self.depth_param)
        if self.backward_graph is not None:
            self.backward_graph.replay()
        for p, g in zip(self.shared, self.grads):
            p.grad = g
        if self.depth_param is not None:
            self.depth_param.grad = self.depth_grad
        return self.shard.sync(self.loss, self.shared, self.depth_param, already_global=already_global)

    def close(self) -> None:
        self.shard.defer_halo = False
        _ops.graph_capturable = self._previous
