"""``ModelWrapperOverfit.training_step`` (flowmap/model/model_wrapper_overfit.py:51-73) replayed as hipGraphs.

At the reference's default operating point (config/overfit.yaml:33-38: 150 frames of ≈ 180×240) one step is ≈ 0.35 ms of kernels that
take the host ≈ 0.45 ms to enqueue (Python glue, four operators, autograd's engine: DESIGN.md §3.10): the step is HOST-bound.
``flowmap_amd.install(graph=True)`` rebinds ``ModelWrapperOverfit.training_step`` to ``GraphedTraining``: the model's forward and the
losses are captured as one hipGraph, ``loss.backward()`` as a second one in the same memory pool (the pattern of
``torch.cuda.make_graphed_callables`` and of ``GraphedShardedStep``), and from then on ``training_step`` replays the first and the loss it
returns replays the second when the trainer calls ``backward()`` on it — whatever the trainer does in between (Lightning's
``optimizer_zero_grad`` between ``training_step`` and ``backward``, its hooks, its logging) stays ordinary Python.  The optimiser is
the caller's and runs eagerly.

What a replay cannot see is host-side control flow, so the graphs exist only while that control flow is constant — the PHASE of the
optimisation — and every call checks it:

* every part of the model and every loss is a class of this package (a reference backbone such as MiDaS, a regressed-extrinsics module or
  a foreign loss keeps the reference's eager ``training_step``);
* every loss is switched on (``global_step >= cfg.enable_after``, loss/loss.py:39-41) and the intrinsics module is past its hand-over
  (intrinsics_softmin.py:74-82: the softmin sweep records focal lengths on the host until ``regression.after_step``);
* tensors on the GPU, the HIP library loaded, gradients enabled, module in training mode, the same batch / flows / tracks objects, the same
  set of parameters requiring gradients, no in-pass Adam update (``FusedAdam.fuse_depth_update``: its step number is a host value), no hooks
  on the parameters' gradients and no data-parallel process group (a replay runs no hook), depth maps below 128 MB (see the end of this note).

When the phase changes (a loss switches on, the intrinsics hand over, ``eval()``), the graphs are dropped, the reference's own
``training_step`` runs — ``warmup`` times in the new phase, so that every cache the kernels' host side keeps is filled — and the new phase is
captured.  A capture that fails leaves the wrapper on the eager path for good, with one warning.

The gradients the captured backward writes live in the graphs' pool and are the same tensors every step: ``backward()`` re-points
``param.grad`` at them (a trainer that sets gradients to ``None`` or zeroes them in place between steps — every trainer — sees exactly
what the eager step gives; gradient ACCUMULATION over several ``training_step`` calls is not supported and a loss divided by anything but 1
raises).  Videos whose depth maps exceed the last-level cache (128 MB: where the tap exchange between the flow and tracking losses engages,
DESIGN.md §3.4) keep the eager step: there the kernels are HBM-bound, the host's enqueue time hides behind them, and whether the compact tap
image is current is a per-step decision of the host.
"""

from __future__ import annotations

import os
import warnings
from typing import Callable, List, Optional

import torch
import torch.distributed
from torch import Tensor

from . import _lib, _ops


class GraphedLoss(_ops.RootLoss):
    """The total loss a replayed ``training_step`` returns: ``backward()`` replays the captured backward pass."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        owner = self.__dict__.get("_fm_graphed_training")
        if owner is None:
            return super().backward(gradient, retain_graph, create_graph, inputs)
        if gradient is not None or create_graph or inputs is not None:
            raise RuntimeError("flowmap_amd.install(graph=True): the loss of a replayed training_step takes a plain backward() — no gradient=, "
                               "create_graph= or inputs= (the backward pass is a captured hipGraph).  install(graph=False) for anything else.")
        owner.replay_backward(self)

    def _only_by_one(self, other, what):
        if isinstance(other, (int, float)) and not isinstance(other, bool) and float(other) == 1.0:
            return self
        raise RuntimeError(f"flowmap_amd.install(graph=True): the loss of a replayed training_step was {what} {other!r}; its backward pass is a "
                           "captured hipGraph whose seed is 1 (gradient accumulation / loss scaling need install(graph=False)).")

    def __truediv__(self, other):
        return self._only_by_one(other, "divided by")

    def __mul__(self, other):
        return self._only_by_one(other, "multiplied by")

    __rmul__ = __mul__

    def _only_plus_zero(self, other):
        if isinstance(other, (int, float)) and not isinstance(other, bool) and float(other) == 0.0:
            return self
        raise RuntimeError("flowmap_amd.install(graph=True): a term was added to the loss of a replayed training_step; its backward pass is a "
                           "captured hipGraph of the losses the step computed.  install(graph=False) for a step that adds its own terms.")

    def __add__(self, other):
        return self._only_plus_zero(other)

    __radd__ = __add__

    def __sub__(self, other):
        return self._only_plus_zero(other)

    def __rsub__(self, other):
        raise RuntimeError("flowmap_amd.install(graph=True): the loss of a replayed training_step cannot be negated (captured backward pass).")

    def __neg__(self):
        return self.__rsub__(0)


_classes = None


def _our_classes():
    """(backbone, extrinsics, softmin intrinsics, regressed intrinsics, losses): the classes of this package a captured step may be made of."""
    global _classes
    if _classes is None:
        from .loss import LossFlow, LossTracking
        from .model.backbone import BackboneExplicitDepth
        from .model.extrinsics_procrustes import ExtrinsicsProcrustes
        from .model.intrinsics_softmin import IntrinsicsSoftmin
        from .model.model import IntrinsicsRegressed

        _classes = (BackboneExplicitDepth, ExtrinsicsProcrustes, IntrinsicsSoftmin, IntrinsicsRegressed, (LossFlow, LossTracking))
    return _classes


class GraphedTraining:
    """One wrapper's replayed training step.  ``eager(wrapper, dummy)`` is the training_step it replaces (the reference's own)."""

    def __init__(self, eager: Callable, warmup: int = 2) -> None:
        self.eager = eager
        self.warmup = max(2, int(warmup))  # the second eager step of a phase builds the static scatter / tap plans (a sort: host-synchronising)
        self.key = None
        self.eager_left = 0
        self.forward_graph = self.backward_graph = None
        self.disabled: Optional[str] = None  # why this wrapper stays eager for good
        self.total: Optional[Tensor] = None
        self.values: List[Tensor] = []
        self.errors = None
        self.params: List[Tensor] = []
        self.grads: List[Optional[Tensor]] = []
        self.replays = self.captures = 0
        self.awaiting_backward = False
        self.verify_unit_upstream_every = 64
        self._aliases = None
        self._side_streams = None
        # The losses after the first on streams of their own inside the capture (FLOWMAP_AMD_GRAPH_STREAMS=1).  Measured at 180x240, flow + tracking:
        # 0.348-0.349 ms against 0.355-0.356 on one stream (profiles/r05_training_step_graph_streams_ab.txt) — track_pairs holds two 256-register
        # waves on nearly every SIMD, so the flow pass's blocks find room only at its edges.  2 % does not pay for a second stream's allocator
        # discipline: off by default.
        self.concurrent_losses = os.environ.get("FLOWMAP_AMD_GRAPH_STREAMS", "0").lower() not in ("0", "false", "off", "")

    # ---------------------------------------------------------------- the phase
    def on_device(self, wrapper) -> bool:
        """GPU tensors on the HIP library, gradients on, training mode: where a hipGraph can exist at all."""
        if _lib.using_test_double() or not torch.is_grad_enabled() or not wrapper.training:
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False  # (a data-parallel wrapper reduces gradients from hooks on the parameters' AccumulateGrad nodes: a replay runs none)
        return wrapper.batch.videos.device.type == "cuda"

    def signature(self, wrapper):
        """A hashable description of everything host-side that shapes the step, or None while that is not constant from step to step."""
        step = int(wrapper.global_step)
        model, losses = wrapper.model, wrapper.losses
        backbone_cls, extrinsics_cls, softmin_cls, regressed_cls, loss_cls = _our_classes()
        backbone, intrinsics, extrinsics = (getattr(model, name, None) for name in ("backbone", "intrinsics", "extrinsics"))
        if not isinstance(backbone, backbone_cls) or not isinstance(extrinsics, extrinsics_cls) or not all(isinstance(fn, loss_cls) for fn in losses):
            return None
        if isinstance(intrinsics, softmin_cls):
            reg = intrinsics.cfg.regression
            if reg is None or step <= reg.after_step:  # the sweep, the recording window, the hand-over step itself: host-side state
                return None
        elif not isinstance(intrinsics, regressed_cls):
            return None
        if any(step < fn.cfg.enable_after for fn in losses):
            return None
        # From the size at which the tap exchange engages (depth beyond the last-level cache: 128 MB, _ops.options.tap_exchange_min_bytes) the step is
        # bound by HBM, the host's enqueue time hides behind the kernels, and whether the compact tap image is current is a per-step decision
        # of the host that a replay could not make: the eager step stays.
        if backbone.depth.numel() * backbone.depth.element_size() >= _ops.options.tap_exchange_min_bytes:
            return None
        params = list(wrapper.parameters())
        if any("_fm_fused_adam" in p.__dict__ for p in params):  # FusedAdam.fuse_depth_update: the update's step number is a host value
            return None
        # hooks on a parameter's gradient (a frame shard's halo exchange, a data-parallel reducer, a user's register_hook) run in autograd's
        # backward; the captured backward is differentiated with respect to aliases of the parameters and replayed: it would run none of them
        if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in params):
            return None
        flows, tracks = wrapper.flows, wrapper.tracks
        # the constants of the optimisation as the eager path's caches key them: object, storage and version counter (an edit in place would
        # make the eager step re-derive its packed copies and plans; a replay could not)
        constants = tuple((t.data_ptr(), t._version) for t in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask))
        if tracks is not None:
            constants += tuple((t.xy.data_ptr(), t.xy._version, t.visibility.data_ptr(), t.visibility._version, int(t.start_frame)) for t in tracks)
        return (id(wrapper.batch), id(flows), id(tracks), constants,
                tuple((id(fn), float(fn.cfg.weight)) for fn in losses), tuple((id(p), p.requires_grad, p.data_ptr()) for p in params),
                str(wrapper.batch.videos.device))

    def phase(self, wrapper):
        return self.signature(wrapper) if self.on_device(wrapper) else None

    # ---------------------------------------------------------------- the step
    def __call__(self, wrapper, dummy):
        if self.disabled is not None:
            return self.eager(wrapper, dummy)
        key = self.phase(wrapper)
        if key is None or key != self.key:
            self.drop()
            self.key = key
            self.eager_left = self.warmup
        if key is None:
            return self.eager(wrapper, dummy)
        if self.eager_left > 0:
            self.eager_left -= 1
            return self.eager(wrapper, dummy)
        if self.forward_graph is None:
            try:
                self.capture(wrapper)
            except Exception as exc:  # noqa: BLE001 — whatever the capture could not digest: this wrapper runs the reference's step from here on
                self.drop()
                self.disabled = f"{type(exc).__name__}: {exc}"
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                warnings.warn(f"flowmap_amd.install(graph=True): capturing training_step failed ({self.disabled}); this wrapper keeps the eager step")
                return self.eager(wrapper, dummy)
        self.forward_graph.replay()
        self.replays += 1
        self.awaiting_backward = True
        self.log(wrapper)
        return self.total

    def log(self, wrapper) -> None:
        """model_wrapper_overfit.py:61,66-73 — the values are the graphs' output tensors, refreshed by the replay."""
        for loss_fn, value in zip(wrapper.losses, self.values):
            wrapper.log(f"train/loss/{loss_fn.cfg.name}", value)
        if self.errors is not None:
            wrapper.log("train/intrinsics/fx_error", self.errors[0])
            wrapper.log("train/intrinsics/fy_error", self.errors[1])

    def forward(self, wrapper, model_call=None, side_streams=None):
        """The body of training_step without its logging calls (model_wrapper_overfit.py:51-73).  ``side_streams`` (inside a capture): every loss
        after the first is evaluated on a stream of its own, forked where the model's forward ends and joined before the losses are summed —
        the fused losses are independent passes over the same model output (the flow pass is bound by memory, the tracking kernel by VALU issue),
        and as branches of one hipGraph they overlap at no cost to the host; autograd runs each loss's backward on the stream of its forward."""
        step = wrapper.global_step
        model_output = wrapper.model(wrapper.batch, wrapper.flows, step) if model_call is None else model_call(wrapper.batch, wrapper.flows, step)
        main = torch.cuda.current_stream() if side_streams else None
        used = []
        if side_streams:
            used = list(side_streams[: max(0, len(wrapper.losses) - 1)])
            for side in used:  # fork BEFORE the first loss is enqueued: the branches start where the model's forward ends
                side.wait_stream(main)
        values = []
        for index, loss_fn in enumerate(wrapper.losses):
            if used and index > 0:
                with torch.cuda.stream(used[index - 1]):
                    values.append(loss_fn.forward(wrapper.batch, wrapper.flows, wrapper.tracks, model_output, step))
            else:
                values.append(loss_fn.forward(wrapper.batch, wrapper.flows, wrapper.tracks, model_output, step))
        for side in used:
            main.wait_stream(side)
        total = 0
        for loss in values:
            total = total + loss
        errors = None
        truth = getattr(wrapper.batch, "intrinsics", None)
        if truth is not None:
            k = model_output.intrinsics
            errors = ((truth[..., 0, 0].mean() - k[..., 0, 0].mean()).abs(), (truth[..., 1, 1].mean() - k[..., 1, 1].mean()).abs())
        return total, values, errors

    @staticmethod
    def alias_parameters(named):
        """name -> a new leaf over the parameter's storage that carries what this package keeps on the parameter object."""
        aliases = {}
        for name, p in named:
            alias = p.detach().requires_grad_(True)
            alias.__dict__.update(p.__dict__)
            aliases[name] = alias
        return aliases

    def forward_on_aliases(self, wrapper, aliases, side_streams=None):
        return self.forward(wrapper, lambda *args: torch.func.functional_call(wrapper.model, aliases, args, strict=False), side_streams)

    def capture(self, wrapper) -> None:
        device = wrapper.batch.videos.device
        named = [(name, p) for name, p in wrapper.model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        if {id(p) for p in self.params} != {id(p) for p in wrapper.parameters() if p.requires_grad}:
            raise RuntimeError("parameters outside wrapper.model require gradients")
        for p in self.params:
            # The trainer clears the previous step's gradients between training_step and backward(); the backward captured HERE must see them
            # cleared too: a live gradient over the GradArena's storage makes the fit's backward take fresh zeros (csrc/fm_torch.cpp: GradArena) —
            # captured, that is 4 bytes per weight logit filled on every replay.
            p.grad = None
        _ops.flow_kernel_timing(False)  # event records do not belong in a graph
        torch.cuda.synchronize(device)
        # (while a capture runs, anything that draws random numbers must keep its state on the device: _ops.graph_capturable.  No eligible
        # phase draws any — the softmin sweep is the package's own step — and the switch is global, so it is on for the capture only: left
        # on, every later eager sweep of the process would draw from the device-side state instead of torch's generator)
        capturable_before, _ops.graph_capturable = _ops.graph_capturable, True
        try:
            self._capture(wrapper, device, named)
        finally:
            _ops.graph_capturable = capturable_before

    def _capture(self, wrapper, device, named) -> None:
        self.forward_graph, backward_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        side_streams = [torch.cuda.Stream(device) for _ in range(len(wrapper.losses) - 1)] if (self.concurrent_losses and len(wrapper.losses) > 1) else None
        self._side_streams = side_streams  # (kept: the backward graph's branches were captured on them)
        # The captured autograd graph must not end in the parameters' own AccumulateGrad nodes: those were made by the eager steps on the
        # trainer's stream and live as long as ANYTHING still holds a tensor of one of those steps (a logger, the trainer, a reference cycle
        # the frozen collector never visits); the engine orders every gradient that reaches such a node — torch.autograd.grad included —
        # against the node's stream, i.e. it would tie the capturing stream to the trainer's (torch warns, HIP faults).  So the step is
        # captured on ALIASES of the parameters — new leaves over the same storage, made under the capturing stream, carrying the
        # parameter's derived constants (plans, arenas: everything this package keeps on a tensor lives in its __dict__) — through
        # torch.func.functional_call, and differentiated with respect to them.
        with torch.cuda.graph(self.forward_graph, capture_error_mode="thread_local"):
            aliases = self.alias_parameters(named)
            total, self.values, self.errors = self.forward_on_aliases(wrapper, aliases, side_streams)
        if not isinstance(total, Tensor) or not total.requires_grad:
            raise RuntimeError("the step's total loss does not require gradients")
        seed = _ops.unit_seed(total.device) if (_ops.options.unit_seed and total.dim() == 0 and total.dtype == torch.float32) else None
        with torch.cuda.graph(backward_graph, pool=self.forward_graph.pool(), capture_error_mode="thread_local"):
            grads = torch.autograd.grad([total], [aliases[name] for name, _ in named], None if seed is None else [seed], allow_unused=True)
        self.backward_graph = backward_graph
        self.grads = list(grads)
        self._aliases = aliases  # (kept: the graphs read the parameters' storage through them)
        # the SAME tensor object re-typed (as _ops.as_root_loss does): logging, detach(), clone() see a plain scalar
        total = total.detach()  # (its autograd history was consumed by the captured backward)
        total.requires_grad_(True)
        total.__class__ = GraphedLoss
        total.__dict__["_fm_graphed_training"] = self
        self.total = total
        self.values = [v.detach() for v in self.values]
        if self.errors is not None:  # (un-detached, they kept the captured forward's autograd graph and its saved tensors alive)
            self.errors = tuple(e.detach() for e in self.errors)
        self.captures += 1

    def replay_backward(self, loss) -> None:
        if loss is not self.total or self.backward_graph is None:
            raise RuntimeError("flowmap_amd.install(graph=True): backward() on the loss of a training_step whose graphs were dropped since "
                               "(the phase of the optimisation changed between training_step and backward)")
        if not self.awaiting_backward:
            raise RuntimeError("flowmap_amd.install(graph=True): backward() twice on the loss of one replayed training_step")
        self.awaiting_backward = False
        self.backward_graph.replay()
        for p, g in zip(self.params, self.grads):
            if g is None:
                continue
            if p.grad is None or p.grad is g:
                p.grad = g
            else:  # somebody put a gradient there since the last zero_grad: add, as autograd would
                p.grad = p.grad + g
        every = self.verify_unit_upstream_every
        if every and (self.replays == 1 or self.replays % every == 0):
            _ops.check_unit_flags("a training_step replayed as hipGraphs (flowmap_amd.install(graph=True))")

    def drop(self) -> None:
        # (the loss of the dropped graphs keeps pointing here: a late backward() on it is refused in replay_backward)
        self.forward_graph = self.backward_graph = None
        self.total, self.values, self.errors, self.grads, self.params, self._aliases = None, [], None, [], [], None
        self.awaiting_backward = False


def make_training_step(eager: Callable, warmup: int = 2) -> Callable:
    """The method ``install(graph=True)`` binds as ``ModelWrapperOverfit.training_step``; the state lives on the wrapper object."""

    def training_step(self, dummy):
        state = self.__dict__.get("_fm_graphed_training")
        if state is None or state.eager is not eager:
            state = GraphedTraining(eager, warmup)
            object.__setattr__(self, "_fm_graphed_training", state)  # (nn.Module.__setattr__ would look for parameters / modules)
        return state(self, dummy)

    training_step.__wrapped__ = eager
    training_step.__doc__ = eager.__doc__
    return training_step
