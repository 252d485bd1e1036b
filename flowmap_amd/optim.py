"""Adam for the explicit-depth overfit loop — SURVEY.md §8f rank 2.

``ModelWrapperOverfit.configure_optimizers`` (flowmap/model/model_wrapper_overfit.py:104-105)
builds ``torch.optim.Adam(self.parameters(), lr=cfg.lr)``.  At C1 the parameters are
275 M floats (depth + correspondence-weight logits) and the update streams 7.7 GB — more
than the fused loss.  ``FusedAdam`` is the same optimiser (constructor, ``param_groups``,
``state_dict`` layout: ``step`` / ``exp_avg`` / ``exp_avg_sq``, so checkpoints of either load
into the other) with the update done by ``fm_adam_step``: one streaming pass per tensor.
"""

from __future__ import annotations

from typing import Iterable, Tuple

import torch

from ._lib import check_device, torch_ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False, *, maximize: bool = False, capturable: bool = False) -> None:
        # argument checks and messages of torch.optim.Adam
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        if amsgrad or maximize:
            raise ValueError("flowmap_amd.FusedAdam: amsgrad / maximize are not implemented (the reference uses neither)")
        # capturable (as torch.optim.Adam(capturable=True)): the step counter is a device tensor and
        # the kernel derives the bias corrections from it, so step() can sit inside a hipGraph
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      capturable=bool(capturable)))
        self.counters = {"sparse_updates": 0, "in_pass_updates": 0}  # (tests, bench notes)
        self._support = {}  # parameter -> element list outside which exp_avg / exp_avg_sq are exactly zero, or "dense"
        self._scaled_flags = {}  # device -> int32 flag the flow loss's backward raises when its upstream gradient is not 1
        self.verify_unit_upstream_every = 64  # steps between (synchronising) reads of that flag; 0 = never
        self._in_pass = {}  # parameter -> (step number, element list, halo frames, the forward's DepthSink) of an update the fused flow loss has already applied

    # -- the depth parameter's update inside the fused flow loss (SURVEY.md §8f-2's end state) -----------------------
    def fuse_depth_update(self, param: torch.Tensor, enabled: bool = True, max_touched_fraction: float = 0.02) -> None:
        """Opt in: from now on the fused flow loss (flowmap_amd.loss.LossFlow on lazy surfaces) applies THIS optimiser's
        update of ``param`` (the explicit-depth parameter, backbone_explicit_depth.py:34-41) in its own pass over HBM —
        depth, exp_avg and exp_avg_sq rewritten in place, no dL/ddepth round trip, no separate pass over depth (48 B per
        pixel and frame instead of 32 + 28) — for every pixel that only the flow loss touches.  The pixels the Procrustes
        fit and the tracking loss read or add gradient to (a static set, a fraction of a per cent) are updated by
        ``step()`` from their complete gradient, with the same step number.  Same arithmetic as the separate update.

        What changes for the caller: the parameter moves during ``loss.forward`` instead of ``step()``; ``param.grad``
        is meaningful at those sparse pixels only; the loss must reach ``backward()`` unscaled (the update uses the
        gradient as the forward pass computes it) and cannot be differentiated twice.  It engages only when it can:
        a sparse planned Procrustes fit, no weight decay, not capturable, and at most
        ``max_touched_fraction`` of the elements touched by other operators — the element-list update moves a 64-byte line
        per 4-byte element (≈0.08 ms per million elements on an MI355X) while the fused pass saves ≈0.25 ms per 138 M
        elements, so with the tracking loss at 720p (4.4 % touched) the separate update is the faster one and is kept;
        otherwise the step runs as usual.  On a frame shard (flowmap_amd.sharding.FrameShard.prepare_model) the frames shared
        with a neighbour count as touched whole: their gradient is complete only after the halo exchange, and ``step()``
        updates them with one dense pass per frame; every interior frame is updated in the pass."""
        if not any(p is param for group in self.param_groups for p in group["params"]):
            raise ValueError("flowmap_amd.FusedAdam.fuse_depth_update: the parameter does not belong to this optimiser")
        if enabled:
            param.__dict__["_fm_fused_adam"] = self
            param.__dict__["_fm_fused_adam_fraction"] = float(max_touched_fraction)
        else:
            param.__dict__.pop("_fm_fused_adam", None)

    # -- parameters whose gradient is zero outside a static element list (the correspondence-weight logits) ----------
    def _sparse_update(self, p: torch.Tensor, group) -> bool:
        """dL/dweights of the planned Procrustes fit is zero outside P slots per pair, the same slots every step
        (extrinsics_procrustes.py:33-38 draws them once).  Where exp_avg and exp_avg_sq are zero and the gradient is
        zero, Adam leaves a parameter exactly where it is (0 / (0 + eps)), so updating the slot list alone IS the dense
        update — 28 B x 138 M elements per step saved at 150 x 720p.  Taken only when that premise is known to hold:
        the gradient is the GradArena's storage untouched since backward, no weight decay, and the moments are zero
        outside the list (a fresh state, or verified once after dense steps)."""
        note = p.__dict__.get("_fm_sparse_grad")
        if note is None or group["weight_decay"] != 0 or group.get("capturable"):
            self._support[p] = "dense"
            return False
        arena, elements = note
        if not arena.holds(p.grad) or p.dtype != torch.float32 or not p.is_contiguous():
            self._support[p] = "dense"
            return False
        state = self.state[p]
        if len(state) == 0:
            state["step"] = torch.tensor(0.0, dtype=torch.float32)
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._support[p] = elements
        support = self._support.get(p, "dense")
        if support is not elements:
            if isinstance(support, str) and support == "dense-nonzero":
                return False
            # once: are the moments zero everywhere else?  (earlier dense steps saw the same sparse gradient)
            outside = torch.ones((p.numel(),), dtype=torch.bool, device=p.device)
            outside[elements] = False
            clean = not bool(((state["exp_avg"].reshape(-1) != 0) & outside).any() | ((state["exp_avg_sq"].reshape(-1) != 0) & outside).any())
            self._support[p] = elements if clean else "dense-nonzero"
            if not clean:
                return False
        state["step"] += 1
        beta1, beta2 = group["betas"]
        torch_ops().adam_step_elements(p, p.grad, state["exp_avg"], state["exp_avg_sq"], elements, int(state["step"].item()),
                                       float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), 0.0)
        self.counters["sparse_updates"] += 1
        return True

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._support.clear()  # loaded moments: where they are zero is verified again before the next element-list update
        self._in_pass.clear()

    def in_pass_pending(self, param: torch.Tensor) -> bool:
        return param in self._in_pass

    def begin_in_pass(self, depth: torch.Tensor, sink, t_fwd: torch.Tensor, t_bwd: torch.Tensor, exclude=()):
        """Called by the fused flow loss.  -> None when the update of the parameter behind ``depth`` cannot run inside its
        pass (the step then runs as usual), else ``(operator arguments, ticket)`` with the operator arguments
        (exp_avg, exp_avg_sq, touched mask, step number, [lr, beta1, beta2, eps]).  NOTHING is committed here: the caller hands
        the ticket to ``commit_in_pass`` once the operator has accepted its arguments and launched (an operator that
        refuses — a width that is not a multiple of 4, a misaligned buffer — leaves the optimiser exactly as it was)."""
        from . import _ops

        param = depth if depth._base is None else depth._base
        group = next((g for g in self.param_groups if any(p is param for p in g["params"])), None)
        if (group is None or group["weight_decay"] != 0 or group.get("capturable") or not param.is_contiguous() or param.dtype != torch.float32
                or depth.data_ptr() != param.data_ptr() or depth.numel() != param.numel()):
            return None
        if depth.dim() != 4 or depth.shape[-1] % 4 != 0 or depth.data_ptr() % 16 != 0:
            return None  # the 16-byte vector path of the fused pass does not apply (fm_flow_loss_fused_adam would refuse)
        if param in self._in_pass:
            # the parameter has ALREADY moved in an earlier forward whose step() never came: a forward that was not followed by
            # backward() + step() (a loss recomputed for logging, an exception, gradient accumulation over several forwards).
            # Continuing would silently drop that step's update of every pixel outside the element list.
            raise RuntimeError(
                "flowmap_amd.FusedAdam: the fused flow loss has already applied this step's depth update (fuse_depth_update), but "
                "optimizer.step() has not run since.  With fuse_depth_update every grad-enabled forward must be followed by backward() "
                "and step(); evaluate under torch.no_grad() for logging, and switch fuse_depth_update off for gradient accumulation.")
        registry = param.__dict__.get("_fm_touched", {})
        if not torch_ops().flow_loss_parks(t_fwd, t_bwd, sink, "softmin" in registry):
            return None  # the gradient would not travel through the step's DepthSink (poses not from the fit, ...)
        if "procrustes" not in registry:
            return None  # the fit's backward is not planned (yet): the pixels it reads are not known
        halo = tuple(param.__dict__.get("_fm_halo_frames", ()))  # frame sharding: frames whose gradient is complete only after the exchange
        union = _ops.touched_elements(depth, halo, exclude)  # (``exclude``: consumers whose gradient the pass itself absorbs — the tap exchange)
        if union is None:
            return None
        elements, mask = union
        if elements.numel() > param.__dict__.get("_fm_fused_adam_fraction", 0.02) * param.numel():
            return None  # too many elements would go through the list: the dense update is cheaper
        state = self.state[param]
        if len(state) == 0:
            state["step"] = torch.tensor(0.0, dtype=torch.float32)
            state["exp_avg"] = torch.zeros_like(param, memory_format=torch.preserve_format)
            state["exp_avg_sq"] = torch.zeros_like(param, memory_format=torch.preserve_format)
        if state["exp_avg"].data_ptr() % 16 != 0 or state["exp_avg_sq"].data_ptr() % 16 != 0:
            return None
        step = int(state["step"].item()) + 1
        beta1, beta2 = group["betas"]
        flag = self._scaled_flags.get(param.device)
        if flag is None:  # raised by the loss's backward when its upstream gradient is not 1 (the in-pass update used the unscaled one)
            flag = self._scaled_flags[param.device] = _ops.register_unit_flag(torch.zeros((1,), dtype=torch.int32, device=param.device))
        args = (state["exp_avg"].view(depth.shape), state["exp_avg_sq"].view(depth.shape), mask, step,
                [float(group["lr"]), float(beta1), float(beta2), float(group["eps"])], flag)
        return args, (param, step, elements, halo, sink)

    def commit_in_pass(self, ticket) -> None:
        """The fused pass has launched with the arguments of ``begin_in_pass``: the step counter advances and ``step()`` will
        finish the update (element list, halo frames) from the complete gradient."""
        param, step, elements, halo, sink = ticket
        self.state[param]["step"] += 1
        self._in_pass[param] = (step, elements, halo, sink)
        self.counters["in_pass_updates"] += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p in self._in_pass:  # the fused flow loss has updated every other element in its own pass
                    step, elements, halo, sink = self._in_pass.pop(p)
                    if p.grad is None or not sink.in_pass_confirmed():
                        raise RuntimeError("flowmap_amd.FusedAdam: the flow loss applied the depth update (fuse_depth_update) but its backward() "
                                           "never ran: the gradient of the pixels left to step() does not exist")
                    state = self.state[p]
                    grad = p.grad.contiguous()
                    hyper = (float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), 0.0)
                    for frame in halo:  # frame sharding: the frames shared with a neighbour, dense, from the exchanged (summed) gradient
                        torch_ops().adam_step(p[frame], grad[frame], state["exp_avg"][frame], state["exp_avg_sq"][frame], step, None, *hyper)
                    if sink.tap_absorbed() and not sink.tap_confirmed():
                        raise RuntimeError("flowmap_amd.FusedAdam: the flow pass applied the depth update with the tracking loss's gradient absorbed at its taps "
                                           "(the tap exchange), but that tracking loss never reached backward(): the update of those pixels used a gradient "
                                           "that was not part of the loss.  Evaluate the losses the same way every step, or set flowmap_amd._ops.options.tap_exchange = False.")
                    torch_ops().adam_step_elements(p, grad, state["exp_avg"], state["exp_avg_sq"], elements, step, *hyper)
                    tap_plan = p.__dict__.get("_fm_tap_plan")
                    if tap_plan is not None and tap_plan.pending_in_pass:
                        tap_plan.tag(p)  # the compact tap image the pass left is the parameter's as of now (read around the elements just updated)
                    every = self.verify_unit_upstream_every
                    first = self.counters["in_pass_updates"] == 1  # a loop that scales its loss does so from its first step: one read there catches it at once
                    if every and (first or step % every == 0) and int(self._scaled_flags[p.device].item()) != 0:
                        self._scaled_flags[p.device].zero_()  # reported: a later, correct loop starts clean
                        raise RuntimeError("flowmap_amd.FusedAdam: with fuse_depth_update the loss must reach backward() unscaled — the depth "
                                           "update inside the flow pass used the gradient of the loss itself, but backward() delivered an upstream "
                                           "gradient other than 1 (a scaled or averaged loss, a GradScaler): parameters and optimiser state of "
                                           f"the last {1 if first else every} step(s) are not those of torch.optim.Adam.  Switch fuse_depth_update off for this loop.")
                    continue
                if p.grad is None:
                    continue
                grad = p.grad
                if self._sparse_update(p, group):
                    continue
                if grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if p.dtype != torch.float32 or grad.dtype != torch.float32:
                    raise RuntimeError("flowmap_amd.FusedAdam: parameters and gradients must be float32")
                if not p.is_contiguous():
                    raise RuntimeError("flowmap_amd.FusedAdam: parameters must be contiguous")
                check_device(p, grad)
                state = self.state[p]
                capturable = bool(group.get("capturable", False))
                if len(state) == 0:
                    # host counter as torch's default Adam; device counter when capturable
                    state["step"] = torch.tensor(0.0, dtype=torch.float32, device=p.device if capturable else "cpu")
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                # (the operator bumps the version counters of p, exp_avg and exp_avg_sq, as torch.optim.Adam's in-place ops do)
                torch_ops().adam_step(p, grad, state["exp_avg"], state["exp_avg_sq"], 0 if capturable else int(state["step"].item()),
                                      state["step"] if capturable else None, float(group["lr"]), float(beta1), float(beta2),
                                      float(group["eps"]), float(group["weight_decay"]))
        return loss
