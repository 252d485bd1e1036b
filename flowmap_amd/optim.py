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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
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
