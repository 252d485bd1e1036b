"""Stand-in: the overfitting wrapper's LAYOUT (the reference's is a LightningModule, which this image does not have): a module that owns
the video, its flows / tracks, the model and the losses, whose ``training_step`` evaluates the model and the losses at ``global_step``,
logs each loss (and the focal-length error where the batch carries ground truth) and returns their sum, and whose
``configure_optimizers`` makes Adam at ``cfg.lr``.  ``fit_steps`` is the part of a trainer's loop the tests and the bench need:
training_step → zero_grad → backward → optimiser step → global_step + 1, in the order Lightning's automatic optimisation runs them."""
from dataclasses import dataclass

import torch
from torch import nn


@dataclass
class ModelWrapperOverfitCfg:
    lr: float
    patch_size: int


class ModelWrapperOverfit(nn.Module):
    def __init__(self, cfg, model, batch, flows, tracks, losses, visualizers):
        super().__init__()
        self.cfg = cfg
        self.model = model
        self.batch, self.flows, self.tracks = batch, flows, tracks
        self.losses = losses  # (a plain list, as the reference keeps them: the losses own no parameters)
        self.visualizers = visualizers
        self.global_step = 0
        self.logged = {}

    def log(self, name, value):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value  # (a trainer's logger keeps values, not their autograd history)

    def training_step(self, dummy):
        output = self.model(self.batch, self.flows, self.global_step)
        total = 0
        for fn in self.losses:
            value = fn.forward(self.batch, self.flows, self.tracks, output, self.global_step)
            self.log(f"train/loss/{fn.cfg.name}", value)
            total = total + value
        truth = self.batch.intrinsics
        if truth is not None:
            estimate = output.intrinsics
            self.log("train/intrinsics/fx_error", (truth[..., 0, 0].mean() - estimate[..., 0, 0].mean()).abs())
            self.log("train/intrinsics/fy_error", (truth[..., 1, 1].mean() - estimate[..., 1, 1].mean()).abs())
        return total

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.cfg.lr)

    def fit_steps(self, optimizer, steps):
        """`steps` iterations of a trainer's loop; returns the last loss tensor."""
        loss = None
        for _ in range(steps):
            loss = self.training_step(None)
            loss = loss / 1  # (a trainer divides the step's loss by its gradient-accumulation factor — 1 here — and differentiates the quotient)
            self.kept = loss.detach().clone()  # (and keeps a detached copy for its progress bar)
            if optimizer is not None:
                optimizer.zero_grad(set_to_none=True)
            else:
                self.zero_grad(set_to_none=True)
            loss.backward()
            if optimizer is not None:
                optimizer.step()
            self.global_step += 1
        return loss
