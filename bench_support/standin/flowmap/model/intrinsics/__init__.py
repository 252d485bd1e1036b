"""Stand-in: the intrinsics registry install() replaces entries of."""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)


@dataclass
class IntrinsicsRegressedCfg:
    name: str
    initial_focal_length: float


class IntrinsicsRegressed(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.focal_length = nn.Parameter(torch.tensor(float(cfg.initial_focal_length)))

    def forward(self, batch, flows, backbone_output, global_step):
        b, f, _, h, w = batch.videos.shape
        return orc.focal_to_k(self.focal_length, (h, w)).expand(b, f, 3, 3)


@dataclass
class RegressionCfg:
    after_step: int
    window: int


@dataclass
class IntrinsicsSoftminCfg:
    name: str
    num_procrustes_points: int
    min_focal_length: float
    max_focal_length: float
    num_candidates: int
    regression: Optional[RegressionCfg]


class IntrinsicsSoftmin(nn.Module):
    """(a registry entry for install() to replace; the stand-in has no host sweep of its own)"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, batch, flows, backbone_output, global_step):
        raise NotImplementedError("bench_support/standin: no host softmin sweep")


INTRINSICS = {"regressed": IntrinsicsRegressed, "softmin": IntrinsicsSoftmin}


def get_intrinsics(cfg):
    return INTRINSICS[cfg.name](cfg)
