"""Drop-in for flowmap/model/extrinsics/extrinsics_procrustes.py — the call site of the
Procrustes/SVD pose fit."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from .. import _reference
from .projection import _align_surfaces


@dataclass
class ExtrinsicsProcrustesCfg:
    """flowmap/model/extrinsics/extrinsics_procrustes.py:15-19"""

    name: Literal["procrustes"]
    num_points: Optional[int]
    randomize_points: bool


_index_cache: dict = {}


def procrustes_indices(h: int, w: int, num_points: Optional[int], randomize: bool, device) -> Optional[Tensor]:
    """extrinsics_procrustes.py:34-51.  ``None`` stands for arange(h*w): the kernels then
    derive the pixel from the thread id and read coalesced.  The deterministic linspace
    selection is a constant of (h, w, P) and is cached."""
    if num_points is None:
        return None
    if randomize:
        return torch.randint(0, h * w, (num_points,), dtype=torch.int64, device=device)
    key = (h, w, num_points, str(device))
    if key not in _index_cache:
        if len(_index_cache) > 16:
            _index_cache.clear()
        # Built on the host so the selection is identical on every device (torch.linspace with
        # an integer dtype rounds differently on CPU and GPU); it is a constant, copied once.
        _index_cache[key] = torch.linspace(0, h * w - 1, num_points, dtype=torch.int64).to(device)
    return _index_cache[key]


class ExtrinsicsProcrustes(nn.Module):
    """flowmap/model/extrinsics/extrinsics_procrustes.py:22-59"""

    def __init__(self, cfg: ExtrinsicsProcrustesCfg, num_frames: Optional[int] = None) -> None:
        super().__init__()
        self.cfg = cfg
        self.num_frames = num_frames

    def forward(self, batch, flows, backbone_output, surfaces) -> Tensor:
        ref_cls = _reference.host_twin("ExtrinsicsProcrustes", surfaces, batch)
        if ref_cls is not None:  # host tensors after install(): the reference's own module (flowmap_amd/_reference.py); it has no parameters
            twin = self.__dict__.get("_fm_host_twin")
            if twin is None or type(twin) is not ref_cls:
                twin = self.__dict__["_fm_host_twin"] = ref_cls(self.cfg, self.num_frames)
            return twin.forward(batch, flows, backbone_output, surfaces)
        _, _, h, w, _ = surfaces.shape
        indices = procrustes_indices(h, w, self.cfg.num_points, self.cfg.randomize_points, surfaces.device)
        # Align the depth maps using a Procrustes fit.
        # (the chain may come back unevaluated — LazyExtrinsics — while gradients are recorded and nothing has read it so far: a flow-only
        # training step reads the fit's relative poses, not the chain)
        return _align_surfaces(surfaces, flows.backward, backbone_output.weights, indices, lazy_ok=True)
