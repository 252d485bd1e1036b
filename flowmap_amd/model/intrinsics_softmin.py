"""IntrinsicsSoftmin (flowmap/model/intrinsics/intrinsics_softmin.py:41-141) with a fused
candidate sweep — SURVEY.md §8f rank 1, the reference's DEFAULT intrinsics for the first
1000 optimisation steps.

The reference evaluates every focal-length candidate by repeating depths, flows and weights
of the first two frames ``num_candidates`` (60) times, un-projecting 60 full-resolution
point clouds (1.3 GB at 720p), and only then gathering the 8192 sampled points it needs.
Here the images are read in place: the Procrustes kernels take ``batch_repeat`` (one image
batch entry serves all candidates, gradients accumulate atomically), and the flow error is
evaluated on the 8192 gathered points only — a handful of launches over ~6 MB.

Same constructor, buffers, sub-modules and state (``focal_length_candidates``,
``intrinsics_regressed``, ``window``) as the reference class, so checkpoints load unchanged.
"""

from __future__ import annotations

import weakref
from dataclasses import dataclass
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from .. import _ops, _reference
from .model import IntrinsicsRegressed, IntrinsicsRegressedCfg, focal_lengths_to_intrinsics
from .projection import LazyWeights


@dataclass
class RegressionCfg:
    """intrinsics_softmin.py:27-30"""

    after_step: int
    window: int


@dataclass
class IntrinsicsSoftminCfg:
    """intrinsics_softmin.py:33-40"""

    name: Literal["softmin"]
    num_procrustes_points: int
    min_focal_length: float
    max_focal_length: float
    num_candidates: int
    regression: Optional[RegressionCfg]


class IntrinsicsSoftmin(nn.Module):
    focal_length_candidates: Tensor

    def __init__(self, cfg: IntrinsicsSoftminCfg) -> None:
        super().__init__()
        self.cfg = cfg
        candidates = torch.linspace(cfg.min_focal_length, cfg.max_focal_length, cfg.num_candidates)
        self.register_buffer("focal_length_candidates", candidates, persistent=False)
        self._candidate_cache = None
        self._flow_cache = None
        self.shard = None  # flowmap_amd.sharding.FrameShard.prepare_model: the sweep runs on rank 0 and is broadcast
        if cfg.regression is not None:
            self.intrinsics_regressed = IntrinsicsRegressed(IntrinsicsRegressedCfg("regressed", 0.0))
            self.window = []

    # The reference draws torch.randperm(h*w)[:P] per step (intrinsics_softmin.py:90) — a full
    # device sort of 921 600 keys at 720p (0.27 ms) for 8192 samples.  _ops.random_subset evaluates
    # a keyed pseudo-random permutation at 0..P-1 instead: one launch, no sort, no host sync (the
    # seed comes from torch's CPU generator, so torch.manual_seed still reproduces a run).  Tests
    # override this hook to feed identical indices to both implementations.
    def _draw_indices(self, count: int, device) -> Tensor:
        # randperm(h*w)[:P] simply returns all h*w pixels when the image has fewer than P (intrinsics_softmin.py:90)
        return _ops.random_subset(count, min(self.cfg.num_procrustes_points, count), device)

    def _candidate_intrinsics(self, b: int, image_shape):
        """K of every candidate and its (b*n, 2, 3, 3) spread over the frame pair the sweep fits.
        The candidates are a constant buffer: built once per (shape, device), not once per step."""
        c = self.focal_length_candidates
        key = (b, tuple(image_shape), str(c.device), c.data_ptr(), c._version)
        if self._candidate_cache is None or self._candidate_cache[0] != key:
            n = c.numel()
            candidate_k = focal_lengths_to_intrinsics(c, image_shape)  # (n,3,3)
            k_pair = candidate_k[None, :, None].expand(b, n, 2, 3, 3).reshape(b * n, 2, 3, 3).contiguous()
            self._candidate_cache = (key, candidate_k, k_pair)
        return self._candidate_cache[1], self._candidate_cache[2]

    def _first_pair_flow(self, backward: Tensor) -> Tensor:
        """``flows.backward[:, :1]`` in memory of its own; flows are constants of the optimisation,
        so the copy is made once per flow tensor (for b = 1 the slice is already contiguous)."""
        first = backward[:, :1]
        if first.is_contiguous():
            return first
        key = (backward.data_ptr(), backward._version, tuple(backward.shape))
        if self._flow_cache is None or self._flow_cache[0]() is not backward or self._flow_cache[1] != key:
            self._flow_cache = (weakref.ref(backward), key, first.contiguous())
        return self._flow_cache[2]

    def forward(self, batch, flows, backbone_output, global_step: int) -> Tensor:
        ref_cls = _reference.host_twin("IntrinsicsSoftmin", batch)
        if ref_cls is not None:  # host tensors after install(): the reference's forward on this module's state (cfg, candidates, window,
            return ref_cls.forward(self, batch, flows, backbone_output, global_step)  # intrinsics_regressed: the same names, intrinsics_softmin.py:41-61)
        b, f, _, h, w = batch.videos.shape
        n = self.cfg.num_candidates
        device = batch.videos.device
        reg = self.cfg.regression

        # second stage: a single regressed focal length, initialised from the softmin window
        if reg is not None and global_step >= reg.after_step:
            if global_step == reg.after_step:
                self.intrinsics_regressed.focal_length.data = torch.stack(self.window).mean()
            _ops.note_leading_frames(backbone_output.depths, 0)  # the sweep no longer reads depth
            return self.intrinsics_regressed(batch, flows, backbone_output, global_step)

        def sweep():
            candidate_k, k_pair = self._candidate_intrinsics(b, (h, w))  # (n,3,3), (b*n,2,3,3)
            idx = self._draw_indices(h * w, device)
            bwd_01 = self._first_pair_flow(flows.backward)  # (b,1,h,w,2): the only pair the sweep looks at

            # ---- per-candidate Procrustes fit of frames (0, 1), images read in place -----------
            depths = _ops.LeadingFrames.apply(backbone_output.depths, 2)
            _ops.note_leading_frames(backbone_output.depths, 2)  # random pixels of frames 0 / 1: an in-pass depth update skips both frames
            weights = backbone_output.weights
            sens = 0.0
            if isinstance(weights, LazyWeights):
                weights_01, sens = _ops.LeadingFrames.apply(weights.logits, 1), weights.sensitivity
            else:
                weights_01 = _ops.LeadingFrames.apply(weights, 1)
            rel, _ = _ops.ProcrustesFit.apply(depths, k_pair, None, weights_01, bwd_01, idx, sens, n)  # (b*n,1,4,4): frame 1 -> frame 0

            # ---- pose-induced backward flow error per candidate (intrinsics_softmin.py:105-121): one
            # launch reads the sampled pixels' depth / weight / flow straight from the images
            if sens != 0.0 or not isinstance(weights, LazyWeights):
                score_weights, score_sens = weights_01, sens
            else:  # a LazyWeights with sensitivity 0 cannot be folded into the kernel
                score_weights, score_sens = _ops.LeadingFrames.apply(weights.materialize(), 1), 0.0
            # ... reduced to the softmin weights, the blended K of every frame and its inverse
            # (intrinsics_softmin.py:123-141).  The reference returns K as an expanded view; here it is
            # materialised once and every consumer of the step reads this tensor.
            return _ops.softmin_intrinsics(depths, score_weights, bwd_01, idx, candidate_k, rel.reshape(b * n, 4, 4), score_sens, f)

        if self.shard is not None and self.shard.active:  # frames (0, 1) of the VIDEO live on rank 0
            intrinsics, soft = self.shard.softmin_from_rank0(sweep, b, n, f, device)
        else:
            intrinsics, soft = sweep()


        if reg is not None and global_step >= reg.after_step - reg.window and self.training:
            self.window.append((self.focal_length_candidates * soft).sum().detach())

        return intrinsics

    def unnormalized_focal_lengths(self, image_shape) -> Tensor:
        """intrinsics_softmin.py:143-156"""
        h, w = image_shape
        return self.focal_length_candidates * (h * w) ** 0.5
