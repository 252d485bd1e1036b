"""Shared helpers for the parity tests: build/inject the host test double, convert
oracle inputs to flowmap_amd inputs, run one optimisation step through either side."""

from __future__ import annotations

import subprocess
from pathlib import Path

import torch

import flowmap_amd
from flowmap_amd import Batch, Flows, Tracks
from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
from flowmap_amd.loss.mapping import MappingHuberCfg, MappingL1Cfg, MappingL2Cfg
from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
from oracle import flowmap_oracle as orc

ROOT = Path(__file__).resolve().parent.parent
SIM_SRC = ROOT / "tests" / "host_sim" / "fm_host_sim.cpp"
SIM_LIB = ROOT / "tests" / "host_sim" / "libfm_host_sim.so"


def build_host_sim() -> Path:
    deps = [SIM_SRC, *(ROOT / "flowmap_amd" / "csrc").glob("*.h"), ROOT / "include" / "flowmap_hip.h"]
    if not SIM_LIB.exists() or SIM_LIB.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.run(
            ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", str(SIM_SRC), "-o", str(SIM_LIB)], check=True
        )
    return SIM_LIB


def mapping_cfg(kind: str, delta: float = 0.01):
    return {"huber": MappingHuberCfg("huber", delta), "l1": MappingL1Cfg("l1"), "l2": MappingL2Cfg("l2")}[kind]


def to_flows(of: orc.OFlows, device) -> Flows:
    return Flows(of.forward.to(device), of.backward.to(device), of.forward_mask.to(device), of.backward_mask.to(device))


def to_tracks(ot, device):
    if ot is None:
        return None
    return [Tracks(t.xy.to(device), t.visibility.to(device), t.start_frame) for t in ot]


def run_ours(depth, wlogit, focal, oflows, hw, num_points, otracks=None, kind="huber", device="cpu", lazy=True,
             flow_weight=1000.0, track_weight=100.0, loss_scale=1.0):
    """One step through flowmap_amd exactly as ModelWrapperOverfit.training_step would
    drive it.  Returns dict of loss values and parameter gradients (on CPU)."""
    f = depth.shape[0]
    flowmap_amd.set_lazy_surfaces(lazy)
    try:
        cfg = ModelCfg(
            BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
            IntrinsicsRegressedCfg("regressed", float(focal)),
            ExtrinsicsProcrustesCfg("procrustes", num_points, False),
        )
        model = Model(cfg, num_frames=f, image_shape=tuple(hw))
        model.backbone.depth.data = depth.clone()
        model.backbone.weights.data = wlogit.clone()
        model = model.to(device)
        batch = Batch(torch.zeros((1, f, 3, *hw), device=device))
        flows = to_flows(oflows, device)
        tracks = to_tracks(otracks, device)
        losses = [LossFlow(LossFlowCfg(0, flow_weight, "flow", mapping_cfg(kind)))]
        if tracks is not None:
            losses.append(LossTracking(LossTrackingCfg(0, track_weight, "tracking", mapping_cfg(kind))))
        out = model(batch, flows, 0)
        parts = [fn(batch, flows, tracks, out, 0) for fn in losses]
        total = sum(parts) * loss_scale if loss_scale != 1.0 else sum(parts)
        total.backward()
        return {
            "total": total.detach().cpu(),
            "loss_flow": parts[0].detach().cpu(),
            "loss_tracking": parts[1].detach().cpu() if tracks is not None else torch.zeros(()),
            "extrinsics": out.extrinsics.detach().cpu(),
            "g_depth": model.backbone.depth.grad.cpu(),
            "g_wlogit": model.backbone.weights.grad.cpu(),
            "g_focal": model.intrinsics.focal_length.grad.cpu(),
        }
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def run_oracle(depth, wlogit, focal, oflows, hw, num_points, otracks=None, kind="huber", dtype=torch.float32,
               flow_weight=1000.0, track_weight=100.0):
    d = depth.to(dtype).clone().requires_grad_(True)
    w = wlogit.to(dtype).clone().requires_grad_(True)
    fo = torch.tensor(float(focal), dtype=dtype, requires_grad=True)
    fl = orc.OFlows(*(x.to(dtype) for x in (oflows.forward, oflows.backward, oflows.forward_mask, oflows.backward_mask)))
    tr = None if otracks is None else [orc.OTracks(t.xy.to(dtype), t.visibility, t.start_frame) for t in otracks]
    total, parts, out = orc.explicit_depth_step(d, w, fo, fl, tuple(hw), num_points=num_points, tracks=tr, kind=kind,
                                                flow_weight=flow_weight, track_weight=track_weight)
    total.backward()
    return {
        "total": total.detach(),
        "loss_flow": parts["flow"].detach(),
        "loss_tracking": parts.get("tracking", torch.zeros(())).detach(),
        "extrinsics": out.extrinsics.detach(),
        "g_depth": d.grad,
        "g_wlogit": w.grad,
        "g_focal": fo.grad,
    }
