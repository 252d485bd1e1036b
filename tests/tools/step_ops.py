"""List what ONE optimisation step launches, in order: C-ABI calls and the torch ops around them
(runs on the CPU against the host test double; every listed op is one launch on the GPU).

    python tests/tools/step_ops.py [--tracking] [--softmin] [--adam]
"""
import argparse
import sys
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from helpers import build_host_sim, to_tracks  # noqa: E402

from flowmap_amd import _lib  # noqa: E402

_lib.set_library_for_testing(build_host_sim())
import flowmap_amd  # noqa: E402
from flowmap_amd import Batch, Flows, _ops  # noqa: E402
from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg  # noqa: E402
from flowmap_amd.loss.mapping import MappingHuberCfg  # noqa: E402
from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg  # noqa: E402
from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg  # noqa: E402
from oracle import flowmap_oracle as orc  # noqa: E402  (synthetic inputs only)

VIEW_OPS = ("view", "unsqueeze", "squeeze", "detach", "empty", "select.int", "slice", "expand", "reshape", "alias", "permute",
            "as_strided", "t.default", "transpose", "_local_scalar_dense", "randint", "lift_fresh", "unbind", "split")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracking", action="store_true")
    ap.add_argument("--softmin", action="store_true")
    ap.add_argument("--adam", action="store_true")
    args = ap.parse_args()
    f, h, w = 6, 24, 32
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, of = orc.synth_iid(f, h, w, seed=0)
    flows = Flows(of.forward, of.backward, of.forward_mask, of.backward_mask)
    intr = IntrinsicsSoftminCfg("softmin", 200, 0.5, 2.0, 8, RegressionCfg(1000, 100)) if args.softmin else IntrinsicsRegressedCfg("regressed", 0.85)
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), intr, ExtrinsicsProcrustesCfg("procrustes", 100, False)),
                  num_frames=f, image_shape=(h, w))
    model.backbone.depth.data, model.backbone.weights.data = depth, wlogit
    batch = Batch(torch.zeros((1, f, 3, 1, 1)).expand(1, f, 3, h, w))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    tracks, track_fn = None, None
    if args.tracking:
        tracks = to_tracks(orc.synth_tracks(f, h, w, seed=0, interval=2, radius=2, grid=4), "cpu")
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    opt = flowmap_amd.FusedAdam(model.parameters(), lr=1e-4) if args.adam else None

    def step():
        model.zero_grad(set_to_none=True)
        out = model(batch, flows, 0)
        loss = loss_fn(batch, flows, None, out, 0)
        if track_fn is not None:
            loss = loss + track_fn(batch, flows, tracks, out, 0)
        loss.backward()
        if opt is not None:
            opt.step()

    for _ in range(3):
        step()
    calls = []
    real_call = _lib.call

    class Trace(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(v in name for v in VIEW_OPS):
                calls.append("  " + name)
            return func(*args, **(kwargs or {}))

    def traced(name, *a):
        calls.append(name)
        return real_call(name, *a)

    for mod in (_ops, sys.modules.get("flowmap_amd.optim"), sys.modules.get("flowmap_amd.sharding")):
        if mod is not None and hasattr(mod, "call"):
            mod.call = traced
    with Trace():
        step()
    print("\n".join(calls))
    fm = sum(1 for c in calls if not c.startswith("  "))
    print(f"-- {len(calls)} launches ({fm} C-ABI calls, {len(calls) - fm} torch ops; some C-ABI calls are two or three kernels)")


if __name__ == "__main__":
    main()
