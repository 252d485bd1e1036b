"""The REFERENCE leg of "final ATE vs ref" (BASELINE.json's metric, second half), run by the reference ITSELF: dcharatan/flowmap
(mounted read-only at /root/reference) — its ``Model`` (explicit-depth backbone, ``IntrinsicsSoftmin`` handing over to the
regressed focal length, ``ExtrinsicsProcrustes``), its ``get_losses`` (flow + tracking), ``torch.optim.Adam`` as
``ModelWrapperOverfit.configure_optimizers`` builds it (model_wrapper_overfit.py:104-105) and its own ``compute_ate``
(misc/ate.py:7-25) — driven like ``ModelWrapperOverfit.training_step`` (model_wrapper_overfit.py:51-73) on the seeded synthetic
scene of tests/tools/ate_full_chain.py.

Runs only in the build container (the GPU box has no /root/reference); ~1 h on 6 host threads at 150 x 360x640:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_ate_reference.py [--out tests/golden/ate_150x360x640_imported_reference.json]

The record it writes is a committed fixture; ``python tests/tools/ate_full_chain.py --leg ours --reference <record>`` (GPU box) runs
flowmap_amd from the same initial parameters and compares.  The only thing patched in the reference is ``torch.randperm`` during
``IntrinsicsSoftmin.forward`` (intrinsics_softmin.py:90), so that both legs draw the same pixels at every step
(ate_full_chain.step_indices).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("FLOWMAP_REFERENCE", "/root/reference"))
sys.path[:0] = [str(HERE / "refstubs"), str(REF), str(HERE.parent), str(HERE.parent / "tests"), str(HERE.parent / "tests" / "tools")]

import torch  # noqa: E402

from flowmap.dataset.types import Batch  # noqa: E402
from flowmap.flow.flow_predictor import Flows  # noqa: E402
from flowmap.loss import get_losses  # noqa: E402
from flowmap.loss.loss_flow import LossFlowCfg  # noqa: E402
from flowmap.loss.loss_tracking import LossTrackingCfg  # noqa: E402
from flowmap.loss.mapping.mapping_huber import MappingHuberCfg  # noqa: E402
from flowmap.misc.ate import compute_ate  # noqa: E402
from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg  # noqa: E402
from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg  # noqa: E402
from flowmap.model.intrinsics.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg  # noqa: E402
from flowmap.model.model import Model, ModelCfg  # noqa: E402
from flowmap.tracking.track_predictor import Tracks  # noqa: E402

import ate_full_chain as chain  # noqa: E402  (scene + per-step index sets: the input generators shared with the `ours` leg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(HERE.parent / "tests" / "golden" / "ate_150x360x640_imported_reference.json"))
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--points", type=int, default=1000)
    ap.add_argument("--noise", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--track-grid", type=int, default=16)
    ap.add_argument("--softmin-points", type=int, default=8192)
    ap.add_argument("--num-candidates", type=int, default=60)
    ap.add_argument("--after-step", type=int, default=100)
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--trace-every", type=int, default=10)
    ap.add_argument("--no-softmin", action="store_true")
    ap.add_argument("--tracking-after", type=int, default=0,
                    help="LossTrackingCfg.enable_after (config/loss/tracking.yaml:4-6: 50): the tracking loss is a constant 0 before that step (loss.py:39-46)")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--perturb", type=float, default=0.0,
                    help="relative Gaussian perturbation of the initial depths (1e-7 ~ one fp32 ulp): the reference's OWN sensitivity under this schedule — "
                         "the bar a second implementation is held to.  Run it with --out <other file>, then --merge-sensitivity")
    ap.add_argument("--merge-sensitivity", default=None, metavar="PERTURBED_RECORD",
                    help="no optimisation: read --out (the unperturbed record) and PERTURBED_RECORD, write `self_sensitivity` into --out")
    args = ap.parse_args()
    if args.merge_sensitivity:
        a, b = json.loads(Path(args.out).read_text()), json.loads(Path(args.merge_sensitivity).read_text())
        pa, pb = torch.tensor(a["positions"]), torch.tensor(b["positions"])
        a["self_sensitivity"] = {
            "what": "the imported reference against itself from initial depths perturbed by " + str(b.get("perturb")) + " (relative, Gaussian)",
            "ate": a["ate_reference_path_cpu"], "ate_perturbed": b["ate_reference_path_cpu"],
            "ate_rel_diff": abs(a["ate_reference_path_cpu"] - b["ate_reference_path_cpu"]) / a["ate_reference_path_cpu"],
            "max_position_diff": float((pa - pb).abs().max()), "final_loss": [a["final_loss_reference_path"], b["final_loss_reference_path"]],
            "focal_final": [a["focal_final"], b["focal_final"]], "made_by": b["made_by"],
        }
        Path(args.out).write_text(json.dumps(a))
        print(json.dumps(a["self_sensitivity"]))
        return
    torch.set_num_threads(args.threads)
    f, h, w = args.frames, args.height, args.width
    sc, otracks = chain.scene(args)
    gt_pos = sc["extrinsics_gt"][:, :3, 3]

    intrinsics = (IntrinsicsRegressedCfg("regressed", 0.85 * 1.1) if args.no_softmin else
                  IntrinsicsSoftminCfg("softmin", args.softmin_points, *chain.CANDIDATES, args.num_candidates, RegressionCfg(args.after_step, args.window)))
    cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), intrinsics, ExtrinsicsProcrustesCfg("procrustes", args.points, False), True)
    model = Model(cfg, num_frames=f, image_shape=(h, w))
    d0 = sc["depth_init"].clone()
    if args.perturb != 0.0:
        d0 = d0 * (1.0 + args.perturb * torch.randn(d0.shape, generator=torch.Generator().manual_seed(12345)))
    model.backbone.depth.data = d0
    model.train()
    batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["scene"], ["synthetic"])
    flows = Flows(sc["flows"].forward, sc["flows"].backward, sc["flows"].forward_mask, sc["flows"].backward_mask)
    tracks = [Tracks(t.xy, t.visibility, t.start_frame) for t in otracks]
    losses = get_losses([LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)), LossTrackingCfg(args.tracking_after, 100.0, "tracking", MappingHuberCfg("huber", 0.01))])
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)

    real_randperm = torch.randperm

    def forward(step):
        # (the reference draws torch.randperm(h*w)[:P] from the global generator, intrinsics_softmin.py:90: hand it the step's seeded permutation)
        torch.randperm = lambda n, device=None, **kw: real_randperm(n, generator=torch.Generator().manual_seed(1000 + step)).to(device or "cpu")
        try:
            out = model(batch, flows, step)
        finally:
            torch.randperm = real_randperm
        return out

    loss_trace, focal_trace, t0 = [], [], time.perf_counter()
    for step in range(args.steps):
        opt.zero_grad(set_to_none=True)
        out = forward(step)
        total = 0
        for loss_fn in losses:  # model_wrapper_overfit.py:57-62
            total = total + loss_fn.forward(batch, flows, tracks, out, step)
        total.backward()
        opt.step()
        if step % args.trace_every == 0:
            loss_trace.append(float(total.detach()))
            focal_trace.append(float(out.intrinsics[0, 0, 0, 0].detach()) * w / (h * w) ** 0.5)
        if step % 10 == 0:
            print(f"[imported reference] step {step}: loss {float(total.detach()):.6f} focal {focal_trace[-1]:.6f} ({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)
    final_loss = float(total.detach())
    with torch.no_grad():
        out = forward(args.steps)
    pos = out.extrinsics[0, :, :3, 3]
    ate, _, _ = compute_ate(gt_pos, pos)
    regressed = model.intrinsics if args.no_softmin else model.intrinsics.intrinsics_regressed
    config = {k: getattr(args, k) for k in ("frames", "height", "width", "steps", "lr", "points", "noise", "seed", "track_grid", "softmin_points",
                                            "num_candidates", "after_step", "window", "trace_every", "no_softmin", "tracking_after")}
    result = {
        "made_by": "PYTHONDONTWRITEBYTECODE=1 python oracle/make_ate_reference.py " + " ".join(f"--{k.replace('_', '-')} {v}" for k, v in config.items() if k != "no_softmin")
                   + (" --no-softmin" if args.no_softmin else "") + (f" --perturb {args.perturb}" if args.perturb else ""),
        "reference_kind": "the imported reference (dcharatan/flowmap at /root/reference): flowmap.model.model.Model + flowmap.loss.get_losses + torch.optim.Adam + flowmap.misc.ate.compute_ate",
        "config": config,
        "perturb": args.perturb,
        "ate_reference_path_cpu": float(ate),
        "final_loss_reference_path": final_loss,
        "loss_trace": loss_trace,
        "focal_trace": focal_trace,
        "focal_final": float(regressed.focal_length.detach()),
        "positions": pos.tolist(),
        "seconds": time.perf_counter() - t0,
        "torch_threads": torch.get_num_threads(),
        "torch_version": torch.__version__,
    }
    Path(args.out).write_text(json.dumps(result))
    print(json.dumps({k: v for k, v in result.items() if k != "positions"}))


if __name__ == "__main__":
    main()
