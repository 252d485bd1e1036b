"""'final ATE vs ref' at the headline chain length: 150 frames (149 chained poses), flow + tracking losses, the reference's
default intrinsics schedule — the softmin candidate sweep handing over to a regressed focal length
(config/overfit.yaml:24-31, intrinsics_softmin.py:63-141) — and Adam, through the oracle (reference path restated on
the CPU) and through flowmap_amd from identical initial parameters.

The reference-path leg takes ~10 s per step on 8 host cores at 150 x 360x640, so the two legs run separately:

    python tests/tools/ate_full_chain.py --leg reference --out tests/golden/ate_150x360x640_reference.json     (CPU, anywhere)
    python tests/tools/ate_full_chain.py --leg ours --reference tests/golden/ate_150x360x640_reference.json    (GPU box)

Both legs build the same seeded scene (oracle.synth_scene / synth_tracks, CPU) and feed the softmin sweep the same
per-step index sets (torch.randperm on a CPU generator seeded with the step number).  The `ours` leg prints one JSON line
with both ATEs (misc/ate.py:7-25 restated as oracle.ate) and their difference.
`--device cpu` runs flowmap_amd on the host test double (tests only).
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import flowmap_oracle as orc  # noqa: E402

CANDIDATES = (0.5, 2.0)  # config/model/intrinsics/softmin.yaml: min / max focal length


def step_indices(step: int, n: int, count: int) -> torch.Tensor:
    """The softmin sweep's random pixels of one step (intrinsics_softmin.py:90: randperm(h*w)[:P]), the same for both legs."""
    return torch.randperm(n, generator=torch.Generator().manual_seed(1000 + step))[: min(count, n)]


def scene(args, device="cpu"):
    """``device``: where the scene is GENERATED (fp64 there, fp32 results on the host either way: oracle.synth_scene) — minutes on host cores
    at 150 x 720p, seconds on the GPU; the two differ by the rounding of the device's fp64 elementary functions, i.e. in about one fp32 value of
    10^9 by one ulp."""
    f, h, w = args.frames, args.height, args.width
    sc = orc.synth_scene(f, h, w, seed=args.seed, focal=0.85, depth_noise=args.noise, device=device)
    tracks = orc.synth_tracks(f, h, w, scene=sc, seed=args.seed, interval=5, radius=min(20, f), grid=args.track_grid)
    return sc, tracks


def config(args):
    return {k: getattr(args, k) for k in ("frames", "height", "width", "steps", "lr", "points", "noise", "seed", "track_grid", "softmin_points",
                                          "num_candidates", "after_step", "window", "trace_every", "no_softmin")} | {"tracking_after": getattr(args, "tracking_after", 0)}


def reference_leg(args):
    f, h, w = args.frames, args.height, args.width
    sc, tracks = scene(args)
    gt_pos = sc["extrinsics_gt"][:, :3, 3]
    d = sc["depth_init"].clone()
    if args.perturb != 0.0:  # the reference path's OWN sensitivity: the same run from initial depths moved by ~1 ulp
        d = d * (1.0 + args.perturb * torch.randn(d.shape, generator=torch.Generator().manual_seed(12345)))
    d.requires_grad_(True)
    wl = torch.zeros((f - 1, h, w), requires_grad=True)
    # IntrinsicsRegressed inside IntrinsicsSoftmin starts at 0 (intrinsics_softmin.py:60); --no-softmin: a regressed focal length from the start, 10 % off
    fo = torch.tensor(0.85 * 1.1 if args.no_softmin else 0.0, requires_grad=True)
    opt = torch.optim.Adam([d, wl, fo], lr=args.lr)
    cand = torch.linspace(*CANDIDATES, args.num_candidates)
    idx = orc.procrustes_indices((h, w), args.points)
    window, focal_trace, losses = [], [], []
    t0 = time.perf_counter()

    def forward(step):
        depth, weights = d[None], (100.0 * wl).sigmoid()[None]
        if args.no_softmin or step >= args.after_step:
            if step == args.after_step and not args.no_softmin:
                fo.data = torch.stack(window).mean()
            k = orc.focal_to_k(fo, (h, w)).expand(1, f, 3, 3)
            focal = float(fo.detach())
        else:
            k1 = orc.softmin_intrinsics(depth, weights, sc["flows"].backward, cand, step_indices(step, h * w, args.softmin_points), (h, w))
            k = k1[:, None].expand(1, f, 3, 3)
            focal = float(k1[0, 0, 0].detach()) * w / (h * w) ** 0.5  # = Σ soft·candidates (the window's entry, :133-136)
            if step >= args.after_step - args.window:
                window.append(torch.tensor(focal))
        out = orc.model_forward(depth, weights, k, sc["flows"], idx)
        loss = 1000.0 * orc.flow_loss(out.surfaces, out.extrinsics, k, sc["flows"], (h, w))
        if step >= getattr(args, "tracking_after", 0):  # loss.py:39-46: a constant 0 before LossCfgCommon.enable_after
            loss = loss + 100.0 * orc.tracking_loss(out.surfaces, out.extrinsics, k, tracks, (h, w))
        return loss, out, focal

    for step in range(args.steps):
        opt.zero_grad(set_to_none=True)
        loss, out, focal = forward(step)
        loss.backward()
        opt.step()
        focal_trace.append(focal)
        losses.append(float(loss.detach()))
        if step % 10 == 0:
            print(f"[reference] step {step}: loss {losses[-1]:.6f} focal {focal:.6f} ({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)
    with torch.no_grad():
        _, out, _ = forward(args.steps)
    pos = out.extrinsics[0, :, :3, 3]
    result = {
        "made_by": "python tests/tools/ate_full_chain.py --leg reference " + " ".join(f"--{k.replace('_', '-')} {v}" for k, v in config(args).items())
                   + (f" --perturb {args.perturb}" if args.perturb else ""),
        "config": config(args),
        "perturb": args.perturb,
        "ate_reference_path_cpu": orc.ate(gt_pos, pos),
        "final_loss_reference_path": losses[-1],
        "loss_trace": losses[:: args.trace_every],
        "focal_trace": focal_trace[:: args.trace_every],
        "focal_final": float(fo.detach()),
        "positions": pos.tolist(),
        "seconds": time.perf_counter() - t0,
        "torch_threads": torch.get_num_threads(),
    }
    Path(args.out).write_text(json.dumps(result))
    print(json.dumps({k: v for k, v in result.items() if k != "positions"}))


def ours_leg(args, quiet=False, scene_device="cpu", perturb=0.0, built=None):
    """-> the comparison record (also printed unless ``quiet``).  ``perturb``: relative Gaussian perturbation of the initial depths (the
    schedule's sensitivity seen through this implementation); ``built``: a (scene, tracks) pair from an earlier call, reused."""
    import flowmap_amd
    from flowmap_amd import Batch, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, Model, ModelCfg
    from helpers import to_flows, to_tracks

    ref = json.loads(Path(args.reference).read_text())
    args.tracking_after = 0  # (records made before round 6 carry no such key: the tracking loss from step 0)
    for key, value in ref["config"].items():  # the reference leg's configuration IS the configuration
        setattr(args, key, value)
    f, h, w = args.frames, args.height, args.width
    dev = torch.device(args.device)
    if dev.type == "cpu":
        from helpers import build_host_sim

        _lib.set_library_for_testing(build_host_sim())
    sc, otracks = built if built is not None else scene(args, scene_device)
    gt_pos = sc["extrinsics_gt"][:, :3, 3]
    flowmap_amd.set_lazy_surfaces(True)
    from flowmap_amd.model.model import IntrinsicsRegressedCfg

    intrinsics = (IntrinsicsRegressedCfg("regressed", 0.85 * 1.1) if args.no_softmin else
                  IntrinsicsSoftminCfg("softmin", args.softmin_points, *CANDIDATES, args.num_candidates, RegressionCfg(args.after_step, args.window)))
    cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), intrinsics, ExtrinsicsProcrustesCfg("procrustes", args.points, False))
    model = Model(cfg, num_frames=f, image_shape=(h, w))
    d0 = sc["depth_init"].clone()
    if perturb != 0.0:
        d0 = d0 * (1.0 + perturb * torch.randn(d0.shape, generator=torch.Generator().manual_seed(12345)))
    model.backbone.depth.data = d0
    model = model.to(dev)
    model.train()
    state = {"step": 0}
    if not args.no_softmin:
        model.intrinsics._draw_indices = lambda count, device: step_indices(state["step"], count, args.softmin_points).to(device)
    flows = to_flows(sc["flows"], dev)
    tracks = to_tracks(otracks, dev)
    batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    track_fn = LossTracking(LossTrackingCfg(args.tracking_after, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    opt = flowmap_amd.FusedAdam(model.parameters(), lr=args.lr)
    if args.in_pass:
        opt.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
    losses, focal_trace = [], []
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.steps):
        state["step"] = step
        opt.zero_grad(set_to_none=True)
        out = model(batch, flows, step)
        loss = loss_fn(batch, flows, None, out, step) + track_fn(batch, flows, tracks, out, step)
        loss.backward()
        opt.step()
        if step % args.trace_every == 0:
            losses.append(float(loss.detach()))
            focal_trace.append(float(out.intrinsics[0, 0, 0, 0].detach()) * w / (h * w) ** 0.5)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    final_loss = float(loss.detach())
    state["step"] = args.steps
    with torch.no_grad():
        out = model(batch, flows, args.steps)
    pos = out.extrinsics[0, :, :3, 3].cpu()
    ate_ours = orc.ate(gt_pos, pos)
    ref_pos = torch.tensor(ref["positions"])
    record = {
        "scene": f"synthetic consistent scene, {f} frames @ {h}x{w} ({f - 1} chained poses), seed {args.seed}, depth noise {args.noise}",
        "schedule": f"flow (1000) + tracking (100, {len(otracks)} segments x {args.track_grid ** 2} tracks" + (f", from step {args.tracking_after}" if args.tracking_after else "") + f"); softmin intrinsics ({args.num_candidates} candidates x "
                    f"{args.softmin_points} points) for {args.after_step} steps, window {args.window}, then the regressed focal length; Adam lr {args.lr}, {args.steps} steps"
                    if not args.no_softmin else f"flow (1000) + tracking (100, {len(otracks)} segments x {args.track_grid ** 2} tracks); regressed focal length from the start (10 % off); Adam lr {args.lr}, {args.steps} steps",
        "ate_reference_path_cpu": ref["ate_reference_path_cpu"], "ate_flowmap_amd": ate_ours, "ate_abs_diff": abs(ate_ours - ref["ate_reference_path_cpu"]),
        "max_position_diff": float((pos - ref_pos).abs().max()), "position_scale": float(ref_pos.abs().max()),
        "final_loss_reference_path": ref["final_loss_reference_path"], "final_loss_flowmap_amd": final_loss,
        "focal_final_reference_path": ref["focal_final"], "focal_final_flowmap_amd": float((model.intrinsics if args.no_softmin else model.intrinsics.intrinsics_regressed).focal_length.detach()),
        "loss_trace_max_rel_diff": max(abs(a - b) / abs(b) for a, b in zip(losses, ref["loss_trace"])),
        "focal_trace_max_abs_diff": max(abs(a - b) for a, b in zip(focal_trace, ref["focal_trace"])),
        "trace_every": args.trace_every,
        "loss_trace_rel_diffs": [round(abs(a - b) / abs(b), 9) for a, b in zip(losses, ref["loss_trace"])],
        "focal_trace_abs_diffs": [round(abs(a - b), 9) for a, b in zip(focal_trace, ref["focal_trace"])],
        "seconds_reference_path_cpu": ref["seconds"], "seconds_flowmap_amd": seconds, "device": str(dev),
        "optimizer": "reference path: torch.optim.Adam; flowmap_amd: flowmap_amd.FusedAdam"
                     + (f" with fuse_depth_update ({opt.counters['in_pass_updates']} of {args.steps} depth updates inside the flow pass)" if args.in_pass else ""),
        "reference_made_by": ref["made_by"],
        "perturb": perturb, "positions_flowmap_amd": pos.tolist(),
    }
    if not quiet:
        print(json.dumps({k: v for k, v in record.items() if k != "positions_flowmap_amd"}))
    record["_built"] = (sc, otracks)
    return record


def compare_leg(args):
    """Two runs of the REFERENCE path (the second from initial depths perturbed by --perturb, ~1 ulp): how far apart they end is the
    bar a second implementation can be held to — the schedule (softmin selector + Adam) amplifies rounding-level differences."""
    a, b = json.loads(Path(args.reference).read_text()), json.loads(Path(args.other).read_text())
    pa, pb = torch.tensor(a["positions"]), torch.tensor(b["positions"])
    n = min(len(a["loss_trace"]), len(b["loss_trace"]))
    print(json.dumps({
        "what": "reference path vs reference path from initial depths perturbed by " + str(b.get("perturb")) + " (relative, Gaussian)",
        "ate_reference_path_cpu": a["ate_reference_path_cpu"], "ate_perturbed_reference_path": b["ate_reference_path_cpu"],
        "ate_abs_diff": abs(a["ate_reference_path_cpu"] - b["ate_reference_path_cpu"]),
        "max_position_diff": float((pa - pb).abs().max()), "focal_final": [a["focal_final"], b["focal_final"]],
        "final_loss": [a["final_loss_reference_path"], b["final_loss_reference_path"]],
        "loss_trace_max_rel_diff": max(abs(x - y) / abs(x) for x, y in zip(a["loss_trace"][:n], b["loss_trace"][:n])),
        "focal_trace_max_abs_diff": max(abs(x - y) for x, y in zip(a["focal_trace"][:n], b["focal_trace"][:n])),
        "made_by": [a["made_by"], b["made_by"]],
    }))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=["reference", "ours", "compare"], required=True)
    ap.add_argument("--perturb", type=float, default=0.0, help="reference leg: relative Gaussian perturbation of the initial depths (sensitivity run)")
    ap.add_argument("--other", default=None, help="compare leg: the second reference-path record")
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "ate_150x360x640_reference.json"))
    ap.add_argument("--reference", default=str(ROOT / "tests" / "golden" / "ate_150x360x640_reference.json"))
    ap.add_argument("--device", default="cuda:0")
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
    ap.add_argument("--no-softmin", action="store_true", help="regressed intrinsics from the start (no candidate sweep): the schedule without its chaotic selector")
    ap.add_argument("--in-pass", action="store_true")
    ap.add_argument("--tracking-after", type=int, default=0, help="LossTrackingCfg.enable_after (config/loss/tracking.yaml:4-6: 50)")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    {"reference": reference_leg, "ours": ours_leg, "compare": compare_leg}[a.leg](a)
