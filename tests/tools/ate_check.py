"""Final-ATE comparison: optimise the same synthetic scene (known poses) with the oracle
(the reference's path restated on the CPU) and with flowmap_amd from identical initial
parameters and the same Adam schedule, then report ATE (flowmap/misc/ate.py) of both.

    python tests/tools/ate_check.py --device cuda:0 --frames 16 --height 256 --width 256 --steps 200
Prints one JSON line.  `--device cpu` runs flowmap_amd on the host test double (tests only).
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


def run(args):
    import flowmap_amd
    from flowmap_amd import Batch, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from helpers import to_flows, to_tracks

    f, h, w = args.frames, args.height, args.width
    dev = torch.device(args.device)
    if dev.type == "cpu":
        from helpers import build_host_sim

        _lib.set_library_for_testing(build_host_sim())
    sc = orc.synth_scene(f, h, w, seed=args.seed, focal=0.85, depth_noise=args.noise)
    gt_pos = sc["extrinsics_gt"][:, :3, 3]
    focal0 = 0.85 * 1.1  # start 10 % off
    otracks = orc.synth_tracks(f, h, w, scene=sc, seed=args.seed, interval=5, radius=min(20, f), grid=args.track_grid) if args.tracking else None

    # ---- oracle run (CPU) ----------------------------------------------------------------
    d = sc["depth_init"].clone().requires_grad_(True)
    wl = torch.zeros((f - 1, h, w), requires_grad=True)
    fo = torch.tensor(focal0, requires_grad=True)
    opt = torch.optim.Adam([d, wl, fo], lr=args.lr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.zero_grad(set_to_none=True)
        total, _, out = orc.explicit_depth_step(d, wl, fo, sc["flows"], (h, w), num_points=args.points, tracks=otracks)
        total.backward()
        opt.step()
    t_ref = time.perf_counter() - t0
    with torch.no_grad():
        _, _, out = orc.explicit_depth_step(d, wl, fo, sc["flows"], (h, w), num_points=args.points, tracks=otracks)
    ate_ref, loss_ref = orc.ate(gt_pos, out.extrinsics[0, :, :3, 3]), float(total.detach())

    # ---- flowmap_amd run -------------------------------------------------------------------
    flowmap_amd.set_lazy_surfaces(True)
    cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", focal0),
                   ExtrinsicsProcrustesCfg("procrustes", args.points, False))
    model = Model(cfg, num_frames=f, image_shape=(h, w))
    model.backbone.depth.data = sc["depth_init"].clone()
    model = model.to(dev)
    flows = to_flows(sc["flows"], dev)
    batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01))) if args.tracking else None
    tracks = to_tracks(otracks, dev)
    opt = flowmap_amd.FusedAdam(model.parameters(), lr=args.lr)  # the reference leg above uses torch.optim.Adam
    if args.in_pass:
        opt.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)  # the depth update applied by the flow-loss pass itself
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.zero_grad(set_to_none=True)
        out = model(batch, flows, 0)
        loss = loss_fn(batch, flows, None, out, 0)
        if track_fn is not None:
            loss = loss + track_fn(batch, flows, tracks, out, 0)
        loss.backward()
        opt.step()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t_ours = time.perf_counter() - t0
    with torch.no_grad():
        out = model(batch, flows, 0)
    ate_ours, loss_ours = orc.ate(gt_pos, out.extrinsics[0, :, :3, 3]), float(loss.detach())
    init = orc.ate(gt_pos, orc.model_forward(sc["depth_init"][None], torch.full((1, f - 1, h, w), 0.5),
                                             orc.focal_to_k(torch.tensor(focal0), (h, w)).expand(1, f, 3, 3), sc["flows"],
                                             orc.procrustes_indices((h, w), args.points)).extrinsics[0, :, :3, 3])
    return {
        "scene": f"synthetic consistent scene, {f} frames @ {h}x{w}, seed {args.seed}, depth noise {args.noise}, focal init +10%",
        "steps": args.steps, "lr": args.lr, "procrustes_points": args.points,
        "losses": "flow (1000) + tracking (100)" if args.tracking else "flow (1000)",
        "ate_initial": init, "ate_reference_path_cpu": ate_ref, "ate_flowmap_amd": ate_ours,
        "ate_abs_diff": abs(ate_ref - ate_ours),
        "final_loss_reference_path": loss_ref, "final_loss_flowmap_amd": loss_ours,
        "seconds_reference_path_cpu": t_ref, "seconds_flowmap_amd": t_ours, "device": str(dev),
        "optimizer": "reference path: torch.optim.Adam; flowmap_amd: flowmap_amd.FusedAdam"
                     + (f" with fuse_depth_update ({opt.counters['in_pass_updates']} of {args.steps} depth updates inside the flow pass, "
                        f"{opt.counters['sparse_updates']} element-list updates of the weight logits)" if args.in_pass else ""),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--points", type=int, default=1000)
    ap.add_argument("--noise", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tracking", action="store_true", help="add the tracking loss (tracks = oracle projections of the true surface)")
    ap.add_argument("--in-pass", action="store_true", help="FusedAdam.fuse_depth_update: depth update inside the fused flow pass")
    ap.add_argument("--track-grid", type=int, default=12)
    ap.add_argument("--threads", type=int, default=16, help="torch CPU threads for the reference-path leg")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    print(json.dumps(run(a)))
