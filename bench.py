#!/usr/bin/env python
"""Benchmark of the hot path: overfit iters/sec on synthetic F-frame H×W video.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = what ModelWrapperOverfit.training_step + backward do per optimisation
iteration (model_wrapper_overfit.py:51-62): explicit-depth backbone ->
regressed intrinsics -> unproject -> Procrustes extrinsics -> flow loss -> backward to
(depth, weight logits, focal length).  No optimiser step, exactly like the CPU baseline
in BASELINE.md.  Workload at N=1: BASELINE.json configs[1] (150 frames @ 720x1280, flow
loss only, Procrustes P=1000), inputs resident in HBM.  For N>1 every rank owns its own
150-frame shard of a longer video (weak scaling; frame pairs shard with a one-frame
halo), with one packed all-reduce of the shared-intrinsics gradient + loss and a halo
exchange of the boundary frame's depth gradient over RCCL.

Prints ONE JSON line (rank 0) carrying `roofline` for the fused flow kernel (HIP-event
timed on its launch stream inside the timed region) and `cpu_baseline` (the oracle — a
PyTorch-CPU port of the reference path — timed on a bounded sample of the same workload).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--points", type=int, default=1000,
                    help="Procrustes points (config/model/extrinsics/procrustes.yaml:3); 0 = all pixels (ablation_explicit_depth.yaml:11-12)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads for the CPU baseline (16 measured fastest on the 256-thread EPYC 9575F host: "
                    "8 -> 0.42, 16 -> 0.25, 32 -> 0.35, 64 -> 0.45, 256 -> 5.2 s/iter at 4 frames @720p)")
    ap.add_argument("--items-per-thread", type=int, default=0)
    ap.add_argument("--smooth-flows", action="store_true",
                    help="spatially smooth synthetic flows (low-res noise upsampled) instead of i.i.d. per-pixel noise")
    ap.add_argument("--intrinsics", choices=["regressed", "softmin"], default="regressed",
                    help="softmin = the reference's default first-1000-steps intrinsics (60-candidate sweep, 8192 points)")
    ap.add_argument("--optimizer", choices=["none", "fused", "torch"], default="none",
                    help="add the Adam step (lr 3e-5, config/overfit.yaml:30) to every iteration: flowmap_amd.FusedAdam or "
                         "torch.optim.Adam; the headline metric is fwd+bwd only (none)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in a hipGraph (flowmap_amd.GraphedStep) and replay it: for the launch-bound regime "
                         "(small frames); single GPU only")
    ap.add_argument("--tracking", action="store_true",
                    help="BASELINE.json configs[2]: add the tracking loss (segments every 5 frames, +-20 frames, 35x35 tracks)")
    return ap.parse_args()


def make_inputs(f, h, w, device, seed, smooth=False):
    """i.i.d. synthetic inputs of BASELINE.md §2, generated directly in HBM."""
    g = torch.Generator(device=device).manual_seed(seed)
    depth = 1.10 + 0.05 * torch.rand((f, h, w), device=device, generator=g)
    wlogit = 0.01 * torch.randn((f - 1, h, w), device=device, generator=g)
    from flowmap_amd import Flows

    def flow_field():
        if not smooth:
            return 0.01 * torch.randn((1, f - 1, h, w, 2), device=device, generator=g)
        low = 0.01 * torch.randn((f - 1, 2, max(h // 40, 2), max(w // 40, 2)), device=device, generator=g)
        up = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)
        return up.permute(0, 2, 3, 1)[None].contiguous()

    flows = Flows(
        flow_field(),
        flow_field(),
        torch.rand((1, f - 1, h, w), device=device, generator=g),
        torch.rand((1, f - 1, h, w), device=device, generator=g),
    )
    return depth, wlogit, flows


def make_tracks(f, device, seed, interval=5, radius=20, grid=35):
    """Synthetic track segments laid out as generate_video_tracks does
    (flowmap/tracking/__init__.py:49-70): one segment around every `interval`-th frame,
    +-radius frames, grid x grid query points drifting as a random walk; ~90 % visible."""
    from flowmap_amd import Tracks

    g = torch.Generator(device=device).manual_seed(seed)
    lin = (torch.arange(grid, device=device, dtype=torch.float32) + 0.5) / grid
    query = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), dim=-1).reshape(-1, 2)
    out = []
    for mid in range(0, f, interval):
        start, end = max(0, mid - radius), min(f, mid + radius + 1)
        drift = (0.003 * torch.randn((end - start, query.shape[0], 2), device=device, generator=g)).cumsum(0)
        # every segment tracks its own points (a tracker's query grid sits on the segment's middle
        # frame): jitter the grid inside its cells so that segments do not share pixels exactly
        jitter = (torch.rand((query.shape[0], 2), device=device, generator=g) - 0.5) / grid
        xy = (query + jitter)[None] + drift - drift[mid - start]
        vis = (xy >= 0).all(-1) & (xy < 1).all(-1) & (torch.rand(xy.shape[:2], device=device, generator=g) < 0.9)
        out.append(Tracks(xy[None].contiguous(), vis[None].contiguous(), start))
    return out


def cpu_baseline(frames, h, w, points, iters, threads):
    """The oracle (PyTorch CPU port of the reference path) on a bounded sample: same
    frame size, fewer frames; forward + backward, all host cores."""
    from oracle import flowmap_oracle as orc

    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    depth, wlogit, flows = orc.synth_iid(frames, h, w, seed=0)
    depth.requires_grad_(True)
    wlogit.requires_grad_(True)
    focal = torch.tensor(0.85, requires_grad=True)

    def step():
        for p in (depth, wlogit, focal):
            p.grad = None
        total, _, _ = orc.explicit_depth_step(depth, wlogit, focal, flows, (h, w), num_points=points)
        total.backward()

    step()  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return dt, cores


def _reserve_stdout():
    """The driver reads ONE JSON line from stdout, but RCCL prints a version banner to the C-level
    stdout (fd 1) when the communicator comes up.  Keep a private handle on the real stdout for the
    result line and point fd 1 at stderr for everything else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    args = parse()
    result_stream = _reserve_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("FLOWMAP_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import flowmap_amd
    from flowmap_amd import Batch, _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard

    f, h, w = args.frames, args.height, args.width
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, flows = make_inputs(f, h, w, device, seed=1 + rank, smooth=args.smooth_flows)
    if args.intrinsics == "softmin":
        from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg

        intrinsics_cfg = IntrinsicsSoftminCfg("softmin", 8192, 0.5, 2.0, 60, RegressionCfg(1000, 100))
    else:
        intrinsics_cfg = IntrinsicsRegressedCfg("regressed", 0.85)
    cfg = ModelCfg(
        BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
        intrinsics_cfg,
        ExtrinsicsProcrustesCfg("procrustes", args.points if args.points > 0 else None, False),
    )
    model = Model(cfg, num_frames=f, image_shape=(h, w)).to(device)
    model.backbone.depth.data = depth
    model.backbone.weights.data = wlogit
    batch = Batch(torch.zeros((1, f, 3, 1, 1), device=device).expand(1, f, 3, h, w))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    if args.items_per_thread:
        loss_fn.items_per_thread = args.items_per_thread
    tracks, track_fn = None, None
    if args.tracking:
        from flowmap_amd.loss import LossTracking, LossTrackingCfg

        sharded_tracks = dist is not None and world > 1
        # sharded: ONE track set over the whole (world x 149 + 1)-frame video, global frame indices,
        # identical on every rank; each rank evaluates the sources it owns (FrameShard.tracking_loss)
        tracks = make_tracks(world * (f - 1) + 1 if sharded_tracks else f, device, seed=100 if sharded_tracks else 100 + rank)
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world if dist is None else max(world, 1), dist)
    if dist is not None and world == 1:
        shard.world = 2  # single-rank RCCL self-test: run the collectives, there are no neighbours
        shard.exchange_halo = lambda grad: None
    shard.prepare_flow_loss(loss_fn, flows)  # global valid-sum (one-time all-reduce)

    optimizer = None
    if args.optimizer == "fused":
        optimizer = flowmap_amd.FusedAdam(model.parameters(), lr=3e-5, capturable=args.graph)
    elif args.optimizer == "torch":
        optimizer = torch.optim.Adam(model.parameters(), lr=3e-5)

    kernel_events = []
    _ops.flow_kernel_events = kernel_events  # (start, end) per fused-kernel launch

    def step():
        model.zero_grad(set_to_none=True)
        out = model(batch, flows, 0)
        loss = loss_fn(batch, flows, None, out, 0)
        tracked = None
        if track_fn is not None and dist is not None and world > 1:
            tracked = shard.tracking_loss(track_fn, tracks, out, world * (f - 1))  # global value, this rank's gradients
            (loss + tracked).backward()
        else:
            if track_fn is not None:
                loss = loss + track_fn(batch, flows, tracks, out, 0)
            loss.backward()
        shard.sync(loss, getattr(model.intrinsics, "focal_length", None), model.backbone.depth, already_global=tracked)
        if optimizer is not None:
            optimizer.step()
        return loss

    # one-time precompute, outside warm-up and timing whatever W is: the first step packs the constant
    # flows / masks and reduces the valid sums, the second one plans the static scatters (SURVEY §8d:
    # the metric excludes one-time precompute)
    for _ in range(2):
        step()
    if args.graph:
        if dist is not None or args.optimizer == "torch":
            raise SystemExit("--graph: single GPU, and --optimizer none|fused")
        eager_step = step
        graphed = flowmap_amd.GraphedStep(eager_step, warmup=3)  # disables the per-launch events: kernel_ms stays 0
        step = graphed  # noqa: F811
    if dist is not None:  # create the RCCL communicators / P2P channels outside the timed region
        shard.exchange_halo(torch.zeros((2, h, w), device=device))
        dist.all_reduce(torch.zeros(4, device=device))
    for _ in range(args.warmup):
        step()
    flowmap_amd.freeze_gc()  # a full cyclic-GC pass over torch's import-time objects costs ~50 ms (flowmap_amd/host.py)
    kernel_events.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = sum(s.elapsed_time(e) for s, e in kernel_events) / max(len(kernel_events), 1)
    traffic, traffic_src = None, None
    try:  # HBM bytes per launch from the PMC passes committed under profiles/ (same workload only)
        rec = json.loads((ROOT / "profiles" / "r01_flow_kernel_traffic.json").read_text())
        if rec["workload"] == {"frames": args.frames, "height": args.height, "width": args.width}:
            traffic, traffic_src = rec["hbm_bytes_per_launch"], "profiles/r01_flow_kernel_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)"
    except Exception:
        pass
    n = h * w
    algo_bytes = n * (8 * f + 24 * (f - 1))  # B_flow per launch
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": "overfit iters/sec (150 frames @ 720p) at 1/2/4/8 MI355X; final ATE vs ref",
            "value": world * args.steps / elapsed,
            "unit": "iters/sec (one iter = fwd+bwd over one 150-frame shard; aggregate over GPUs)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE.json configs[1]: {f} frames @ {h}x{w}, flow loss only (huber 0.01, weight 1000), "
                f"explicit-depth backbone, {args.intrinsics} intrinsics, Procrustes P={args.points}; fwd+bwd, "
                + ("no optimiser" if optimizer is None else f"+ Adam step ({type(optimizer).__module__}.{type(optimizer).__name__})")
                + ("; whole step replayed as one hipGraph" if args.graph else "")
                + (f"; + tracking loss (configs[2]): {len(tracks)} segments x {tracks[0].xy.shape[2]} tracks" if tracks else ""),
                "frames_per_gpu": f,
                "height": h,
                "width": w,
                "parallelism": f"frame-shard x{world}" if world > 1 else "single GPU",
                "loss": float(loss.item()),
            },
            "roofline": {
                "kernel": "fm::flow_fused_kernel<VEC=4, huber, GRAD, PACKED>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": kernel_ms,
                "launches_timed": len(kernel_events),
            },
        }
        if world == 1 and args.cpu_frames >= 2:
            dt, cores = cpu_baseline(args.cpu_frames, h, w, args.points, args.cpu_iters, args.cpu_threads)
            scaled = dt * (f - 1) / (args.cpu_frames - 1)  # per-pair cost is constant (optimistic for the CPU)
            result["cpu_baseline"] = {
                "value": 1.0 / scaled,
                "unit": "iters/sec",
                "cores": cores,
                "kind": "port",
                "host_logical_cpus": os.cpu_count(),
                "sample": f"oracle (PyTorch-CPU port of the reference path), {cores} torch threads, {args.cpu_frames} frames @ {h}x{w}, fwd+bwd, "
                f"{args.cpu_iters} timed iters after 1 warm-up: {dt:.3f} s/iter, scaled by pairs ({f - 1}/{args.cpu_frames - 1}) "
                f"to {f} frames",
                "sample_seconds_per_iter": dt,
            }
        print(json.dumps(result), file=result_stream, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
