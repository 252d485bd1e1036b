#!/usr/bin/env python
"""Benchmark of the hot path: overfit iters/sec on synthetic F-frame H×W video (BASELINE.json's metric).

    python bench.py                                   # N = 1: config c1, 20 warm-up + 100 timed steps
    python bench.py --config c2|c3|c4 [--scaling strong|weak] [--inputs scene|iid]
    python bench.py --gpus N --steps K --warmup W      # N > 1: launches its own N ranks (one per GPU, RCCL), equivalent to
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --one-gpu [--backend gloo]   # FUNCTIONAL N-rank run on a one-GPU box: the ranks share cuda:0 and meet over RCCL's socket
                                                           # transport (a host id per rank) or gloo; tests/test_gpu_multirank.py; timing meaningless

One "step" = what ModelWrapperOverfit.training_step + backward do per optimisation iteration
(model_wrapper_overfit.py:51-62): explicit-depth backbone -> intrinsics -> unproject ->
Procrustes extrinsics -> enabled losses -> backward to (depth, weight logits, focal length).  No optimiser
step unless --optimizer says so, exactly like the CPU baseline in BASELINE.md.

Configs (BASELINE.json configs[1..4]), all inputs resident in HBM:
    c1  150 frames @ 720x1280, flow loss, Procrustes P = 1000            (the headline; default)
    c2  c1 + tracking loss: 30 segments x 1225 tracks
    c3  65 frames @ 1080x1920, flow loss                                 (BASELINE: 4 GPUs x 16 pairs)
    c4  1200 frames @ 1080x1920 i.i.d. inputs over 8 GPUs = 150 frames per GPU (weak scaling by definition)
Inputs: `scene` = a geometrically consistent scene (static bumpy surface, smooth camera path, flows induced
by the true geometry; the default of c1-c3 as §8d specifies), `iid` = independent noise per pixel (c4).

Scaling for N > 1: `strong` (default for c1-c3: the metric is "150 frames @ 720p at 1/2/4/8 GPUs") shards
the ONE video by frame pairs over the ranks (flowmap_amd.sharding.shard_pairs: a one-frame halo, one packed
all-reduce of [loss, shared-parameter gradients], a neighbour exchange of the halo frame's dL/ddepth; with
tracking an all-gather of the poses and an all-reduce of their gradients); `weak` (c4) gives every rank its
own 150-frame shard.  `value` is whole-job throughput either way: iterations of the whole workload per second.

Strong scaling on ONE GPU (`--share K [--share-rank R]`): rank R's share of a K-GPU strong-scaling run — its pairs, its halo
frames, every collective executed on a one-rank RCCL communicator, the halo exchange stood in for by the local copies and
adds it causes (flowmap_amd.sharding.FrameShard(proxy=True)) — i.e. everything a rank does per step except the time its
bytes spend on xGMI.  `--graph` replays the step (collectives included) as one hipGraph.  tools/scaling_proxy.sh runs
K = 1, 2, 4, 8 eager and graphed -> profiles/r03_strong_scaling_proxy.jsonl; DESIGN.md §5 turns it into a projection.

The default step (`--model installed`) is built by a reference-LAYOUT `flowmap` package after flowmap_amd.install(): the real package when it
is importable, else bench_support/standin/flowmap (the GPU box has no reference).  The stand-in is the HOST APPLICATION's stand-in, not the product: after
install() every arithmetic name it resolves is this library's; its own arithmetic (the oracle's, for the CPU tests) is imported lazily and refuses
GPU tensors, and the line reports `via_install.oracle_imported_by_the_timed_path` (false) and the kernels of a step (all `fm::`).

Prints ONE JSON line (rank 0) with `roofline` for the fused flow kernel (HIP events on its launch stream
inside the timed region), `roofline_tracking` when the tracking loss runs (track_pairs: VALU-bound, GFLOP/s)
and, at N = 1, `cpu_baseline` (the oracle — a PyTorch-CPU port of the reference path — timed on a bounded
sample of the same workload).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

sys.dont_write_bytecode = True  # (`import flowmap` may find the real, read-only reference on the caller's path: never write bytecode next to it)
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))  # (the test tree is NOT on the path of the timed step: only the checker legs below — `ate`, `cpu_baseline` — add it / import oracle/)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_PEAK_GFLOPS = 157300.0  # same guide: fp32 vector peak with packed FMA
TRACK_VALU_PER_RESIDUAL = 60.3  # measured: SQ_INSTS_VALU (26 019 per wave x 1 984 waves) x 64 lanes / residuals at C2 (profiles/r05_c2_sq_counters_scaled_residual.csv; 67.2 before the scaled-residual loop)
TRACK_FLOPS_PER_RESIDUAL = 90.0  # track_pair_term (csrc/fm_pose.h): ~60 VALU instructions, FMAs counted twice

CONFIGS = {
    "c1": dict(frames=150, height=720, width=1280, tracking=False, inputs="scene", scaling="strong", ref="BASELINE.json configs[1]"),
    "c2": dict(frames=150, height=720, width=1280, tracking=True, inputs="scene", scaling="strong", ref="BASELINE.json configs[2]"),
    "c3": dict(frames=65, height=1080, width=1920, tracking=False, inputs="scene", scaling="strong", ref="BASELINE.json configs[3]"),
    "c4": dict(frames=150, height=1080, width=1920, tracking=False, inputs="iid", scaling="weak",
               ref="BASELINE.json configs[4] (1200 frames over 8 GPUs: 150 frames per GPU)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c1")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None, help="default: the config's (strong for c1-c3, weak for c4)")
    ap.add_argument("--inputs", choices=["scene", "iid"], default=None, help="default: the config's (scene for c1-c3, iid for c4)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--tracking", action="store_true", help="add the tracking loss to any config (c2 = c1 --tracking)")
    ap.add_argument("--points", type=int, default=1000,
                    help="Procrustes points (config/model/extrinsics/procrustes.yaml:3); 0 = all pixels (ablation_explicit_depth.yaml:11-12)")
    ap.add_argument("--cpu-frames", type=int, default=-1,
                    help="frames of the CPU-baseline leg, taken from the front of the SAME inputs the GPU leg runs on (0 = skip; default: the whole "
                         "video when the host has >= 96 GB of free memory — one iteration of 150 x 720p takes the oracle ~40 GB and ~30 s — "
                         "else a 32-frame sample, labelled as such)")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed iterations of the CPU-baseline leg after its warm-up (BASELINE.md §3: 1 warm-up + 3)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads for the CPU baseline (16 measured fastest on the 256-thread EPYC 9575F host: "
                    "8 -> 0.42, 16 -> 0.25, 32 -> 0.35, 64 -> 0.45, 256 -> 5.2 s/iter at 4 frames @720p)")
    ap.add_argument("--items-per-thread", type=int, default=0)
    ap.add_argument("--intrinsics", choices=["regressed", "softmin"], default="regressed",
                    help="softmin = the reference's default first-1000-steps intrinsics (60-candidate sweep, 8192 points)")
    ap.add_argument("--optimizer", choices=["none", "fused", "in_pass", "torch"], default="none",
                    help="add the Adam step (lr 3e-5, config/overfit.yaml:30) to every iteration: flowmap_amd.FusedAdam or "
                         "torch.optim.Adam; the headline metric is fwd+bwd only (none)")
    ap.add_argument("--graph", nargs="?", const="whole", default=None, choices=["whole", "compute", "off"],
                    help="replay the step as a hipGraph: `whole` (flowmap_amd.GraphedStep: everything incl. the sharded step's RCCL collectives; "
                         "what a bare --graph means), `compute` (flowmap_amd.GraphedShardedStep: forward + backward in the graph, collectives issued "
                         "eagerly after the replay), `off`.  Default: off on one GPU (the headline number is measured eagerly, with HIP events around the "
                         "flow kernel), `compute` for a multi-rank strong-scaling run of the flow loss — a rank's ~15 kernels take ~0.2 ms at 8 GPUs and "
                         "cannot hide ~0.45 ms of eager enqueueing")
    ap.add_argument("--halo", choices=["auto", "oneshot", "early", "ghost"], default="auto",
                    help="strong scaling: how the boundary frames' dL/ddepth reaches the neighbour — `oneshot`: one 3.7 MB (720p) exchange per boundary and "
                         "direction after backward; `early` (FrameShard.enable_early_halo): the dense part right after the flow pass, under the rest of the "
                         "step, and a sparse correction (~20 KB) after backward (with --graph compute: between the forward and the backward replay).  "
                         "`ghost` (FrameShard.enable_ghost_halo): no frame travels at all — each rank is handed the neighbouring pair's constant flow once, receives its 64-byte pose "
                         "per step and evaluates the neighbour's dense part itself (fm_flow_ghost_terms), the same sparse correction after backward.  "
                         "auto: ghost for a multi-rank strong-scaling run, oneshot otherwise (the --share proxy states its mode explicitly)")
    ap.add_argument("--share", type=int, default=0,
                    help="K > 0: run ONE rank's share of a K-GPU strong-scaling run on this GPU (its pairs + halo frames, collectives on a "
                         "one-rank RCCL communicator): the per-rank step time of a K-GPU run without the wire time")
    ap.add_argument("--share-rank", type=int, default=-1, help="which rank's share (default: an interior rank with the largest share)")
    ap.add_argument("--count-launches", action="store_true", help="count the kernel launches of one step with torch.profiler (after the timed region)")
    ap.add_argument("--sustained-steps", type=int, default=100,
                    help="N = 1: after the timed K steps, N more steps timed separately and reported as `sustained` (informational; `value` stays the K steps "
                         "after W warm-up steps).  The first ~30 launches after an idle period run on a power-management transient (profiles/r04_driver_command_ramp.txt): "
                         "a 20-step run measures that transient, a 2000-step overfit run the sustained state.  0 = skip")
    ap.add_argument("--whole", action="store_true",
                    help="config c4 WHOLE on one GPU: all 1200 frames @ 1080x1920 (79.6 GB of inputs + a 59.7 GB packed copy of the 288 GB) instead of "
                         "one GPU's 150-frame shard; implies --release-originals")
    ap.add_argument("--release-originals", action="store_true",
                    help="flowmap_amd.release_flow_originals(flows) once the flows are packed: the forward flow and both masks (half of the inputs) are "
                         "given back; the CPU-baseline sample is copied to the host first")
    ap.add_argument("--no-tap-exchange", action="store_true",
                    help="flow + tracking: run the two losses as in round 3 (the tracking loss after the flow pass, sampling the depth images and "
                         "read-modify-writing dL/ddepth at its taps) instead of the tap exchange (flowmap_amd/_ops.py: TapPlan)")
    ap.add_argument("--ate", choices=["auto", "on", "off"], default="auto",
                    help="the metric's second half, `final ATE vs ref`, MEASURED BY THIS RUN: after the timed region, the optimisation the imported reference "
                         "ran once on the build container's CPU (oracle/make_ate_reference.py -> tests/golden/ate_*_imported_reference.json: 150 frames, flow + "
                         "tracking, softmin -> regressed intrinsics, Adam) is run here by flowmap_amd from the same initial parameters on the same seeded scene, and "
                         "both ATEs go into the line's `ate` block.  auto: at N = 1 on the headline config when a fixture is present (seconds on the GPU)")
    ap.add_argument("--ate-fixture", default=None, help="the reference leg's record (default: the 720p one under tests/golden/, else the 360p one)")
    ap.add_argument("--model", choices=["installed", "direct"], default="installed",
                    help="how the step's modules are built.  `installed` (default): flowmap_amd.install() patches a reference-LAYOUT `flowmap` package "
                         "(the real dcharatan/flowmap when it is importable, else bench_support/standin — the GPU box has no /root/reference) and the step is that "
                         "package's own Model(get_backbone, get_intrinsics, get_extrinsics) + get_losses, i.e. what an unmodified overfit.py runs "
                         "(model_wrapper_overfit.py:51-62).  `direct`: flowmap_amd.model.model.Model and the loss classes constructed by hand")
    ap.add_argument("--training-step", choices=["off", "eager", "graph"], default="off",
                    help="drive the step through the reference-layout package's ModelWrapperOverfit.training_step (model_wrapper_overfit.py:51-73) in a "
                         "trainer's order — training_step, zero_grad, backward, optimiser, global_step + 1: `eager` = the package's own method after "
                         "install(); `graph` = install(graph=True): forward + losses and backward replayed as two hipGraphs (flowmap_amd/training.py)")
    ap.add_argument("--default-resolution", choices=["auto", "on", "off"], default="auto",
                    help="the reference's default operating point (config/overfit.yaml:33-38: 150 frames of ~180x240, flow + tracking) beside the headline: "
                         "two child runs of this file — the package's ModelWrapperOverfit.training_step eager and under install(graph=True) — reported "
                         "under `default_resolution` (auto: with the default headline run on one GPU, like the ATE leg)")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto",
                    help="the process group's backend: auto = nccl (RCCL) on GPUs, gloo on the CPU dry run.  `gloo` on GPUs (it moves GPU tensors through the "
                         "host) is an alternative for --one-gpu")
    ap.add_argument("--one-gpu", action="store_true",
                    help="every rank computes on cuda:0: the FUNCTIONAL multi-rank run on a one-GPU box (tests/test_gpu_multirank.py; timing is meaningless).  "
                         "Over RCCL (the default backend) every rank declares a host of its own (NCCL_HOSTID) and the ranks meet over RCCL's socket transport on "
                         "the loopback interface — RCCL refuses two ranks of one host on one device; over --backend gloo the tensors are staged through the host")
    ap.add_argument("--torch-baseline", type=int, default=0, metavar="STEPS",
                    help="after the timed region: the reference's op sequence on stock PyTorch-ROCm on this GPU (tests/tools/torch_gpu_reference_ops.py "
                         "in a process of its own, 1 warm-up + STEPS steps on i.i.d. inputs of the workload's size) as `rocm_torch_baseline`")
    return ap.parse_args()


# --------------------------------------------------------------------------------------
# Synthetic inputs, generated directly in HBM
# --------------------------------------------------------------------------------------


def make_iid(f, h, w, device, seed):
    """i.i.d. inputs of BASELINE.md §2: depth U(1.10,1.15), flows N(0,0.01²), masks U(0,1), weight logits N(0,0.01²)."""
    from flowmap_amd import Flows

    g = torch.Generator(device=device).manual_seed(seed)
    depth = 1.10 + 0.05 * torch.rand((f, h, w), device=device, generator=g)
    wlogit = 0.01 * torch.randn((f - 1, h, w), device=device, generator=g)
    flows = Flows(
        0.01 * torch.randn((1, f - 1, h, w, 2), device=device, generator=g),
        0.01 * torch.randn((1, f - 1, h, w, 2), device=device, generator=g),
        torch.rand((1, f - 1, h, w), device=device, generator=g),
        torch.rand((1, f - 1, h, w), device=device, generator=g),
    )
    return depth, wlogit, flows, None


def make_scene(f, h, w, device, seed, focal=0.85, depth_noise=0.05):
    """A consistent scene (§8d): a static bumpy surface z = height(x, y) seen by a smoothly moving camera.  Depth per
    frame by fixed-point ray casting; flows = where the true geometry sends every pixel (computed with this
    package's own compute_forward_flow / compute_backward_flow: the inputs of a TIMING run, not a parity
    check); masks = 1 where the target stays in frame; initial depth = truth x (1 + 5 % smooth noise)."""
    from flowmap_amd import Flows
    from flowmap_amd.model import projection as fm

    gen = torch.Generator().manual_seed(seed)
    t_axis = torch.linspace(0, 1, f, dtype=torch.float64)
    ph = torch.rand(6, generator=gen, dtype=torch.float64) * 2 * math.pi
    trans = torch.stack([0.25 * torch.sin(2 * math.pi * t_axis + ph[0]), 0.10 * torch.sin(4 * math.pi * t_axis + ph[1]), 0.15 * t_axis], -1)
    ang = torch.stack([0.04 * torch.sin(2 * math.pi * t_axis + ph[2]), 0.06 * torch.sin(2 * math.pi * t_axis + ph[3]),
                       0.02 * torch.sin(2 * math.pi * t_axis + ph[4])], -1)
    cx, cy, cz, sx, sy, sz = *torch.cos(ang).unbind(-1), *torch.sin(ang).unbind(-1)
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], -1).reshape(f, 3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).reshape(f, 3, 3)
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], -1).reshape(f, 3, 3)
    ext = torch.eye(4, dtype=torch.float64).repeat(f, 1, 1)
    ext[:, :3, :3] = rz @ ry @ rx
    ext[:, :3, 3] = trans
    ext = (torch.linalg.inv(ext[0])[None] @ ext).float().to(device)  # first pose = identity, like get_extrinsics
    s = (h * w) ** 0.5
    k = torch.tensor([[focal * s / w, 0, 0.5], [0, focal * s / h, 0.5], [0, 0, 1.0]], device=device)
    xy, _ = fm.sample_image_grid((h, w), device)
    rays = torch.cat([xy, torch.ones_like(xy[..., :1])], -1) @ torch.linalg.inv(k).T  # (h, w, 3), z component 1

    def height(xw, yw):
        return 2.0 + 0.25 * torch.sin(1.7 * xw + 0.3) * torch.cos(1.3 * yw - 0.2) + 0.1 * torch.sin(3.1 * xw * yw)

    depth = torch.empty((f, h, w), device=device)
    group = max(1, min(f, (1 << 26) // (h * w)))  # frames per batch of the ray casting: ~0.8 GB of temporaries, 10x fewer launches
    for i in range(0, f, group):
        r, c = ext[i : i + group, :3, :3], ext[i : i + group, :3, 3]  # (g,3,3), (g,3)
        d = torch.full((r.shape[0], h, w), 2.0, device=device)
        step = r[:, 2, 2].clamp_min(0.5)[:, None, None]
        for _ in range(40):
            pw = torch.einsum("ghwk,gjk->ghwj", rays[None] * d[..., None], r) + c[:, None, None]
            d = d + (height(pw[..., 0], pw[..., 1]) - pw[..., 2]) / step
        depth[i : i + group] = d
    with torch.no_grad():
        kk = k.expand(1, f, 3, 3).contiguous()
        fwd = torch.empty((1, f - 1, h, w, 2), device=device)
        bwd = torch.empty_like(fwd)
        for a in range(0, f - 1, 16):  # in chunks: the explicit (frames, h, w, 3) surfaces are 11 MB per frame at 720p
            b = min(a + 16, f - 1)
            surf = fm.unproject_dense(xy, depth[None, a : b + 1], kk[:, a : b + 1, None, None])
            fwd[:, a:b] = fm.compute_forward_flow(surf, ext[None, a : b + 1], kk[:, a : b + 1]) - xy
            bwd[:, a:b] = fm.compute_backward_flow(surf, ext[None, a : b + 1], kk[:, a : b + 1]) - xy

    def inside(flow):
        pos = flow + xy
        return ((pos >= 0).all(-1) & (pos < 1).all(-1)).float()

    flows = Flows(fwd, bwd, inside(fwd), inside(bwd))
    smooth = torch.nn.functional.interpolate(torch.randn((1, f, max(h // 16, 2), max(w // 16, 2)), generator=gen).to(device), size=(h, w),
                                             mode="bilinear", align_corners=False)[0]
    wlogit = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator(device=device).manual_seed(seed), device=device)
    return depth * (1 + depth_noise * smooth), wlogit, flows, {"depth": depth, "extrinsics": ext, "intrinsics": k}


def make_tracks(f, device, seed, scene=None, hw=None, interval=5, radius=20, grid=35):
    """Track segments laid out as generate_video_tracks does (flowmap/tracking/__init__.py:49-70): one segment
    around every `interval`-th frame, +-radius frames, grid x grid query points on the middle frame; ~90 % visible.
    With a scene the tracks follow the true surface points (projected with this package's kernels), otherwise
    they drift as a random walk."""
    from flowmap_amd import Tracks
    from flowmap_amd.model import projection as fm

    g = torch.Generator(device=device).manual_seed(seed)
    lin = (torch.arange(grid, device=device, dtype=torch.float32) + 0.5) / grid
    query = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), dim=-1).reshape(-1, 2)
    out = []
    for mid in range(0, f, interval):
        start, end = max(0, mid - radius), min(f, mid + radius + 1)
        # every segment tracks its own points (a tracker's query grid sits on the segment's middle frame):
        # jitter the grid inside its cells so that segments do not share pixels exactly
        q = query + (torch.rand((query.shape[0], 2), device=device, generator=g) - 0.5) / grid
        if scene is not None:
            h, w = hw
            with torch.no_grad():
                xy, _ = fm.sample_image_grid((h, w), device)
                kk = scene["intrinsics"].expand(1, 1, 3, 3).contiguous()
                surf = fm.unproject_dense(xy, scene["depth"][None, mid : mid + 1], kk[:, :, None, None])  # (1,1,h,w,3)
                pts = torch.nn.functional.grid_sample(surf[0].permute(0, 3, 1, 2), (q * 2 - 1)[None, :, None], mode="bilinear",
                                                      padding_mode="border", align_corners=False)[0, :, :, 0].T  # (P,3)
                rel = torch.linalg.inv(scene["extrinsics"][start:end]) @ scene["extrinsics"][mid]  # (n,4,4)
                cam = pts @ rel[:, :3, :3].transpose(1, 2) + rel[:, None, :3, 3]
                proj = cam / (cam[..., 2:] + 1e-5)
                xy_t = (proj @ scene["intrinsics"].T)[..., :2]
        else:
            drift = (0.003 * torch.randn((end - start, q.shape[0], 2), device=device, generator=g)).cumsum(0)
            xy_t = q[None] + drift - drift[mid - start]
        vis = (xy_t >= 0).all(-1) & (xy_t < 1).all(-1) & (torch.rand(xy_t.shape[:2], device=device, generator=g) < 0.9)
        out.append(Tracks(xy_t[None].contiguous(), vis[None].contiguous(), start))
    return out


def cpu_baseline(depth, wlogit, flows, focal, frames, h, w, points, iters, threads):
    """The oracle (PyTorch CPU port of the reference path) on the first `frames` frames of the SAME inputs the GPU leg
    runs on (copied to the host once): forward + backward on `threads` host cores."""
    from oracle import flowmap_oracle as orc

    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    depth = depth[:frames].detach().cpu().clone().requires_grad_(True)
    wlogit = wlogit[: frames - 1].detach().cpu().clone().requires_grad_(True)
    oflows = orc.OFlows(*(x[:, : frames - 1].detach().cpu().contiguous() for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
    focal = torch.tensor(float(focal), requires_grad=True)

    def step():
        for p in (depth, wlogit, focal):
            p.grad = None
        total, _, _ = orc.explicit_depth_step(depth, wlogit, focal, oflows, (h, w), num_points=points)
        total.backward()
        return float(total.detach())

    t0 = time.perf_counter()
    loss = step()  # warm-up (allocator, page faults: ~40 GB of autograd temporaries at 150 x 720p)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    return (time.perf_counter() - t0) / iters, cores, first, loss


def torch_baseline(frames, h, w, points, steps, timeout=600):
    """SURVEY.md §8d's third column: the reference's op sequence executed by stock PyTorch-ROCm on this GPU, in a process of its own
    (its ~40 GB of autograd temporaries and a possible device fault stay out of this one) — a measured ratio, not a target."""
    import subprocess

    cmd = [sys.executable, str(ROOT / "tests" / "tools" / "torch_gpu_reference_ops.py"), "--frames", str(frames), "--height", str(h), "--width", str(w),
           "--points", str(points), "--iters", str(steps)]
    if frames * h * w > 24 * 720 * 1280:  # the whole-video call faults from 32 frames @ 720p on (profiles/r04_stock_pytorch_rocm_fault_*): 16-frame windows
        cmd += ["--chunk", str(max(2, (16 * 720 * 1280) // (h * w)))]
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        if run.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"failed": True, "returncode": run.returncode, "stderr_tail": run.stderr[-600:]}
    except subprocess.TimeoutExpired:
        return {"failed": True, "timeout_s": timeout}


def reference_layout_package():
    """The `flowmap` package flowmap_amd.install() patches: whatever `import flowmap` finds (the real dcharatan/flowmap on the caller's path),
    else the stand-in of the reference's module LAYOUT under bench_support/standin (registries, factories, import-site bindings; the GPU box has no
    reference).  Returns (kind, where)."""
    try:
        import flowmap
    except ImportError:
        sys.path.insert(0, str(ROOT / "bench_support" / "standin"))
        import flowmap
    where = str(Path(flowmap.__file__).resolve().parent)
    return ("stand-in of the reference's layout (bench_support/standin/flowmap)" if where.startswith(str(ROOT)) else "dcharatan/flowmap"), where


def installed_modules(model_cfg_parts, num_frames, image_shape, with_tracking):
    """After flowmap_amd.install(): the reference-layout package's OWN Model and loss factory (model/model.py:41-55, loss/__init__.py:16-17) —
    every part resolved through the registries install() rebound (BACKBONES, INTRINSICS, EXTRINSICS, LOSSES, MAPPINGS)."""
    from flowmap.loss import get_losses
    from flowmap.loss.loss_flow import LossFlowCfg
    from flowmap.loss.loss_tracking import LossTrackingCfg
    from flowmap.loss.mapping import MappingHuberCfg
    from flowmap.model.backbone import BackboneExplicitDepthCfg
    from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap.model.intrinsics import IntrinsicsRegressedCfg, IntrinsicsSoftminCfg
    from flowmap.model.model import Model, ModelCfg

    backbone, intrinsics, extrinsics = model_cfg_parts
    if intrinsics[0] == "softmin":
        try:
            from flowmap.model.intrinsics.intrinsics_softmin import RegressionCfg
        except ImportError:
            from flowmap.model.intrinsics import RegressionCfg
        intrinsics_cfg = IntrinsicsSoftminCfg(*intrinsics[:-1], RegressionCfg(*intrinsics[-1]))
    else:
        intrinsics_cfg = IntrinsicsRegressedCfg(*intrinsics)
    model = Model(ModelCfg(BackboneExplicitDepthCfg(*backbone), intrinsics_cfg, ExtrinsicsProcrustesCfg(*extrinsics), True),
                  num_frames=num_frames, image_shape=image_shape)
    cfgs = [LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))]
    if with_tracking:
        cfgs.append(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    return model, get_losses(cfgs)


def ate_leg(device, fixture):
    """`final ATE vs ref`, by this run: flowmap_amd optimises the fixture's scene from the fixture's initial parameters on `device` and its camera
    positions are scored with the reference's ATE (misc/ate.py:7-25) next to the imported reference's own result.  A CHECKER leg like
    `cpu_baseline`: the scene generator and the ATE restatement come from tests/tools + oracle/ and nothing here is timed as the product."""
    import types

    sys.path[:0] = [str(ROOT / "tests" / "tools"), str(ROOT / "tests")]
    import ate_full_chain as chain

    ref = json.loads(Path(fixture).read_text())
    base = dict(reference=str(fixture), device=str(device), in_pass=False, **ref["config"])
    t0 = time.perf_counter()
    first = chain.ours_leg(types.SimpleNamespace(**base), quiet=True, scene_device=str(device))
    built = first.pop("_built")
    scene_and_run_s = time.perf_counter() - t0
    # the schedule's own sensitivity, seen through this implementation: the same run from initial depths moved by 1e-7 (relative, Gaussian)
    again = chain.ours_leg(types.SimpleNamespace(**base), quiet=True, perturb=1e-7, built=built)
    again.pop("_built")
    rel = abs(first["ate_flowmap_amd"] - first["ate_reference_path_cpu"]) / first["ate_reference_path_cpu"]
    out = {
        "measured_by_this_run": True,
        "fixture": str(Path(fixture).relative_to(ROOT)) if str(fixture).startswith(str(ROOT)) else str(fixture),
        "reference": ref.get("reference_kind", "the reference path as restated by oracle/flowmap_oracle.py (tests/tools/ate_full_chain.py --leg reference)")
                     + ", on the build container's CPU: " + ref.get("made_by", ""),
        "scene": first["scene"], "schedule": first["schedule"],
        "ate_reference": first["ate_reference_path_cpu"], "ate_flowmap_amd": first["ate_flowmap_amd"], "ate_abs_diff": first["ate_abs_diff"], "ate_rel_diff": rel,
        "final_loss_reference": first["final_loss_reference_path"], "final_loss_flowmap_amd": first["final_loss_flowmap_amd"],
        "focal_final_reference": first["focal_final_reference_path"], "focal_final_flowmap_amd": first["focal_final_flowmap_amd"],
        "loss_trace_max_rel_diff": first["loss_trace_max_rel_diff"], "max_position_diff": first["max_position_diff"], "position_scale": first["position_scale"],
        "seconds_reference_cpu": first["seconds_reference_path_cpu"], "seconds_flowmap_amd": first["seconds_flowmap_amd"],
        "seconds_scene_synthesis_and_run": scene_and_run_s,
        "self_sensitivity": {"what": "flowmap_amd against itself from initial depths perturbed by 1e-7 (relative): what rounding-level differences become under this schedule",
                             "ate_perturbed": again["ate_flowmap_amd"],
                             "ate_rel_diff": abs(again["ate_flowmap_amd"] - first["ate_flowmap_amd"]) / first["ate_flowmap_amd"]},
    }
    sens = ref.get("self_sensitivity")  # (written into the fixture by oracle/make_ate_reference.py --perturb: the reference against itself)
    if sens:
        out["self_sensitivity"]["reference_ate_rel_diff"] = sens.get("ate_rel_diff")
        out["self_sensitivity"]["reference_made_by"] = sens.get("made_by")
    return out


def default_resolution_leg(timeout=90):  # (a child takes ~10 s; a stuck one must not cost the headline line its place in the driver's time limit)
    """config/overfit.yaml:33-38 — the resolution an unmodified `overfit.py` runs at — measured by THIS run in two child processes of this file:
    150 frames of 180x240, flow + tracking, stepped through the reference-layout package's ModelWrapperOverfit.training_step in a trainer's order
    of calls, eager after install() and replayed as hipGraphs after install(graph=True) (flowmap_amd/training.py).  Informational; a failure here
    leaves the headline line alone."""
    import subprocess

    out = {"what": "150 frames @ 180x240 (config/overfit.yaml:33-38), flow + tracking, fwd+bwd through the package's ModelWrapperOverfit.training_step in a "
                   "trainer's order of calls; measured by child runs of this file (bench.py --height 180 --width 240 --tracking --training-step eager|graph)"}
    for mode in ("eager", "graph"):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--height", "180", "--width", "240", "--tracking", "--training-step", mode, "--steps", "100",
               "--warmup", "20", "--cpu-frames", "0", "--ate", "off", "--sustained-steps", "0", "--default-resolution", "off"]
        env = dict(os.environ, FLOWMAP_BENCH_NO_PROFILER="1")
        try:
            done = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
            line = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
            if done.returncode != 0 or not line:
                out[mode] = {"failed": (done.stderr or "")[-300:]}
                continue
            rec = json.loads(line[-1])
            step = (rec.get("via_install") or {}).get("training_step") or {}
            out[mode] = {"ms_per_step": rec["ms_per_step"], "iters_per_sec": rec["value"], "steps": rec["steps"], "loss": rec["config"]["loss"],
                         **({k: step.get(k) for k in ("captures", "replays", "disabled")} if mode == "graph" else {})}
        except Exception as exc:  # noqa: BLE001
            out[mode] = {"failed": repr(exc)[:300]}
    if all("ms_per_step" in out.get(m, {}) for m in ("eager", "graph")):
        out["graph_over_eager"] = out["eager"]["ms_per_step"] / out["graph"]["ms_per_step"]
    return out


def default_ate_fixture():
    for name in ("ate_150x720x1280_200_steps_imported_reference.json", "ate_150x720x1280_imported_reference.json", "ate_150x360x640_imported_reference.json"):
        if (ROOT / "tests" / "golden" / name).exists():
            return ROOT / "tests" / "golden" / name
    return None


def c0_ate_fixture():
    """BASELINE.json configs[0] (16 frames @ 256x256) on the reference's REAL schedule (config/overfit.yaml:24-31: 2000 Adam steps at lr 3e-5;
    config/model/intrinsics/softmin.yaml:13-14: hand-over after step 1000, window 100; config/loss/tracking.yaml:4-6: tracking from step 50;
    config/tracking/cotracker.yaml:3: 35 x 35 tracks), run once by the imported reference (oracle/make_ate_reference.py, 17 min on 4 host threads)."""
    path = ROOT / "tests" / "golden" / "ate_c0_16x256x256_full_schedule_imported_reference.json"
    return path if path.exists() else None


def _cpu_frames(args, f_video, h, w):
    """Frames of the CPU-baseline leg: the whole video when it is no bigger than 1.25 x the headline workload and the host has the memory
    (one iteration of 150 x 720p takes the oracle ~40 GB and ~25 s), else a 32-frame sample from the front of the same inputs."""
    if args.cpu_frames >= 0:
        return min(args.cpu_frames, f_video)
    try:
        avail_gb = int(next(line for line in open("/proc/meminfo") if line.startswith("MemAvailable")).split()[1]) / 2**20
    except Exception:  # noqa: BLE001
        avail_gb = 0.0
    small = f_video * h * w <= 1.25 * 150 * 720 * 1280
    return f_video if (avail_gb >= 96 and small) else min(32, f_video)


def count_launches(step, device):
    """Kernel launches of one step, counted by torch.profiler's device activity (None when the profiler is unavailable)."""
    try:
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step()
            torch.cuda.synchronize(device)
        names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()
                 and not e.name.lower().startswith(("memcpy", "memset"))]
        return len(names), sorted(set(names))
    except Exception as exc:  # noqa: BLE001
        return None, [f"torch.profiler failed: {exc}"]


def _reserve_stdout():
    """The driver reads ONE JSON line from stdout, but RCCL prints a version banner to the C-level
    stdout (fd 1) when the communicator comes up.  Keep a private handle on the real stdout for the
    result line and point fd 1 at stderr for everything else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` — one process per GPU over RCCL, rank 0 prints the ONE JSON
    line — like the reference, which goes multi-GPU without a launcher (overfit.py:94-108: Lightning spawns when device_count() > 1).  The
    process image is replaced (exec): exit code, signals and stdout are the launcher's.  FLOWMAP_BENCH_LAUNCHER names the script the ranks
    run instead of this file (tests/test_bench_dryrun.py: the launcher that injects the host test double for the gloo dry run)."""
    on_gpu = os.environ.get("FLOWMAP_BENCH_DEVICE", "cuda") == "cuda"
    if on_gpu and not args.one_gpu:
        visible = torch.cuda.device_count()
        if visible < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible to this process; refusing to report an {args.gpus}-GPU number from fewer ranks")
    script = os.environ.get("FLOWMAP_BENCH_LAUNCHER") or str(Path(__file__).resolve())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *sys.argv[1:]]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver supports dmabuf IPC only: RCCL across processes needs it)
    env.setdefault("OMP_NUM_THREADS", "8")  # (torch.distributed.run would set 1 and warn; the ranks' host work is launch glue)
    print(f"[bench] --gpus {args.gpus}: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.share:
            raise SystemExit("--share K runs one rank's share on ONE process: --gpus 1")
        launch_ranks(args)  # (does not return)
    result_stream = _reserve_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # the line's n_gpus is the process group's size; a launcher that started a different number of ranks than --gpus asks for would put a
        # number for the wrong N into a scaling record
        raise SystemExit(f"bench.py --gpus {args.gpus} under WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # tests/test_bench_dryrun.py runs this file's multi-rank glue over gloo with CPU tensors and the host test double
    # injected by its launcher; without that launcher CPU tensors raise in the first operator (there is no CPU path)
    on_gpu = os.environ.get("FLOWMAP_BENCH_DEVICE", "cuda") == "cuda"
    if args.share and world > 1:
        raise SystemExit("--share K runs one rank's share on ONE process")
    backend = ("nccl" if on_gpu else "gloo") if args.backend == "auto" else args.backend
    if args.one_gpu:
        if not on_gpu:
            raise SystemExit("--one-gpu: GPU ranks")
        local_rank = 0
        if backend == "nccl":
            # RCCL refuses two ranks of one HOST on one device ("Duplicate GPU detected"): every rank declares a host of its own and the ranks meet over
            # RCCL's socket transport on the loopback interface (tools/probes/rccl_one_gpu_probe.py) — RCCL's own point-to-point and collective code,
            # proxy threads and stream semantics, everything but xGMI
            os.environ.update(NCCL_HOSTID=f"flowmap-amd-one-gpu-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                              NCCL_SHM_DISABLE="1")
            if args.graph == "whole" and world > 1:
                # (measured, round 6: hipStreamEndCapture segfaults on a capture that holds RCCL calls over the socket transport — its proxy steps are
                # host-function nodes; over xGMI peer-to-peer, which needs no proxy, this has not run)
                raise SystemExit("--one-gpu with --graph whole: RCCL's socket transport cannot be captured into a hipGraph here; use --graph compute (the default) or off")
    if world > 1 or args.share > 1:  # --share: a one-rank process group, so that every collective of the sharded step executes
        import torch.distributed as dist

        for key, value in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(key, value)
        import datetime

        if on_gpu and backend == "nccl":
            torch.cuda.set_device(local_rank)
            # (a collective that never completes — this path has not run with a peer yet — fails after five minutes instead of RCCL's default ten)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=5))
        if not args.share and dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: the process group has {dist.get_world_size()} rank(s)")
    else:
        dist = None
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)
    sync_device = torch.cuda.synchronize if on_gpu else (lambda: None)

    import flowmap_amd
    from flowmap_amd import Batch, Flows, _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard, shard_frames, shard_pairs

    cfg = dict(CONFIGS[args.config])
    for key in ("frames", "height", "width", "scaling", "inputs"):
        if getattr(args, key) is not None:
            cfg[key] = getattr(args, key)
    cfg["tracking"] = cfg["tracking"] or args.tracking
    if args.whole:
        if args.config != "c4" or world > 1 or args.share:
            raise SystemExit("--whole: config c4 on one GPU")
        cfg["frames"] = 1200 if args.frames is None else args.frames
        cfg["ref"] = "BASELINE.json configs[4] WHOLE on one GPU (1200 frames; the config shards them over 8)"
        args.release_originals = True
    f_video, h, w = cfg["frames"], cfg["height"], cfg["width"]
    # the (rank, world) the VIDEO is cut for: the process group's, or — `--share K` — rank R of K on this one GPU
    cut_world, cut_rank = (args.share, args.share_rank) if args.share > 1 else (world, rank)
    if args.share > 1 and cut_rank < 0:
        cut_rank = 1 if args.share > 2 else 0  # an interior rank (two neighbours) with the largest share
    if args.share > 1 and not 0 <= cut_rank < args.share:
        raise SystemExit("--share-rank must be in [0, K)")
    strong = (cfg["scaling"] == "strong" and world > 1) or args.share > 1
    if args.graph is None:
        args.graph = "compute" if (strong and world > 1 and on_gpu and not cfg["tracking"] and args.intrinsics == "regressed"
                                   and args.optimizer in ("none", "fused")) else "off"
    if args.graph == "off":
        args.graph = None
    if args.halo == "auto":
        # (round 6: `ghost` — every form has now run between real ranks on a GPU, over gloo: tests/test_gpu_multirank.py and, at the metric's size with 8
        # ranks, profiles/r06_multirank_rccl_one_gpu.txt; none has run over RCCL.  The ghost form sends 64 bytes per boundary and step where the early form
        # sends a 3.7 MB frame: the one that DESIGN.md §5 projects to meet the 8-GPU target.  `--halo early` / `oneshot` remain)
        args.halo = "ghost" if (strong and world > 1) else "oneshot"
    if args.graph == "compute" and (not strong or cfg["tracking"] or args.intrinsics != "regressed"):
        raise SystemExit("--graph compute: a frame-sharded run of the flow loss with regressed intrinsics (the tracking loss and the softmin sweep have collectives inside forward / backward)")
    flowmap_amd.set_lazy_surfaces(True)
    if args.no_tap_exchange:
        _ops.options.tap_exchange = False
    if os.environ.get("FLOWMAP_THREE_LAUNCH_BWD"):  # A/B: the planned Procrustes backward as the three launches of round 2
        from flowmap_amd._lib import torch_ops

        torch_ops().set_one_launch_backward(False)

    # ---- inputs: the whole video of this rank's job, then (strong scaling) its shard of it ----
    seed = 1 if (strong or world == 1) else 1 + rank  # strong: every rank builds the SAME video and keeps its frames
    maker = make_scene if cfg["inputs"] == "scene" else make_iid
    depth, wlogit, flows, scene = maker(f_video, h, w, device, seed)
    tracks = None
    if cfg["tracking"]:
        if args.share > 1:
            raise SystemExit("--share with the tracking loss: the pose all-gather needs the other ranks' poses; run the flow loss")
        if world > 1 and not strong:
            raise SystemExit("--tracking with weak scaling: every rank would need its own track set over a shared video; use --scaling strong")
        tracks = make_tracks(f_video, device, seed=100, scene=scene, hw=(h, w))
    total_pairs = f_video - 1
    if strong:
        a, b = shard_pairs(total_pairs, cut_world)[cut_rank]
        lo, hi = shard_frames((a, b))
        # the ghost halo's constants: the flow and mask of the pair before this shard's first frame (backward direction) and of the pair after
        # its last frame (forward direction) — the two terms the neighbours would otherwise send the gradient of
        ghost_prev = (flows.backward[0, a - 1].clone(), flows.backward_mask[0, a - 1].clone()) if (args.halo == "ghost" and a > 0) else None
        ghost_next = (flows.forward[0, b].clone(), flows.forward_mask[0, b].clone()) if (args.halo == "ghost" and b < total_pairs) else None
        depth, wlogit = depth[lo : hi + 1].clone(), wlogit[a:b].clone()
        flows = Flows(*(x[:, a:b].contiguous() for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
        del scene
        if on_gpu:
            torch.cuda.empty_cache()
    f = depth.shape[0]  # frames resident on this rank

    focal0 = 0.85 if cfg["inputs"] == "iid" else 0.8
    parts = (("explicit_depth", 1.0, 100.0),
             ("softmin", 8192, 0.5, 2.0, 60, (1000, 100)) if args.intrinsics == "softmin" else ("regressed", focal0),
             ("procrustes", args.points if args.points > 0 else None, False))

    def direct_modules():  # this package's mirror of model/model.py and the loss classes, constructed by hand
        if parts[1][0] == "softmin":
            from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg

            intrinsics_cfg = IntrinsicsSoftminCfg(*parts[1][:-1], RegressionCfg(*parts[1][-1]))
        else:
            intrinsics_cfg = IntrinsicsRegressedCfg(*parts[1])
        made = Model(ModelCfg(BackboneExplicitDepthCfg(*parts[0]), intrinsics_cfg, ExtrinsicsProcrustesCfg(*parts[2])), num_frames=f, image_shape=(h, w))
        fns = [LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))]
        if tracks is not None:
            from flowmap_amd.loss import LossTracking, LossTrackingCfg

            fns.append(LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01))))
        return made, fns

    package = None
    if args.model == "installed":
        # THE DROP-IN PATH (north_star: "so overfit.py drops it in unchanged"): install() rebinds the registries and import sites of
        # a reference-layout `flowmap` package, and the step below is that package's Model + get_losses, not classes picked by hand.
        package = reference_layout_package()
        flowmap_amd.install(graph=args.training_step == "graph")
        from flowmap.dataset.types import Batch as PackageBatch
        from flowmap.flow.flow_predictor import Flows as PackageFlows
        from flowmap.tracking.track_predictor import Tracks as PackageTracks
        import dataclasses as _dc

        model, loss_fns = installed_modules(parts, f, (h, w), tracks is not None)
        assert type(model.backbone).__module__.startswith("flowmap_amd") and type(loss_fns[0]).__module__.startswith("flowmap_amd"), "install() did not rebind the registries"
        flows = PackageFlows(flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)
        if tracks is not None:
            tracks = [PackageTracks(t.xy, t.visibility, t.start_frame) for t in tracks]
        videos = torch.zeros((1, f, 3, 1, 1), device=device).expand(1, f, 3, h, w)
        batch = PackageBatch(videos, *([None] * (len(_dc.fields(PackageBatch)) - 1)))
    else:
        model, loss_fns = direct_modules()
        batch = Batch(torch.zeros((1, f, 3, 1, 1), device=device).expand(1, f, 3, h, w))
    model = model.to(device)
    model.backbone.depth.data = depth
    model.backbone.weights.data = wlogit
    loss_fn = loss_fns[0]
    track_fn = loss_fns[1] if tracks is not None else None
    if args.items_per_thread:
        loss_fn.items_per_thread = args.items_per_thread
    shard = FrameShard(cut_rank, cut_world, dist, proxy=args.share > 1)
    if strong:
        shard.prepare_flow_loss(loss_fn, flows)  # global valid-sum (one-time all-reduce)
        shard.prepare_model(model)  # softmin sweep on rank 0, broadcast; halo exchange from the gradient hook (flowmap_amd/sharding.py)

    optimizer = None
    if args.optimizer in ("fused", "in_pass"):
        optimizer = flowmap_amd.FusedAdam(model.parameters(), lr=3e-5, capturable=args.graph == "whole")
        if args.optimizer == "in_pass":  # the depth update applied by the flow-loss pass itself (FusedAdam.fuse_depth_update)
            if args.graph:
                raise SystemExit("--optimizer in_pass: no --graph (the step number is a host value)")
            optimizer.fuse_depth_update(model.backbone.depth)
    elif args.optimizer == "torch":
        optimizer = torch.optim.Adam(model.parameters(), lr=3e-5)
    shared = [p for name, p in model.named_parameters() if not name.startswith("backbone.")]  # intrinsics: shared by all frames
    wrapper = None
    if args.training_step != "off":
        if package is None or strong or world > 1 or args.graph:
            raise SystemExit("--training-step: the installed path (--model installed) on one GPU, without --graph / --share")
        try:
            from flowmap.model.model_wrapper_overfit import ModelWrapperOverfit, ModelWrapperOverfitCfg
        except ImportError as exc:
            raise SystemExit(f"--training-step: {package[0]} has no importable model_wrapper_overfit here ({exc})")
        wrapper = ModelWrapperOverfit(ModelWrapperOverfitCfg(3e-5, 32), model, batch, flows, tracks, loss_fns, [])
        wrapper.train()

    def make_step(model, loss_fn, track_fn, batch, flows, tracks):
        def compute():  # zero_grad + forward + backward of this rank's frames: no collective
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, 0)
            loss = loss_fn(batch, flows, None, out, 0)
            loss.backward()
            return loss

        def trainer_step():  # what a trainer's loop does around training_step (Lightning's automatic optimisation: closure = step -> zero_grad -> backward)
            loss = wrapper.training_step(None) / 1  # (Lightning: closure_loss = training_step_output / accumulate_grad_batches)
            (optimizer if optimizer is not None else model).zero_grad(set_to_none=True)
            loss.backward()
            if optimizer is not None:
                optimizer.step()
            wrapper.global_step += 1
            return loss

        if wrapper is not None and wrapper.model is model:
            return trainer_step

        def step():
            tracked = None
            if track_fn is not None and strong:
                model.zero_grad(set_to_none=True)
                out = model(batch, flows, 0)
                loss = loss_fn(batch, flows, None, out, 0)
                tracked = shard.tracking_loss(track_fn, tracks, out, total_pairs)  # global value, this rank's gradients
                (loss + tracked).backward()
            elif track_fn is not None:
                model.zero_grad(set_to_none=True)
                out = model(batch, flows, 0)
                # (every loss is handed the tracks, as ModelWrapperOverfit.training_step does, model_wrapper_overfit.py:57-62: from the second step on
                # the flow loss evaluates the tracking loss ahead of its pass — the tap exchange, flowmap_amd/_ops.py: TapPlan)
                loss = loss_fn(batch, flows, tracks, out, 0) + track_fn(batch, flows, tracks, out, 0)
                loss.backward()
            else:
                loss = compute()
            if strong:
                loss = shard.sync(loss, shared, model.backbone.depth, already_global=tracked)
            if optimizer is not None:
                optimizer.step()
            return loss

        return step

    step = make_step(model, loss_fn, track_fn, batch, flows, tracks)

    # Host housekeeping FIRST: a full cyclic-GC pass over torch's import-time objects costs ~50 ms (flowmap_amd/host.py: freeze_gc).  Until round 4 it
    # ran between the warm-up steps and the timed region and left the GPU idle for that long: the timed region then started from an idle part's
    # power state, not from the state the warm-up steps exist to reach — the flow kernel climbs from 0.78 to 0.95 ms over its first launches after an
    # idle period and needs ~17 launches to settle (profiles/r04_driver_command_ramp.txt, r04_driver_command_gc_placement.txt).  Now nothing but
    # the contract's barrier + synchronize separates precompute, warm-up and the timed steps.  (A/B: FLOWMAP_BENCH_GC=early|late = before / after warm-up.)
    gc_when = os.environ.get("FLOWMAP_BENCH_GC", "first")
    if os.environ.get("FLOWMAP_TAP_EXCHANGE_MIN_BYTES"):  # A/B: the depth size from which the tap exchange engages (default 128 MB, flowmap_amd/_ops.py)
        from flowmap_amd import _ops as _fm_ops

        _fm_ops.options.tap_exchange_min_bytes = int(os.environ["FLOWMAP_TAP_EXCHANGE_MIN_BYTES"])
    if os.environ.get("FLOWMAP_PLAIN_LOSS"):  # A/B: the losses as plain tensors (autograd's ones_like fill + the flow loss's seed check: two more launches)
        from flowmap_amd import _ops as _fm_ops

        _fm_ops.options.unit_seed = False
    if gc_when == "first":
        flowmap_amd.freeze_gc()
    # one-time precompute, outside warm-up and timing whatever W is: the first step packs the constant
    # flows / masks and reduces the valid sums, the second one plans the static scatters (SURVEY §8d:
    # the metric excludes one-time precompute)
    for _ in range(3):
        step()
    cpu_sample = None
    if args.release_originals:
        if strong:
            raise SystemExit("--release-originals: single-GPU runs")
        n_cpu = _cpu_frames(args, f_video, h, w)
        if world == 1 and n_cpu >= 2 and args.points > 0:  # the CPU leg's inputs, before the originals go
            cpu_sample = Flows(*(x[:, : n_cpu - 1].cpu() for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
        released = flowmap_amd.release_flow_originals(flows)
        if on_gpu:
            torch.cuda.empty_cache()
        print(f"[bench] released {released / 1e9:.1f} GB of flow originals (the fused flow loss reads the packed copy)", file=sys.stderr)
        step()
    early_halo = False
    if args.halo == "early" and strong:
        early_halo = shard.enable_early_halo(model.backbone.depth)  # (collective over neighbours; the scatter plans exist by now)
        step()
    if args.halo == "ghost" and strong:
        early_halo = shard.enable_ghost_halo(model.backbone.depth, ghost_prev, ghost_next)
        shard.set_proxy_ghost_flows(flows)  # (the --share proxy: the shard's own boundary pairs stand in for the neighbours')
        step()
    if dist is not None:  # create the RCCL communicators / P2P channels outside the timed region
        shard.exchange_halo(torch.zeros((2, h, w), device=device))
        dist.all_reduce(torch.zeros(4, device=device))
    eager_flow_ms = []
    if args.graph:
        if args.optimizer == "torch":
            raise SystemExit("--graph: --optimizer none|fused")
        if on_gpu:  # kernel events are not recorded inside a graph: time the flow kernel on a few eager steps first
            _ops.flow_kernel_timing(True)
            for _ in range(5):
                step()
            torch.cuda.synchronize(device)
            eager_flow_ms = _ops.flow_kernel_times()
            _ops.flow_kernel_timing(False)
        if args.graph == "compute":
            def forward_only():  # zero_grad + model + flow loss of this rank's frames (backward is captured as a second graph)
                model.zero_grad(set_to_none=True)
                return loss_fn(batch, flows, None, model(batch, flows, 0), 0)

            # (the capture holds no collective — those stay eager around the graphs — so a rank whose capture fails can still meet the others
            # in the agreement below; if ANY rank could not capture, every rank runs the eager step: slower, but a number instead of a hang)
            sharded, failure = None, None
            try:
                sharded = flowmap_amd.GraphedShardedStep(forward_only, shard, shared, model.backbone.depth, warmup=3)
            except Exception as exc:  # noqa: BLE001
                failure = exc
            ok = torch.tensor([0.0 if sharded is None else 1.0], device=device)
            if dist is not None:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                print(f"[bench] rank {rank}: hipGraph capture of the sharded step failed ({failure!r}); every rank falls back to the eager step", file=sys.stderr)
                args.graph = None
                eager_step = step

                def step():  # noqa: F811
                    return eager_step()
            else:
                def step():  # noqa: F811
                    loss = sharded()
                    if optimizer is not None:
                        optimizer.step()
                    return loss
        else:
            step = flowmap_amd.GraphedStep(step, warmup=3)  # noqa: F811
    if gc_when == "early":
        flowmap_amd.freeze_gc()
    for _ in range(args.warmup):
        step()
    if gc_when == "late":
        flowmap_amd.freeze_gc()
    if not args.graph and on_gpu:
        _ops.flow_kernel_timing(True)  # HIP events on the launch stream around the fused flow kernel / track_pairs
    if dist is not None:
        dist.barrier()
    sync_device()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if dist is not None:
        dist.barrier()
    sync_device()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    flow_ms = _ops.flow_kernel_times() if not args.graph else eager_flow_ms
    track_ms = _ops.flow_kernel_times(tracking=True)
    sustained = None
    if world == 1 and on_gpu and not args.graph and args.sustained_steps > 0:  # the same step, straight on: the state a long run is in
        t1 = time.perf_counter()
        for _ in range(args.sustained_steps):
            step()
        sync_device()
        dt = time.perf_counter() - t1
        more = _ops.flow_kernel_times()
        _ops.flow_kernel_times(tracking=True)
        sustained = {"steps": args.sustained_steps, "after_steps": args.steps, "ms_per_step": dt / args.sustained_steps * 1e3, "value": args.sustained_steps / dt,
                     "kernel_ms": sum(more) / max(len(more), 1),
                     "note": "informational: the steps that follow the timed region without a pause — `value` above is the contract's K steps after W warm-up steps, "
                             "which at K = 20, W = 5 lie on the power-management transient of a GPU that was idle (kernel_ms_per_launch shows it)"}
    _ops.flow_kernel_timing(False)
    direct = None
    if package is not None and world == 1 and args.share <= 1 and not args.graph and optimizer is None and not args.release_originals:
        # the same step with this package's own Model and loss classes constructed by hand (what rounds 1-4 timed), on the same parameters and inputs,
        # straight after the steps above: the drop-in path must cost what the hand-built one costs
        direct_model, direct_fns = direct_modules()
        direct_model = direct_model.to(device)
        direct_model.backbone.depth.data = model.backbone.depth.data
        direct_model.backbone.weights.data = model.backbone.weights.data
        direct_step = make_step(direct_model, direct_fns[0], direct_fns[1] if tracks is not None else None,
                                Batch(batch.videos), Flows(flows.forward, flows.backward, flows.forward_mask, flows.backward_mask), tracks)
        n_direct = args.sustained_steps if (on_gpu and args.sustained_steps > 0) else args.steps
        for _ in range(3 + args.warmup):
            direct_loss = direct_step()
        sync_device()
        t1 = time.perf_counter()
        for _ in range(n_direct):
            direct_loss = direct_step()
        sync_device()
        direct = {"what": "flowmap_amd.model.model.Model + flowmap_amd.loss classes constructed by hand (--model direct), same parameters and inputs, "
                          f"{n_direct} steps straight after the steps above (compare with `sustained`, which the installed path ran just before)",
                  "steps": n_direct, "ms_per_step": (time.perf_counter() - t1) / n_direct * 1e3, "loss": float(direct_loss.item())}
        del direct_model, direct_step
    # What ran so far — precompute, warm-up, the timed steps, the sustained steps, the hand-built twin — is the product path.  The oracle is a
    # checker: the legs below (`ate`, `cpu_baseline`) import it; nothing above may have (the stand-in package reaches it lazily and refuses GPU tensors).
    oracle_loaded_by_product_path = any(name == "oracle" or name.startswith("oracle.") for name in sys.modules)
    kernel_ms = sum(flow_ms) / max(len(flow_ms), 1)
    traffic, traffic_src = None, None
    taps_on = _ops.counters["flow_tap_passes"] > 0
    for name in ("r06_flow_kernel_traffic.json", "r05_flow_kernel_traffic.json", "r04_flow_kernel_traffic.json", "r03_flow_kernel_traffic.json", "r02_flow_kernel_traffic.json", "r01_flow_kernel_traffic.json"):  # HBM bytes per launch from the PMC passes (same workload only)
        try:
            rec = json.loads((ROOT / "profiles" / name).read_text())
            if {k: rec["workload"][k] for k in ("frames", "height", "width")} == {"frames": f, "height": h, "width": w}:
                in_pass_on = args.optimizer == "in_pass" and optimizer.counters["in_pass_updates"] > 0
                key = {(False, False): "", (False, True): "flow_fused_kernel_taps", (True, False): "flow_fused_kernel_adam",
                       (True, True): "flow_fused_kernel_adam_taps"}[(in_pass_on, taps_on)]  # (the kernel instance that ran)
                if key and key not in rec:
                    continue
                entry = rec[key] if key else rec.get("driver_command_round4", rec)  # (r05: the plain instance is the record's top level)  # (the plain instance: re-measured in round 4 on the driver's own command)
                traffic, traffic_src = entry["hbm_bytes_per_launch"], f"profiles/{name}"
                break
        except Exception:
            pass
    n = h * w
    algo_bytes = n * (8 * f + 24 * (f - 1))  # B_flow per launch (this rank's frames)
    in_pass = args.optimizer == "in_pass" and optimizer.counters["in_pass_updates"] > 0
    if in_pass:  # depth, exp_avg, exp_avg_sq read and rewritten (24 B per pixel and frame), no dL/ddepth written
        algo_bytes = n * (24 * f + 24 * (f - 1))
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    launches, launch_names = (None, [])
    under_rocprof = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR"))
    if on_gpu and rank == 0 and (args.count_launches or (world == 1 and not args.graph and not under_rocprof and not os.environ.get("FLOWMAP_BENCH_NO_PROFILER"))):
        launches, launch_names = count_launches(step, device)  # (two tracers in one process do not mix: skipped under rocprofv3)

    if rank == 0:
        jobs = 1 if (strong or world == 1) else world  # weak scaling: every rank completes its own workload per step
        ms_per_step = elapsed / args.steps * 1e3
        step_gbs = algo_bytes / (ms_per_step * 1e-3) / 1e9
        workload = (f"{cfg['ref']}: {f_video} frames @ {h}x{w}, {cfg['inputs']} inputs, flow loss (huber 0.01, weight 1000)"
                    + (f" + tracking loss (weight 100): {len(tracks)} segments x {tracks[0].xy.shape[2]} tracks" if tracks else "")
                    + f", explicit-depth backbone, {args.intrinsics} intrinsics, Procrustes P={args.points if args.points > 0 else 'all pixels'}; "
                    + (f"modules built by the {package[0]} package after flowmap_amd.install(); " if package is not None else "modules constructed by hand (--model direct); ")
                    + ("stepped through that package's ModelWrapperOverfit.training_step" + (" replayed as hipGraphs (install(graph=True))" if args.training_step == "graph" else "") + "; " if wrapper is not None else "")
                    + "fwd+bwd, "
                    + ("no optimiser" if optimizer is None else f"+ Adam step ({type(optimizer).__module__}.{type(optimizer).__name__}"
                       + (", depth update inside the flow pass)" if args.optimizer == "in_pass" and optimizer.counters["in_pass_updates"] > 0
                          else ", fuse_depth_update requested but the touched set is too large: separate update)" if args.optimizer == "in_pass" else ")"))
                    + ("; whole step replayed as one hipGraph" if args.graph == "whole" else
                   "; forward + backward replayed as one hipGraph, collectives issued eagerly after the replay" if args.graph == "compute" else "")
                + (f"; PROXY: rank {cut_rank}'s share of a {cut_world}-GPU strong-scaling run on one GPU (pairs [{a}, {b}) of {total_pairs}, {f} frames incl. halo), "
                   "every collective on a one-rank RCCL communicator, halo exchange replaced by its local copies/adds; xGMI wire time NOT included" if args.share > 1 else ""))
        result = {
            "metric": "overfit iters/sec (150 frames @ 720p) at 1/2/4/8 MI355X; final ATE vs ref",
            "value": jobs * args.steps / elapsed,
            "unit": "iters/sec (one iter = fwd+bwd over the whole video"
                    + (", sharded by frame pairs over the GPUs)" if strong else "; weak scaling: one video per GPU, aggregate over GPUs)" if world > 1 else ")"),
            "n_gpus": world,
            # the ranks that actually met in the process group (RCCL on the GPU; gloo in the CPU dry run), and the frames each of them holds
            "rccl_ranks": dist.get_world_size() if (dist is not None and not args.share) else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "frame_split": ([list(shard_frames(pr)) for pr in shard_pairs(total_pairs, cut_world)] if strong
                            else [[0, f_video - 1]] * world),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "config": args.config,
                "inputs": cfg["inputs"],
                "video_frames": f_video if (strong or world == 1) else f_video * world,
                "frames_per_gpu": f,
                "pairs_per_gpu": f - 1,
                "height": h,
                "width": w,
                "halo_exchange": (("ghost (the neighbour's dense part evaluated from its 64-byte pose, sparse correction after backward)" if args.halo == "ghost" else
                                   "early (dense part after the flow pass, sparse correction after backward)") if early_halo else "one shot after backward") if strong else None,
                "ranks_share_one_gpu": bool(args.one_gpu),
                "parallelism": (f"frame-pair shards x{world} (1-frame halo, packed all-reduce of loss + shared gradients, halo exchange)" if strong
                                else f"{world} independent 150-frame shards" if world > 1 else "single GPU"),
                "loss": float(loss.item()),
            },
            "roofline": {
                "kernel": "fm::flow_fused_kernel<VEC=4, huber, GRAD, PACKED" + (", ADAM>" if in_pass else ">"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_measured_in": traffic_src,
                "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes of an earlier run of this workload (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950 "
                                "correction), not a live counter: traffic is data-independent" if traffic is not None else None,
                "step_frac": step_gbs / HBM_PEAK_GBS,
                "step_achieved": step_gbs,
                "launches_per_step": launches,
                "kernels_of_a_step": launch_names,
                "kernel_timing": "HIP events on the launch stream, inside the timed region" if not args.graph else "HIP events on 5 eager steps before the capture (a replayed graph records none)",
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": kernel_ms,
                "launches_timed": len(flow_ms),
                # every launch of the timed region in order (is a short run still on a clock ramp?  VERDICT r3 item 1b)
                "kernel_ms_per_launch": [round(x, 4) for x in flow_ms] if len(flow_ms) <= 200 else None,
                "kernel_ms_first5_last5": [sum(flow_ms[:5]) / 5, sum(flow_ms[-5:]) / 5] if len(flow_ms) >= 10 else None,
            },
        }
        # the metric's second half ("final ATE vs ref"), measured by THIS run (ate_leg above); records of earlier runs are quoted under
        # `quoted_records`, never beside the measured values
        fixture = Path(args.ate_fixture) if args.ate_fixture else default_ate_fixture()
        want_ate = args.ate == "on" or (args.ate == "auto" and world == 1 and args.share <= 1 and on_gpu and args.config == "c1" and not args.whole
                                        and (f_video, h, w) == (150, 720, 1280) and tracks is None and optimizer is None and not args.graph
                                        and not under_rocprof)
        if want_ate and fixture is not None and fixture.exists():
            try:
                result["ate"] = ate_leg(device, fixture)
            except Exception as exc:  # noqa: BLE001  (the timing line must not be lost to the comparison leg)
                result["ate"] = {"measured_by_this_run": False, "failed": repr(exc)[:400]}
        if want_ate and c0_ate_fixture() is not None:
            # ... and the one place where the softmin window, the `enable_after` gate and all 2000 Adam steps of the reference's schedule meet the
            # reference end to end: configs[0] on config/overfit.yaml's own schedule
            try:
                result["ate_c0_full_schedule"] = ate_leg(device, c0_ate_fixture())
            except Exception as exc:  # noqa: BLE001
                result["ate_c0_full_schedule"] = {"measured_by_this_run": False, "failed": repr(exc)[:400]}
        if args.default_resolution == "on" or (args.default_resolution == "auto" and want_ate and wrapper is None):
            result["default_resolution"] = default_resolution_leg()
        quoted = {}
        try:
            rec = json.loads((ROOT / "profiles" / "r04_ate_150x360x640_vs_imported_reference.json").read_text())
            quoted["ate_150x360x640_round4"] = {"record": "profiles/r04_ate_150x360x640_vs_imported_reference.json", "ate_reference": rec["ate_reference_path_cpu"],
                                                "ate_flowmap_amd": rec["ate_flowmap_amd"], "note": "200 Adam steps at lr 1e-3, measured in round 4, not by this run"}
        except Exception:  # noqa: BLE001
            pass
        if package is not None:
            result["via_install"] = {
                "package": package[0], "package_path": package[1],
                "what": "flowmap_amd.install() on that package, then ITS Model (get_backbone / get_intrinsics / get_extrinsics) and get_losses: `value`, `ms_per_step`, "
                        "`roofline` and `sustained` of this line are measured on this path",
                "modules": {"backbone": f"{type(model.backbone).__module__}.{type(model.backbone).__name__}",
                            "intrinsics": f"{type(model.intrinsics).__module__}.{type(model.intrinsics).__name__}",
                            "extrinsics": f"{type(model.extrinsics).__module__}.{type(model.extrinsics).__name__}",
                            "model": f"{type(model).__module__}.{type(model).__name__}",
                            "losses": [f"{type(fn).__module__}.{type(fn).__name__}" for fn in loss_fns]},
                "ms_per_step": ms_per_step, "launches_per_step": launches,
                # every kernel of a step is this library's (a torch kernel here would mean some name fell through to eager arithmetic), and the
                # oracle had not been imported when the timed steps ended
                "kernels_not_from_libflowmap_hip": [k for k in launch_names if "fm::" not in k and not k.startswith(("softmin_", "random_"))
                                                    and "#" not in k],
                "oracle_imported_by_the_timed_path": oracle_loaded_by_product_path,
                "sustained_ms_per_step": sustained["ms_per_step"] if sustained is not None else None,
                "training_step": None if wrapper is None else {
                    "mode": args.training_step,
                    "what": "the step is the package's ModelWrapperOverfit.training_step in a trainer's order (training_step, zero_grad, backward, optimiser, global_step + 1)"
                            + ("; install(graph=True): forward + losses and backward replayed as two hipGraphs" if args.training_step == "graph" else ""),
                    **({k: getattr(wrapper.__dict__.get("_fm_graphed_training"), k, None) for k in ("captures", "replays", "disabled")} if args.training_step == "graph" else {})},
                "direct": direct,
                "installed_over_direct": (sustained["ms_per_step"] / direct["ms_per_step"]) if (sustained is not None and direct is not None) else None,
            }
        if sustained is not None:
            n_bytes = algo_bytes
            sustained["roofline_frac"] = (n_bytes / (sustained["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if sustained["kernel_ms"] > 0 else None
            result["sustained"] = sustained
        if track_ms:
            residuals = sum(int(t.xy.shape[1]) ** 2 * int(t.xy.shape[2]) for t in tracks)
            if strong:  # this rank evaluates the sources it owns: its share of the residuals
                own = FrameShard.owned_sources(total_pairs, world, rank)
                residuals = sum(sum(1 for fr in range(t.xy.shape[1]) if own[0] <= t.start_frame + fr < own[1]) * int(t.xy.shape[1]) * int(t.xy.shape[2])
                                for t in tracks)
            t_ms = sum(track_ms) / len(track_ms)
            gflops = residuals * TRACK_FLOPS_PER_RESIDUAL / (t_ms * 1e-3) / 1e9
            result["roofline_tracking"] = {
                "kernel": "fm::track_pairs_kernel<huber, GRAD> (+ track_reduce, finalize" + (", tap_grad: one fm_track_loss_fused_fwd_taps call)" if _ops.counters["flow_tap_absorbs"] else ": one fm_track_loss_fused_fwd call)"),
                "tap_exchange": {"flow_passes_with_taps": _ops.counters["flow_tap_passes"], "absorbed": _ops.counters["flow_tap_absorbs"],
                                 "sampled_from_tap_image": _ops.counters["track_tap_samples"]},
                "bound": "valu",
                "achieved": gflops,
                "peak": FP32_PEAK_GFLOPS,
                "unit": "GFLOP/s",
                "frac": gflops / FP32_PEAK_GFLOPS,
                "traffic": None,
                "residuals_per_launch": residuals,
                "flops_per_residual": TRACK_FLOPS_PER_RESIDUAL,
                # the kernel's own currency: VALU issue slots.  SQ_INSTS_VALU of track_pairs = 26 019 per wave x 1 984 waves at C2
                # (profiles/r05_c2_sq_counters_scaled_residual.csv) = 60.3 instructions per residual and lane; a wave64 instruction occupies its SIMD for 4 cycles
                # (packed fp32 ones for ~8: the floor below is optimistic), 1024 SIMDs, ~2.1 GHz sustained under load
                "issue_roofline": {"valu_instructions_per_residual": TRACK_VALU_PER_RESIDUAL, "measured_in": "profiles/r05_c2_sq_counters_scaled_residual.csv",
                                   "issue_bound_ms": residuals * TRACK_VALU_PER_RESIDUAL / 64 * 4 / (1024 * 2.1e9) * 1e3,
                                   "frac": (residuals * TRACK_VALU_PER_RESIDUAL / 64 * 4 / (1024 * 2.1e9) * 1e3) / t_ms if t_ms > 0 else None},
                "kernel_ms": t_ms,
                "launches_timed": len(track_ms),
            }
        if args.share > 1:
            result["proxy"] = {"share_of": cut_world, "rank": cut_rank, "pairs": [a, b], "frames_resident": f, "video_pairs": total_pairs,
                               "per_rank_ms_per_step": ms_per_step, "wire_time_included": False}
        cpu_frames = _cpu_frames(args, f_video, h, w)
        if world == 1 and args.share <= 1 and cpu_frames >= 2 and args.points > 0:
            cpu_frames = min(cpu_frames, f)
            focal0 = 0.85 if cfg["inputs"] == "iid" else 0.8
            dt, cores, first, cpu_loss = cpu_baseline(model.backbone.depth.data, model.backbone.weights.data, cpu_sample if cpu_sample is not None else flows,
                                                      focal0, cpu_frames, h, w, args.points, args.cpu_iters, args.cpu_threads)
            whole = cpu_frames == f_video
            scaled = dt if whole else dt * (f_video - 1) / (cpu_frames - 1)  # per-pair cost is constant (optimistic for the CPU)
            # What the port costs next to the code it stands for: oracle/make_cpu_calibration.py timed the IMPORTED reference (its own Model +
            # LossFlow) and the port interleaved on the same inputs in the build container (the GPU box has no reference to time)
            calibration = None
            try:
                cal = json.loads((ROOT / "tests" / "golden" / "cpu_calibration.json").read_text())
                calibration = {"port_over_reference": cal["port_over_reference"], "measured_at": cal["port_over_reference_measured_at"],
                               "threads": cal["threads"], "host": cal["host"], "record": "tests/golden/cpu_calibration.json (" + cal["made_by"] + ")",
                               "by_size": {"x".join(str(r[k]) for k in ("frames", "height", "width")): round(r["port_over_reference"], 3) for r in cal["sizes"]}}
            except Exception:  # noqa: BLE001
                pass
            result["cpu_baseline"] = {
                "value": 1.0 / scaled,
                # the same measurement restated for the reference's own code: seconds x (reference / port) from the calibration record
                "reference_equivalent_value": (calibration["port_over_reference"] / scaled) if calibration else None,
                "port_over_reference": calibration,
                "unit": "iters/sec",
                "cores": cores,
                "kind": "port",
                "frames": cpu_frames,
                "whole_workload": whole,
                "host_logical_cpus": os.cpu_count(),
                "sample": (f"oracle (PyTorch-CPU port of the reference path, flow loss), {cores} torch threads, "
                           + (f"the WHOLE workload: the same {f_video} frames @ {h}x{w} ({cfg['inputs']} inputs) the GPU leg ran on" if whole else
                              f"SAMPLE: the first {cpu_frames} of the {f_video} frames @ {h}x{w} ({cfg['inputs']} inputs) the GPU leg ran on, scaled by pairs "
                              f"({f_video - 1}/{cpu_frames - 1})")
                           + f", fwd+bwd, {args.cpu_iters} timed iteration(s) after 1 warm-up ({first:.1f} s): {dt:.3f} s/iter"),
                "sample_seconds_per_iter": dt,
                "loss": cpu_loss,
                "loss_rel_diff_vs_gpu": (abs(cpu_loss - float(loss.item())) / abs(cpu_loss)) if (whole and optimizer is None) else None,
            }
        if args.torch_baseline > 0 and on_gpu and world == 1:
            result["rocm_torch_baseline"] = torch_baseline(f_video, h, w, args.points, args.torch_baseline)
        elif (f_video, h, w, args.points) == (150, 720, 1280, 1000) and tracks is None and optimizer is None:
            # (measured once, not by this run — like `ate`: the reference's op sequence on stock PyTorch-ROCm takes 20 s per step)
            record = ROOT / "profiles" / "r04_stock_pytorch_rocm_150_frames_in_16_frame_windows.json"
            try:
                rec = json.loads(record.read_text())
                quoted["rocm_torch_baseline"] = {"record": f"profiles/{record.name}", "ms_per_step": rec["ms_per_step"], "iters_per_sec": rec["iters_per_sec"],
                                                 "frames_per_call": rec["frames_per_call"], "note": "measured once (--torch-baseline N measures it in this run); " + rec["note"]}
            except Exception:  # noqa: BLE001
                pass
        if quoted:
            result["quoted_records"] = {"what": "numbers of EARLIER runs kept under profiles/, quoted for context: nothing under this key was measured by this run", **quoted}
        print(json.dumps(result), file=result_stream, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
