"""Frame-pair sharding (flowmap_amd/sharding.py) with world_size 2 over gloo on CPU:
the product host layer (through the host test double) on each shard + FrameShard.sync
must reproduce the unsharded oracle: loss, dL/dfocal, dL/ddepth incl. the halo frame."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent

from flowmap_amd.sharding import shard_frames, shard_pairs  # noqa: E402


def test_shard_pairs_cover_everything():
    for pairs in (1, 5, 149, 1199):
        for world in (1, 2, 4, 8):
            ranges = shard_pairs(pairs, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == pairs
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
            for r in ranges:
                first, last = shard_frames(r)
                assert last - first == r[1] - r[0]


def _focal_close(got, ref, ref32=None):
    """dL/dfocal as helpers.compare_step judges it: 1e-4 of the fp64 oracle's, or twice the gap of the reference path's own fp32 evaluation."""
    from helpers import focal_close

    focal_close(got, ref, ref32)


def _shard_close(got, truth, ref32, what):
    """A shard's slice of a gradient against the fp64 oracle's: 1e-4, or twice the gap the reference path's own fp32 evaluation (the fp32
    oracle) has on the same slice — measured here (these are i.i.d. inputs: conftest.assert_close_or_reference_gap)."""
    from conftest import assert_close_or_reference_gap

    assert_close_or_reference_gap(got, truth, ref32, 1e-4, what=what)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, f, h, w, points, out_path, with_tracks=False, softmin=False, device="cpu", backend="gloo"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":  # RCCL between ranks that share ONE GPU: a host of its own per rank, the socket transport on loopback (tools/probes/rccl_one_gpu_probe.py)
        import datetime

        os.environ.update(NCCL_HOSTID=f"flowmap-amd-test-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(torch.device(device))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device), timeout=datetime.timedelta(minutes=3))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import flowmap_amd
    from flowmap_amd import Batch, Flows, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard
    from helpers import build_host_sim
    from oracle import flowmap_oracle as orc

    dev = torch.device(device)
    if dev.type == "cpu":
        _lib.set_library_for_testing(build_host_sim())
    # (device = "cuda:0": the ranks share the one GPU of a gpurun box and meet over gloo, which moves GPU tensors — the HIP library on every rank,
    # real peers: tests at the bottom of this file, -m gpu)
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    nf = hi - lo + 1
    intrinsics_cfg = IntrinsicsRegressedCfg("regressed", 0.85)
    if softmin:
        from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg

        intrinsics_cfg = IntrinsicsSoftminCfg("softmin", 64, 0.5, 2.0, 7, None)
    cfg = ModelCfg(
        BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
        intrinsics_cfg,
        ExtrinsicsProcrustesCfg("procrustes", points, False),
    )
    model = Model(cfg, num_frames=nf, image_shape=(h, w))
    if softmin:  # the same sample on every run (the reference draws randperm(h*w)[:P] per step)
        fixed = torch.linspace(0, h * w - 1, 64).to(torch.int64)
        model.intrinsics._draw_indices = lambda count, device: fixed.to(device)
    model.backbone.depth.data = depth[lo : hi + 1].clone()
    model.backbone.weights.data = wlogit[a:b].clone()
    model = model.to(dev)
    local = Flows(flows.forward[:, a:b].contiguous().to(dev), flows.backward[:, a:b].contiguous().to(dev),
                  flows.forward_mask[:, a:b].contiguous().to(dev), flows.backward_mask[:, a:b].contiguous().to(dev))
    batch = Batch(torch.zeros((1, nf, 3, h, w), device=dev))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)  # softmin sweep on rank 0 + broadcast; halo exchange from the depth gradient's hook
    out = model(batch, local, 0)
    loss = loss_fn(batch, local, None, out, 0)
    track_total = None
    if with_tracks:
        from helpers import to_tracks

        tracks = to_tracks(orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5), dev)  # GLOBAL frame indices
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
        # global value on every rank; its autograd gradients are this rank's share of the term
        track_total = shard.tracking_loss(track_fn, tracks, out, f - 1)
        (loss + track_total).backward()
    else:
        loss.backward()
    shared = [p for name, p in model.named_parameters() if not name.startswith("backbone.")]  # a LIST of shared parameters
    total = shard.sync(loss, shared, model.backbone.depth, already_global=track_total)
    g_focal = None if softmin else model.intrinsics.focal_length.grad.cpu().clone()
    torch.save(
        {"loss": total.cpu().clone(), "track": None if track_total is None else track_total.detach().cpu().clone(), "g_focal": g_focal,
         "k": out.intrinsics.detach().cpu().clone(), "g_depth": model.backbone.depth.grad.cpu().clone(),
         "g_w": model.backbone.weights.grad.cpu().clone(), "frames": (lo, hi), "pairs": (a, b)},
        f"{out_path}.{rank}",
    )
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_match_unsharded_oracle(tmp_path):
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 7, 12, 16, 40
    world = 2
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, _free_port(), f, h, w, points, out), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=torch.float32)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert_close(r["loss"], ref["total"], 1e-5, what="global loss")
        _focal_close(r["g_focal"], ref, ref32)
        lo, hi = r["frames"]
        a, b = r["pairs"]
        assert_close(r["g_depth"], ref["g_depth"][lo : hi + 1], 1e-4, what="g_depth shard (halo summed)")
        _shard_close(r["g_w"], ref["g_wlogit"][a:b], ref32["g_wlogit"][a:b], "g_wlogit shard")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_tracking_matches_unsharded_oracle(tmp_path, world):
    """Flow + tracking losses with the video split over `world` ranks: track segments straddle
    the shard borders (sources evaluated where their depth lives, poses all-gathered and chained,
    [sum, count] and the pose gradients all-reduced) — loss and every gradient as unsharded."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 8, 12, 16, 40
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, _free_port(), f, h, w, points, out, True), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    otracks = orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5)
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float32)
    assert float(ref["loss_tracking"]) > 0
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert_close(r["track"], ref["loss_tracking"], 1e-5, what="global tracking loss")
        assert_close(r["loss"], ref["total"], 1e-5, what="global loss")
        _focal_close(r["g_focal"], ref, ref32)
        lo, hi = r["frames"]
        a, b = r["pairs"]
        _shard_close(r["g_depth"], ref["g_depth"][lo : hi + 1], ref32["g_depth"][lo : hi + 1], "g_depth shard (halo summed)")
        _shard_close(r["g_w"], ref["g_wlogit"][a:b], ref32["g_wlogit"][a:b], "g_wlogit shard")


@pytest.mark.timeout(300)
def test_sharded_softmin_intrinsics_match_the_unsharded_step(tmp_path):
    """The reference's default intrinsics for its first 1000 steps under frame sharding: the candidate sweep fits
    frames (0, 1) of the VIDEO, so rank 0 runs it and broadcasts K (gradients reduced back onto rank 0).  Every
    rank must see the K of the unsharded step, and loss / gradients must be those of the unsharded step."""
    sys.path.insert(0, str(ROOT / "tests"))
    import flowmap_amd
    from conftest import assert_close
    from flowmap_amd import Batch, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, Model, ModelCfg
    from helpers import build_host_sim, to_flows
    from oracle import flowmap_oracle as orc

    f, h, w, points, world = 7, 12, 16, 40, 3
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, _free_port(), f, h, w, points, out, False, True), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]

    # the unsharded step through the same product path (itself checked against the oracle in cases.case_softmin_step)
    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.set_lazy_surfaces(True)
    try:
        depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
        model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsSoftminCfg("softmin", 64, 0.5, 2.0, 7, None),
                               ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=f, image_shape=(h, w))
        fixed = torch.linspace(0, h * w - 1, 64).to(torch.int64)
        model.intrinsics._draw_indices = lambda count, device: fixed.to(device)
        model.backbone.depth.data, model.backbone.weights.data = depth.clone(), wlogit.clone()
        fl = to_flows(flows, "cpu")
        batch = Batch(torch.zeros((1, f, 3, h, w)))
        o = model(batch, fl, 0)
        loss = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))(batch, fl, None, o, 0)
        loss.backward()
    finally:
        flowmap_amd.set_lazy_surfaces(False)
        _lib.set_library_for_testing(None)
    for r in res:
        lo, hi = r["frames"]
        a, b = r["pairs"]
        assert_close(r["k"][0, 0], o.intrinsics[0, 0], 1e-6, what="K on every rank = the video's K")
        assert_close(r["loss"], loss.detach(), 1e-5, what="global loss")
        assert_close(r["g_depth"], model.backbone.depth.grad[lo : hi + 1], 1e-4, abs_=1e-7, what="g_depth shard (sweep gradient on rank 0, halo summed)")
        assert_close(r["g_w"], model.backbone.weights.grad[a:b], 3e-4, abs_=1e-7, what="g_wlogit shard")


# ---- round 3: the capture-safe sync (persistent buffer over several steps), bucketed shared-module gradients, in-pass Adam on shards ----


def _shared_backbone(frames_total, lo, hi, h, w, depth, wlogit, hidden):
    """Test-only backbone: explicit depth / weight logits (frame-local) modulated by a SHARED network of ~hidden² parameters —
    the role BackboneMidas plays in the reference (backbone_midas.py:42-127): every frame's depth depends on the same weights."""
    from torch import nn

    from flowmap_amd.model.projection import LazyWeights
    from flowmap_amd.types import BackboneOutput

    class SharedBackbone(nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.depth = nn.Parameter(depth[lo : hi + 1].clone())
            self.weights = nn.Parameter(wlogit[lo:hi].clone())
            self.register_buffer("codes", torch.randn((frames_total, 8), generator=g)[lo : hi + 1].clone())
            torch.manual_seed(11)  # the same initial weights on every rank and in the unsharded run
            self.net = nn.Sequential(nn.Linear(8, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, h * w))

        def forward(self, batch, flows):
            offset = 0.05 * torch.tanh(self.net(self.codes)).view(-1, h, w)
            return BackboneOutput((self.depth * torch.exp(offset))[None], LazyWeights(self.weights[None], 100.0))

    return SharedBackbone()


def _bucket_worker(rank, world, port, f, h, w, points, hidden, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import flowmap_amd
    from flowmap_amd import Batch, Flows, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard, SharedGradientBuckets
    from helpers import build_host_sim
    from oracle import flowmap_oracle as orc

    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    nf = hi - lo + 1
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=nf, image_shape=(h, w))
    model.backbone = _shared_backbone(f, lo, hi, h, w, depth, wlogit, hidden)
    local = Flows(flows.forward[:, a:b].contiguous(), flows.backward[:, a:b].contiguous(),
                  flows.forward_mask[:, a:b].contiguous(), flows.backward_mask[:, a:b].contiguous())
    batch = Batch(torch.zeros((1, nf, 3, h, w)))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist if world > 1 else None)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    net_params = list(model.backbone.net.parameters())
    buckets = SharedGradientBuckets(shard, net_params, bucket_mb=25.0)
    launched = []
    for step in range(2):  # two steps: the persistent buffers are reused, `.grad` of the shared parameters lives in them
        model.zero_grad(set_to_none=True)
        out = model(batch, local, 0)
        loss = loss_fn(batch, local, None, out, 0)
        loss.backward()
        launched.append(sum(bk[3] is not None for bk in buckets.buckets))  # buckets whose all-reduce started DURING backward
        buckets.finish()
        total = shard.sync(loss, [model.intrinsics.focal_length], model.backbone.depth)
    if rank == 0:
        assert sum(p.numel() for p in net_params) >= 10_000_000 or hidden < 3000
        torch.save({"loss": total.clone(), "g_net": [p.grad.clone() for p in net_params], "g_focal": model.intrinsics.focal_length.grad.clone(),
                    "g_depth": model.backbone.depth.grad.clone(), "frames": (lo, hi), "buckets": len(buckets.buckets), "launched": launched,
                    "in_bucket": all(p.grad.data_ptr() == view.data_ptr() for bk in buckets.buckets for p, view in bk[1])}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_shared_module_gradients_are_bucketed_and_match_the_unsharded_step(tmp_path):
    """A 10^7-parameter shared module next to the explicit-depth parameters (north_star: "RCCL all-reduce of the
    intrinsics/shared-backbone gradients"; BackboneMidas in the reference): SharedGradientBuckets reduces its gradient in
    25 MB buckets launched from gradient hooks while backward still runs; the result equals the unsharded gradient."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close

    f, h, w, points, hidden = 7, 12, 16, 40, 3162
    sharded, single = str(tmp_path / "sharded.pt"), str(tmp_path / "single.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), f, h, w, points, hidden, sharded), nprocs=2, join=True)
    mp.spawn(_bucket_worker, args=(1, _free_port(), f, h, w, points, hidden, single), nprocs=1, join=True)
    got, ref = torch.load(sharded), torch.load(single)
    assert got["buckets"] >= 2 and got["in_bucket"]  # 40 MB of parameters: more than one bucket; .grad lives in the buckets
    assert all(n == got["buckets"] for n in got["launched"])  # every bucket's all-reduce was in flight before backward returned
    assert_close(got["loss"], ref["loss"], 1e-5, what="global loss")
    for i, (a, b) in enumerate(zip(got["g_net"], ref["g_net"])):
        assert float(b.abs().max()) > 0
        assert_close(a, b, 1e-4, abs_=1e-7 * float(b.abs().max()), what=f"shared parameter {i} (sum over ranks)")
    lo, hi = got["frames"]
    assert_close(got["g_depth"], ref["g_depth"][lo : hi + 1], 1e-4, abs_=1e-9, what="explicit depth (halo summed)")
    assert abs(float(got["g_focal"]) - float(ref["g_focal"])) <= 1e-4 * abs(float(ref["g_focal"])) + 1e-6 * abs(float(ref["loss"]))


def _adam_worker(rank, world, port, f, h, w, points, steps, out_path, ghost=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import flowmap_amd
    from flowmap_amd import Batch, Flows, FusedAdam, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard
    from helpers import build_host_sim, to_flows
    from oracle import flowmap_oracle as orc

    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.set_lazy_surfaces(True)
    sc = orc.synth_scene(f, h, w, seed=21)
    flows = to_flows(sc["flows"], "cpu")
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    nf = hi - lo + 1
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.9),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=nf, image_shape=(h, w))
    model.backbone.depth.data = sc["depth_init"][lo : hi + 1].clone()
    model.backbone.weights.data = (0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(21)))[a:b].clone()
    local = Flows(flows.forward[:, a:b].contiguous(), flows.backward[:, a:b].contiguous(),
                  flows.forward_mask[:, a:b].contiguous(), flows.backward_mask[:, a:b].contiguous())
    batch = Batch(torch.zeros((1, nf, 3, h, w)))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist if world > 1 else None)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    if world > 1:
        optimizer = FusedAdam(model.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
    else:
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(steps):
        optimizer.zero_grad(set_to_none=True)
        out = model(batch, local, step)
        loss = loss_fn(batch, local, None, out, step)
        loss.backward()
        losses.append(float(shard.sync(loss, [model.intrinsics.focal_length], model.backbone.depth)))
        optimizer.step()
        if ghost and world > 1 and step == 1:  # (the Procrustes plan exists: the static pixel lists can be exchanged)
            prev = (flows.backward[0, a - 1], flows.backward_mask[0, a - 1]) if rank > 0 else None
            nxt = (flows.forward[0, b], flows.forward_mask[0, b]) if rank < world - 1 else None
            assert shard.enable_ghost_halo(model.backbone.depth, prev, nxt) is True
    torch.save({"depth": model.backbone.depth.detach().clone(), "weights": model.backbone.weights.detach().clone(),
                "focal": model.intrinsics.focal_length.detach().clone(), "losses": losses, "frames": (lo, hi), "pairs": (a, b),
                "in_pass": getattr(optimizer, "counters", {}).get("in_pass_updates", 0), "ghost": shard.ghost_evaluations}, f"{out_path}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("ghost", [False, True], ids=["one-shot halo", "ghost halo"])
def test_in_pass_adam_on_frame_shards_follows_the_unsharded_optimiser(tmp_path, ghost):
    """FusedAdam.fuse_depth_update on a frame shard: interior frames are updated inside the flow pass, the frames shared
    with a neighbour densely after the halo exchange — over several steps the parameters of every rank follow
    torch.optim.Adam on the unsharded video (both copies of a shared frame included)."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close

    f, h, w, points, steps, world = 9, 24, 32, 60, 6, 3
    out, single = str(tmp_path / "shard"), str(tmp_path / "single")
    mp.spawn(_adam_worker, args=(world, _free_port(), f, h, w, points, steps, out, ghost), nprocs=world, join=True)
    mp.spawn(_adam_worker, args=(1, _free_port(), f, h, w, points, steps, single), nprocs=1, join=True)
    ref = torch.load(f"{single}.0")
    for rank in range(world):
        got = torch.load(f"{out}.{rank}")
        lo, hi = got["frames"]
        a, b = got["pairs"]
        assert got["in_pass"] == steps - 1  # every step after the plan exists
        assert got["ghost"] == (steps - 2 if ghost else 0)  # (switched on after the second step)
        for s_, (x, y) in enumerate(zip(got["losses"], ref["losses"])):
            assert abs(x - y) <= 2e-5 * abs(y), (rank, s_, x, y)
        assert_close(got["depth"], ref["depth"][lo : hi + 1], 2e-6, abs_=2e-6, what=f"depth parameters of rank {rank}")
        assert_close(got["weights"], ref["weights"][a:b], 2e-6, abs_=2e-6, what=f"weight logits of rank {rank}")
        assert abs(float(got["focal"]) - float(ref["focal"])) <= 2e-6
    assert float((ref["depth"] - torch.load(f"{single}.0")["depth"]).abs().max()) == 0.0


def _early_worker(rank, world, port, f, h, w, points, with_tracks, out_path, ghost=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import flowmap_amd
    from flowmap_amd import Batch, Flows, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard
    from helpers import build_host_sim, to_tracks
    from oracle import flowmap_oracle as orc

    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    nf = hi - lo + 1
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=nf, image_shape=(h, w))
    model.backbone.depth.data = depth[lo : hi + 1].clone()
    model.backbone.weights.data = wlogit[a:b].clone()
    local = Flows(flows.forward[:, a:b].contiguous(), flows.backward[:, a:b].contiguous(),
                  flows.forward_mask[:, a:b].contiguous(), flows.backward_mask[:, a:b].contiguous())
    batch = Batch(torch.zeros((1, nf, 3, h, w)))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    tracks = to_tracks(orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5), "cpu") if with_tracks else None
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    assert shard.enable_early_halo(model.backbone.depth) is False  # nothing is planned yet: the one-shot exchange stays
    history, modes = [], []
    for step in range(5):
        model.zero_grad(set_to_none=True)
        out = model(batch, local, 0)
        loss = loss_fn(batch, local, None, out, 0)
        tracked = None
        if with_tracks:
            tracked = shard.tracking_loss(track_fn, tracks, out, f - 1)
            (loss + tracked).backward()
        else:
            loss.backward()
        modes.append(shard._early is not None and shard._halo is not None and shard._halo[5])  # the sparse rest is what is in flight
        total = shard.sync(loss, [model.intrinsics.focal_length], model.backbone.depth, already_global=tracked)
        history.append((total.clone(), model.backbone.depth.grad.clone(), model.intrinsics.focal_length.grad.clone()))
        if step == 1 and not ghost:
            assert shard.enable_early_halo(model.backbone.depth) is True  # (collective: every rank calls it here)
        if step == 1 and ghost:  # the ghost halo: the neighbouring pairs' constant flows, handed over once
            prev = (flows.backward[0, a - 1], flows.backward_mask[0, a - 1]) if rank > 0 else None
            nxt = (flows.forward[0, b], flows.forward_mask[0, b]) if rank < world - 1 else None
            assert shard.enable_ghost_halo(model.backbone.depth, prev, nxt) is True
    assert shard.ghost_evaluations == (3 if ghost else 0)
    torch.save({"history": history, "modes": modes, "frames": (lo, hi)}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("with_tracks", [False, True], ids=["flow", "flow+tracking"])
def test_ghost_halo_gives_the_same_gradients(tmp_path, with_tracks, world):
    """FrameShard.enable_ghost_halo(): the neighbour's dense part of a shared frame's gradient is EVALUATED from the 64-byte pose it sends
    (fm_flow_ghost_terms) instead of received as a frame; the sparse rest travels after backward — loss and gradients of every step as
    with the one-shot exchange and as the unsharded oracle."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 9, 12, 16, 40
    out = str(tmp_path / "ghost")
    mp.spawn(_early_worker, args=(world, _free_port(), f, h, w, points, with_tracks, out, True), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    otracks = orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5) if with_tracks else None
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float32)
    for rank in range(world):
        got = torch.load(f"{out}.{rank}")
        assert got["modes"] == [False, False, True, True, True], got["modes"]
        lo, hi = got["frames"]
        for step, (total, g_depth, g_focal) in enumerate(got["history"]):
            assert_close(total, ref["total"], 1e-5, what=f"global loss, step {step}")
            _shard_close(g_depth, ref["g_depth"][lo : hi + 1], ref32["g_depth"][lo : hi + 1], f"g_depth of rank {rank} (ghost halo), step {step}")
            _focal_close(g_focal, ref, ref32)
        for ghost_step in (2, 3, 4):  # against the one-shot exchange of step 1: the same terms, the dense one evaluated here instead of there
            assert_close(got["history"][ghost_step][1], got["history"][1][1], 1e-5, abs_=1e-9, what=f"ghost vs one-shot exchange, rank {rank}")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("with_tracks", [False, True], ids=["flow", "flow+tracking"])
def test_early_halo_exchange_gives_the_same_gradients(tmp_path, with_tracks):
    """FrameShard.enable_early_halo(): the boundary frames' dense gradient is sent when the flow loss's forward pass ends, the sparse
    rest (Procrustes / track pixels) after backward — loss and gradients of every step as with the one-shot exchange and as unsharded."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points, world = 8, 12, 16, 40, 3
    out = str(tmp_path / "early")
    mp.spawn(_early_worker, args=(world, _free_port(), f, h, w, points, with_tracks, out), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    otracks = orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5) if with_tracks else None
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float32)
    for rank in range(world):
        got = torch.load(f"{out}.{rank}")
        assert got["modes"] == [False, False, True, True, True], got["modes"]
        lo, hi = got["frames"]
        for step, (total, g_depth, g_focal) in enumerate(got["history"]):
            assert_close(total, ref["total"], 1e-5, what=f"global loss, step {step}")
            _shard_close(g_depth, ref["g_depth"][lo : hi + 1], ref32["g_depth"][lo : hi + 1], f"g_depth of rank {rank} (halo summed), step {step}")
            _focal_close(g_focal, ref, ref32)
        for early_step in (2, 3, 4):  # against the one-shot exchange of step 1: the same sums in another order
            assert_close(got["history"][early_step][1], got["history"][1][1], 1e-5, abs_=1e-9, what=f"early vs one-shot exchange, rank {rank}")


def _kept_grad_worker(rank, world, port, f, h, w, points, out_path):
    """Steps that KEEP their gradient tensors (optimizer.zero_grad(set_to_none=False)): after the first sync() every shared parameter's .grad
    is the view of its slot in the persistent reduction buffer, zeroed in place and accumulated into by the next backward."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import flowmap_amd
    from flowmap_amd import Batch, Flows, _lib
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard
    from helpers import build_host_sim
    from oracle import flowmap_oracle as orc

    _lib.set_library_for_testing(build_host_sim())
    flowmap_amd.set_lazy_surfaces(True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    nf = hi - lo + 1
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=nf, image_shape=(h, w))
    model.backbone.depth.data = depth[lo : hi + 1].clone()
    model.backbone.weights.data = wlogit[a:b].clone()
    local = Flows(*(x[:, a:b].contiguous() for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
    batch = Batch(torch.zeros((1, nf, 3, h, w)))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    history, aliased = [], []
    focal = model.intrinsics.focal_length
    for step in range(4):
        model.zero_grad(set_to_none=False)
        loss = loss_fn(batch, local, None, model(batch, local, 0), 0)
        loss.backward()
        aliased.append(shard._packed is not None and focal.grad is not None and focal.grad.data_ptr() == shard._packed[2][0].data_ptr())
        total = shard.sync(loss, [focal], model.backbone.depth)
        history.append((total.clone(), model.backbone.depth.grad.clone(), focal.grad.clone()))
    # an in-place edit of dL/ddepth between backward (whose hook posted the halo exchange) and sync(): refused, not exchanged twice
    model.zero_grad(set_to_none=False)
    loss = loss_fn(batch, local, None, model(batch, local, 0), 0)
    loss.backward()
    model.backbone.depth.grad.mul_(1.0)
    refused = False
    try:
        shard.sync(loss, [focal], model.backbone.depth)
    except RuntimeError as exc:
        refused = "modified in place" in str(exc)
    shard.finish_halo_exchange()  # (complete what was posted: every rank does)
    torch.save({"history": history, "aliased": aliased, "refused": refused, "frames": (lo, hi)}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sync_with_gradients_kept_across_steps(tmp_path):
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points, world = 7, 12, 16, 40, 2
    out = str(tmp_path / "kept")
    mp.spawn(_kept_grad_worker, args=(world, _free_port(), f, h, w, points, out), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=torch.float32)
    for rank in range(world):
        got = torch.load(f"{out}.{rank}")
        assert got["aliased"] == [False, True, True, True], got["aliased"]  # from the second step on the gradient IS its slot of the buffer
        assert got["refused"]
        lo, hi = got["frames"]
        for step, (total, g_depth, g_focal) in enumerate(got["history"]):
            assert_close(total, ref["total"], 1e-5, what=f"global loss, step {step}")
            assert_close(g_depth, ref["g_depth"][lo : hi + 1], 1e-4, what=f"g_depth of rank {rank}, step {step}")
            _focal_close(g_focal, ref, ref32)


# ---- round 6: the same checks with the HIP library on every rank — the ranks share the one GPU of a gpurun box and meet over gloo ----


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("backend,world,with_tracks", [("nccl", 2, False), ("nccl", 3, False), ("nccl", 2, True), ("nccl", 3, True), ("nccl", 4, True), ("gloo", 2, False),
                                                       ("gloo", 3, True)])
def test_real_ranks_on_the_gpu_match_the_unsharded_oracle(tmp_path, world, with_tracks, backend):
    """FrameShard between REAL ranks running this package's kernels on cuda:0, which they share — over RCCL (every rank a host of its own: the socket
    transport on loopback, RCCL's own point-to-point / collective code and stream semantics) and over gloo (GPU tensors staged through the host): loss,
    dL/dfocal, every rank's slice of dL/ddepth INCLUDING the halo frames summed across the border, dL/dweights — against the unsharded fp64 oracle, gates
    as on the host double.  What this leaves untested is xGMI."""
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 9, 24, 32, 60
    out = str(tmp_path / "shard")
    mp.spawn(_worker, args=(world, _free_port(), f, h, w, points, out, with_tracks, False, "cuda:0", backend), nprocs=world, join=True)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=9)
    otracks = orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5) if with_tracks else None
    ref = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, otracks, dtype=torch.float32)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    assert sorted(r["frames"] for r in res) == [shard_frames(pr) for pr in shard_pairs(f - 1, world)]
    for r in res:
        if with_tracks:
            assert float(ref["loss_tracking"]) > 0
            assert_close(r["track"], ref["loss_tracking"], 1e-5, what="global tracking loss")
        assert_close(r["loss"], ref["total"], 1e-5, what="global loss")
        _focal_close(r["g_focal"], ref, ref32)
        lo, hi = r["frames"]
        a, b = r["pairs"]
        _shard_close(r["g_depth"], ref["g_depth"][lo : hi + 1], ref32["g_depth"][lo : hi + 1], "g_depth shard (halo summed)")
        _shard_close(r["g_w"], ref["g_wlogit"][a:b], ref32["g_wlogit"][a:b], "g_wlogit shard")
