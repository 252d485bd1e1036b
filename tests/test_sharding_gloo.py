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


def _focal_close(got, ref):
    """dL/dfocal as helpers.compare_step judges it: 1e-4 of itself or a few fp32 roundings of the cancelling terms it sums."""
    from helpers import FOCAL_ULPS

    err = abs(float(got) - float(ref["g_focal"]))
    bound = max(1e-4 * abs(float(ref["g_focal"])), FOCAL_ULPS * 2.0**-24 * ref["g_focal_terms"])
    assert err <= bound, f"g_focal: {float(got):.6e} vs {float(ref['g_focal']):.6e} (bound {bound:.2e})"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, f, h, w, points, out_path, with_tracks=False, softmin=False):
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
    from helpers import build_host_sim
    from oracle import flowmap_oracle as orc

    _lib.set_library_for_testing(build_host_sim())
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
    local = Flows(flows.forward[:, a:b].contiguous(), flows.backward[:, a:b].contiguous(),
                  flows.forward_mask[:, a:b].contiguous(), flows.backward_mask[:, a:b].contiguous())
    batch = Batch(torch.zeros((1, nf, 3, h, w)))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    shard = FrameShard(rank, world, dist)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)  # softmin sweep on rank 0 + broadcast; halo exchange from the depth gradient's hook
    out = model(batch, local, 0)
    loss = loss_fn(batch, local, None, out, 0)
    track_total = None
    if with_tracks:
        from helpers import to_tracks

        tracks = to_tracks(orc.synth_tracks(f, h, w, seed=9, interval=2, radius=3, grid=5), "cpu")  # GLOBAL frame indices
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
        # global value on every rank; its autograd gradients are this rank's share of the term
        track_total = shard.tracking_loss(track_fn, tracks, out, f - 1)
        (loss + track_total).backward()
    else:
        loss.backward()
    shared = [p for name, p in model.named_parameters() if not name.startswith("backbone.")]  # a LIST of shared parameters
    total = shard.sync(loss, shared, model.backbone.depth, already_global=track_total)
    g_focal = None if softmin else model.intrinsics.focal_length.grad.clone()
    torch.save(
        {"loss": total.clone(), "track": None if track_total is None else track_total.detach().clone(), "g_focal": g_focal, "k": out.intrinsics.detach().clone(),
         "g_depth": model.backbone.depth.grad.clone(),
         "g_w": model.backbone.weights.grad.clone(), "frames": (lo, hi), "pairs": (a, b)},
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
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert_close(r["loss"], ref["total"], 1e-5, what="global loss")
        _focal_close(r["g_focal"], ref)
        lo, hi = r["frames"]
        a, b = r["pairs"]
        assert_close(r["g_depth"], ref["g_depth"][lo : hi + 1], 1e-4, what="g_depth shard (halo summed)")
        assert_close(r["g_w"], ref["g_wlogit"][a:b], 3e-4, what="g_wlogit shard")


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
    assert float(ref["loss_tracking"]) > 0
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert_close(r["track"], ref["loss_tracking"], 1e-5, what="global tracking loss")
        assert_close(r["loss"], ref["total"], 1e-5, what="global loss")
        _focal_close(r["g_focal"], ref)
        lo, hi = r["frames"]
        a, b = r["pairs"]
        assert_close(r["g_depth"], ref["g_depth"][lo : hi + 1], 2e-4, what="g_depth shard (halo summed)")
        assert_close(r["g_w"], ref["g_wlogit"][a:b], 5e-4, what="g_wlogit shard")


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
