"""Frame-pair sharding over RCCL on the GPUs of one node: flow + tracking losses with the video split over
``torch.cuda.device_count()`` ranks (one process per GPU, backend "nccl" = RCCL over xGMI) against the unsharded
fp64 oracle — the multi-GPU twin of tests/test_sharding_gloo.py.  On a box with fewer than two GPUs (the 1-GPU gpurun boxes)
three RCCL ranks share the one device (round 6: a host id per rank, RCCL's socket transport); the first multi-GPU lease adds xGMI.  The very same worker also runs over gloo on
the host test double (CPU, two ranks) so that the test's own logic is exercised before that day."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, f, h, w, points, out_path, on_gpu=True, one_rank_extras=False, one_gpu=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    if on_gpu:
        torch.set_num_threads(4)  # (the scene is synthesised on the host by every rank: on a 256-thread box the ranks' default thread pools fight each other — 100 s instead of 10)
        index = 0 if one_gpu else rank
        if one_gpu and world > 1:
            # RCCL refuses two ranks of one HOST on one device: every rank declares a host of its own and the ranks meet over RCCL's socket transport on
            # loopback (tools/probes/rccl_one_gpu_probe.py) — RCCL's own point-to-point / collective code and stream semantics, everything but xGMI
            os.environ.update(NCCL_HOSTID=f"flowmap-amd-test-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1", NCCL_SHM_DISABLE="1")
        torch.cuda.set_device(index)
        dev = torch.device("cuda", index)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:  # the same code over gloo on the host test double
        from flowmap_amd import _lib
        from helpers import build_host_sim

        torch.set_num_threads(2)
        _lib.set_library_for_testing(build_host_sim())
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import flowmap_amd
    from flowmap_amd import Batch, Flows
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard, shard_frames, shard_pairs
    from helpers import to_tracks
    from oracle import flowmap_oracle as orc

    flowmap_amd.set_lazy_surfaces(True)
    sc = orc.synth_scene(f, h, w, seed=5)
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(5))
    a, b = shard_pairs(f - 1, world)[rank]
    lo, hi = shard_frames((a, b))
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.8),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=hi - lo + 1, image_shape=(h, w))
    model.backbone.depth.data = sc["depth_init"][lo : hi + 1].clone()
    model.backbone.weights.data = wl[a:b].clone()
    model = model.to(dev)
    fl = sc["flows"]
    local = Flows(*(x[:, a:b].contiguous().to(dev) for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
    batch = Batch(torch.zeros((1, hi - lo + 1, 3, h, w), device=dev))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
    tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=5, interval=3, radius=5, grid=8), dev)
    shard = FrameShard(rank, world, dist, force_collectives=world == 1)  # (one rank: every collective still executes, on a one-member group)
    assert shard.active
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    for _ in range(3):  # step 1 atomics, step 2 builds the static plans, step 3 runs on them
        model.zero_grad(set_to_none=True)
        out = model(batch, local, 0)
        loss = loss_fn(batch, local, None, out, 0)
        tracked = shard.tracking_loss(track_fn, tracks, out, f - 1)
        (loss + tracked).backward()
        total = shard.sync(loss, [model.intrinsics.focal_length], model.backbone.depth, already_global=tracked)
    result = {"loss": total.cpu(), "g_focal": model.intrinsics.focal_length.grad.cpu(), "g_depth": model.backbone.depth.grad.cpu(),
              "g_w": model.backbone.weights.grad.cpu(), "frames": (lo, hi), "pairs": (a, b)}
    if one_rank_extras:
        result.update(_one_rank_extras(dist, dev, sc, wl, f, h, w, points, on_gpu))
    torch.save(result, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _one_rank_extras(dist, dev, sc, wl, f, h, w, points, on_gpu):
    """The rest of the sharded step's communication code on the same one-member group: an interior rank's SHARE of a three-rank run
    (FrameShard(proxy=True), what bench.py --share 3 runs) with each halo form — one shot, early, ghost (fm_halo_ghost_begin / fm_flow_ghost_terms /
    fm_halo_delta_sparse / fm_halo_scatter) — and, on the GPU, the ghost form replayed as hipGraphs (GraphedShardedStep, collectives eager).
    Returned per form: the share's loss and the gradients of its interior frames / its pairs' weights, which no halo touches and which the
    caller holds against the unsharded oracle (scaled by the share's own normaliser)."""
    import flowmap_amd
    from flowmap_amd import Batch, Flows
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from flowmap_amd.sharding import FrameShard, shard_frames, shard_pairs

    share_of, share_rank = 3, 1
    a, b = shard_pairs(f - 1, share_of)[share_rank]
    lo, hi = shard_frames((a, b))
    fl = sc["flows"]
    out = {"share": (a, b, lo, hi)}
    for form in ("oneshot", "early", "ghost") + (("ghost_graphed",) if on_gpu else ()):
        model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.8),
                               ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=hi - lo + 1, image_shape=(h, w))
        model.backbone.depth.data = sc["depth_init"][lo : hi + 1].clone()
        model.backbone.weights.data = wl[a:b].clone()
        model = model.to(dev)
        local = Flows(*(x[:, a:b].contiguous().to(dev) for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
        batch = Batch(torch.zeros((1, hi - lo + 1, 3, h, w), device=dev))
        loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
        shard = FrameShard(share_rank, share_of, dist, proxy=True)
        shard.prepare_flow_loss(loss_fn, local)
        shard.prepare_model(model)
        shared = [model.intrinsics.focal_length]

        def forward_only():
            model.zero_grad(set_to_none=True)
            return loss_fn(batch, local, None, model(batch, local, 0), 0)

        def step():
            loss = forward_only()
            loss.backward()
            return shard.sync(loss, shared, model.backbone.depth)

        for _ in range(3):
            step()
        if form == "early":
            assert shard.enable_early_halo(model.backbone.depth)
        elif form.startswith("ghost"):
            assert shard.enable_ghost_halo(model.backbone.depth)
            shard.set_proxy_ghost_flows(local)
        if form == "ghost_graphed":
            step = flowmap_amd.GraphedShardedStep(forward_only, shard, shared, model.backbone.depth, warmup=3)
        for _ in range(2):
            total = step()
        if form.startswith("ghost"):
            assert shard.ghost_evaluations > 0
        out[form] = {"loss": total.detach().cpu(), "valid": float(local.forward_mask.sum() + local.backward_mask.sum()),
                     "g_depth_interior": model.backbone.depth.grad[1:-1].detach().cpu(), "g_w": model.backbone.weights.grad.detach().cpu()}
    return out


def _run_and_compare(tmp_path, world, on_gpu, one_gpu=False):
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 2 * world + 3, 48, 64, 200
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "rccl")
    mp.spawn(_worker, args=(world, port, f, h, w, points, out, on_gpu, world == 1, one_gpu), nprocs=world, join=True)
    sc = orc.synth_scene(f, h, w, seed=5)
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(5))
    tracks = orc.synth_tracks(f, h, w, scene=sc, seed=5, interval=3, radius=5, grid=8)
    ref = run_oracle(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), points, tracks, dtype=torch.float64)
    for r in (torch.load(f"{out}.{rank}") for rank in range(world)):
        lo, hi = r["frames"]
        a, b = r["pairs"]
        assert_close(r["loss"], ref["total"], 1e-4, what="global loss")
        assert_close(r["g_focal"], ref["g_focal"], 1e-4, what="g_focal (all-reduced)")
        assert_close(r["g_depth"], ref["g_depth"][lo : hi + 1], 1e-4, what="g_depth shard (halo summed)")
        assert_close(r["g_w"], ref["g_wlogit"][a:b], 1e-4, what="g_wlogit shard")
    if world == 1:
        # the share of an interior rank of three, per halo form: the gradients no halo touches are the whole video's flow-loss gradients times
        # (global Σmask / the share's Σmask) — a share on a one-member group normalises by its own masks
        flow_only = run_oracle(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), points, None, dtype=torch.float64)
        valid_all = float(sc["flows"].forward_mask.sum() + sc["flows"].backward_mask.sum())
        a, b, lo, hi = r["share"]
        forms = [k for k in r if k in ("oneshot", "early", "ghost", "ghost_graphed")]
        assert forms[:3] == ["oneshot", "early", "ghost"] and (not on_gpu or "ghost_graphed" in forms)
        for form in forms:
            ratio = valid_all / r[form]["valid"]
            assert_close(r[form]["g_depth_interior"], flow_only["g_depth"][lo + 1 : hi] * ratio, 1e-4, what=f"{form}: interior frames' dL/ddepth")
            assert_close(r[form]["g_w"], flow_only["g_wlogit"][a:b] * ratio, 1e-4, what=f"{form}: dL/dweights of the share's pairs")
            assert_close(r[form]["loss"], r["oneshot"]["loss"], 1e-5, what=f"{form}: loss vs the one-shot form")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_sharded_step_matches_unsharded_oracle(tmp_path):
    world = torch.cuda.device_count()
    if world < 2:
        # round 6: a one-GPU box no longer skips — three RCCL ranks share the device (a host id each, the socket transport)
        _run_and_compare(tmp_path, 3, on_gpu=True, one_gpu=True)
        return
    _run_and_compare(tmp_path, min(world, 8), on_gpu=True)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_every_collective_call_site_runs_on_a_one_rank_rccl_communicator(tmp_path):
    """the driver's GPU boxes have ONE GPU, so the >= 2-GPU test above skips there and no `nccl` code would run in the GPU
    suite.  The same worker with world = 1 and the collectives forced on (FrameShard(force_collectives=True)): the set-up all-reduce of
    Σmask, the packed all-reduce of [loss, shared gradients], the tracking loss's pose all-gather + [Σρ, count] all-reduce + pose-gradient
    all-reduce — each on a real RCCL communicator — against the unsharded fp64 oracle; then an interior rank's share of three on the same
    communicator with the one-shot, early and ghost halo forms and the ghost form replayed as hipGraphs.  The point-to-point exchange itself
    (the share has no peer) is the test above and tests/test_gpu_multirank.py: real RCCL ranks, sharing the one GPU since round 6."""
    _run_and_compare(tmp_path, 1, on_gpu=True)


@pytest.mark.timeout(600)
def test_the_same_worker_over_gloo_on_the_host_double(tmp_path):
    _run_and_compare(tmp_path, 2, on_gpu=False)


@pytest.mark.timeout(600)
def test_the_one_rank_worker_over_gloo_on_the_host_double(tmp_path):
    _run_and_compare(tmp_path, 1, on_gpu=False)
