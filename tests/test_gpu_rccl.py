"""Frame-pair sharding over RCCL on the GPUs of one node: flow + tracking losses with the video split over
``torch.cuda.device_count()`` ranks (one process per GPU, backend "nccl" = RCCL over xGMI) against the unsharded
fp64 oracle — the multi-GPU twin of tests/test_sharding_gloo.py.  Skips itself on a box with fewer than two GPUs
(the 1-GPU gpurun boxes); the first multi-GPU lease validates the RCCL path.  The very same worker also runs over gloo on
the host test double (CPU, two ranks) so that the test's own logic is exercised before that day."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, f, h, w, points, out_path, on_gpu=True):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    if on_gpu:
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
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
    shard = FrameShard(rank, world, dist)
    shard.prepare_flow_loss(loss_fn, local)
    shard.prepare_model(model)
    for _ in range(3):  # step 1 atomics, step 2 builds the static plans, step 3 runs on them
        model.zero_grad(set_to_none=True)
        out = model(batch, local, 0)
        loss = loss_fn(batch, local, None, out, 0)
        tracked = shard.tracking_loss(track_fn, tracks, out, f - 1)
        (loss + tracked).backward()
        total = shard.sync(loss, [model.intrinsics.focal_length], model.backbone.depth, already_global=tracked)
    torch.save({"loss": total.cpu(), "g_focal": model.intrinsics.focal_length.grad.cpu(), "g_depth": model.backbone.depth.grad.cpu(),
                "g_w": model.backbone.weights.grad.cpu(), "frames": (lo, hi), "pairs": (a, b)}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _run_and_compare(tmp_path, world, on_gpu):
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import assert_close
    from helpers import run_oracle
    from oracle import flowmap_oracle as orc

    f, h, w, points = 2 * world + 3, 48, 64, 200
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "rccl")
    mp.spawn(_worker, args=(world, port, f, h, w, points, out, on_gpu), nprocs=world, join=True)
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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_sharded_step_matches_unsharded_oracle(tmp_path):
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip(f"needs >= 2 GPUs for an RCCL run (this box has {world})")
    _run_and_compare(tmp_path, min(world, 8), on_gpu=True)


@pytest.mark.timeout(600)
def test_the_same_worker_over_gloo_on_the_host_double(tmp_path):
    _run_and_compare(tmp_path, 2, on_gpu=False)
