"""cProfile of the HOST side of one step at the reference's default operating point (150 frames of 180x240, flow + tracking, FusedAdam): the
step is host-bound there (0.49 ms eager for 0.35 ms of kernels).  Run through gpurun."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import flowmap_amd  # noqa: E402
from flowmap_amd import Batch  # noqa: E402
from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg  # noqa: E402
from flowmap_amd.loss.mapping import MappingHuberCfg  # noqa: E402
from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg  # noqa: E402

on_double = "--host-double" in sys.argv  # (the serial CPU build of the kernels: call counts and Python time are real, kernel time is not)
dev = torch.device("cpu") if on_double else torch.device("cuda", 0)
if on_double:
    sys.path.insert(0, str(ROOT / "tests"))
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
f, h, w = (12, 24, 32) if on_double else (150, 180, 240)
flowmap_amd.set_lazy_surfaces(True)
depth, wlogit, flows, scene = bench.make_scene(f, h, w, dev, 1)
tracks = bench.make_tracks(f, dev, seed=100, scene=scene, hw=(h, w), **({"interval": 3, "radius": 3, "grid": 6} if on_double else {}))
if "--installed" in sys.argv:  # the drop-in path: the reference-layout package's own Model / get_losses after install() (bench.py --model installed)
    package = bench.reference_layout_package()
    flowmap_amd.install()
    import dataclasses

    from flowmap.dataset.types import Batch as PackageBatch
    from flowmap.flow.flow_predictor import Flows as PackageFlows
    from flowmap.tracking.track_predictor import Tracks as PackageTracks

    model, (loss_fn, track_fn) = bench.installed_modules((("explicit_depth", 1.0, 100.0), ("regressed", 0.8), ("procrustes", min(1000, h * w // 4), False)), f, (h, w), True)
    model = model.to(dev)
    flows = PackageFlows(flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)
    tracks = [PackageTracks(t.xy, t.visibility, t.start_frame) for t in tracks]
    batch = PackageBatch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w), *([None] * (len(dataclasses.fields(PackageBatch)) - 1)))
    print("installed path:", package)
else:
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.8),
                           ExtrinsicsProcrustesCfg("procrustes", min(1000, h * w // 4), False)), num_frames=f, image_shape=(h, w)).to(dev)
    batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
model.backbone.depth.data = depth
model.backbone.weights.data = wlogit
opt = flowmap_amd.FusedAdam(model.parameters(), lr=3e-5) if "--adam" in sys.argv else None
sync = (lambda: None) if on_double else torch.cuda.synchronize


def step():
    model.zero_grad(set_to_none=True)
    out = model(batch, flows, 0)
    loss = loss_fn(batch, flows, tracks, out, 0) + track_fn(batch, flows, tracks, out, 0)
    loss.backward()
    if opt is not None:
        opt.step()


for _ in range(20):
    step()
flowmap_amd.freeze_gc()
sync()
t0 = time.perf_counter()
for _ in range(300):
    step()
host = time.perf_counter() - t0  # (the host's time to ENQUEUE 300 steps: it is ahead of the GPU only if the step is GPU-bound)
sync()
total = time.perf_counter() - t0
print(f"300 steps: host enqueue {host / 300 * 1e3:.3f} ms/step, wall {total / 300 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
sync()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(34)
