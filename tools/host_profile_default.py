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

dev = torch.device("cuda", 0)
f, h, w = 150, 180, 240
flowmap_amd.set_lazy_surfaces(True)
depth, wlogit, flows, scene = bench.make_scene(f, h, w, dev, 1)
tracks = bench.make_tracks(f, dev, seed=100, scene=scene, hw=(h, w))
model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.8),
                       ExtrinsicsProcrustesCfg("procrustes", 1000, False)), num_frames=f, image_shape=(h, w)).to(dev)
model.backbone.depth.data = depth
model.backbone.weights.data = wlogit
batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
opt = flowmap_amd.FusedAdam(model.parameters(), lr=3e-5) if "--adam" in sys.argv else None


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
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
host = time.perf_counter() - t0  # (the host's time to ENQUEUE 300 steps: it is ahead of the GPU only if the step is GPU-bound)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"300 steps: host enqueue {host / 300 * 1e3:.3f} ms/step, wall {total / 300 * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(34)
