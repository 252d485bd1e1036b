"""How long does the host need to ENQUEUE one bench step (no device sync)?  If this approaches the
GPU time per step the loop is launch-bound and a hipGraph capture would pay; run through gpurun."""
import json
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
f, h, w = 150, 720, 1280
flowmap_amd.set_lazy_surfaces(True)
depth, wlogit, flows = bench.make_inputs(f, h, w, dev, seed=1)
model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                       ExtrinsicsProcrustesCfg("procrustes", 1000, False)), num_frames=f, image_shape=(h, w)).to(dev)
model.backbone.depth.data = depth
model.backbone.weights.data = wlogit
batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
tracks = bench.make_tracks(f, dev, seed=100)
from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg  # noqa: E402

soft = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsSoftminCfg("softmin", 8192, 0.5, 2.0, 60, RegressionCfg(1000, 100)),
                      ExtrinsicsProcrustesCfg("procrustes", 1000, False)), num_frames=f, image_shape=(h, w)).to(dev)
soft.backbone.depth.data = depth
soft.backbone.weights.data = wlogit
opt = flowmap_amd.FusedAdam(model.parameters(), lr=3e-5)
out = {}
for name, with_tracks, with_opt, use_soft in (("flow", False, False, False), ("flow+tracking", True, False, False),
                                                ("flow+tracking+adam", True, True, False), ("flow, softmin intrinsics", False, False, True)):
    model_used = soft if use_soft else model
    def step():
        model_used.zero_grad(set_to_none=True)
        o = model_used(batch, flows, 0)
        loss = loss_fn(batch, flows, None, o, 0)
        if with_tracks:
            loss = loss + track_fn(batch, flows, tracks, o, 0)
        loss.backward()
        if with_opt:
            opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    t_enqueue = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 50
    out[name] = {"host_enqueue_ms_per_step": t_enqueue * 1e3, "wall_ms_per_step": t_total * 1e3}
print(json.dumps(out))
