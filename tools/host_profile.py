"""cProfile of the HOST side of one bench step at a launch-bound size (run through gpurun)."""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import flowmap_amd  # noqa: E402
from flowmap_amd import Batch  # noqa: E402
from flowmap_amd.loss import LossFlow, LossFlowCfg  # noqa: E402
from flowmap_amd.loss.mapping import MappingHuberCfg  # noqa: E402
from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg  # noqa: E402

dev = torch.device("cuda", 0)
f, h, w = 16, 720, 1280
flowmap_amd.set_lazy_surfaces(True)
depth, wlogit, flows, _ = bench.make_iid(f, h, w, dev, 1)
model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                       ExtrinsicsProcrustesCfg("procrustes", 1000, False)), num_frames=f, image_shape=(h, w)).to(dev)
model.backbone.depth.data = depth
model.backbone.weights.data = wlogit
batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))


def step():
    model.zero_grad(set_to_none=True)
    out = model(batch, flows, 0)
    loss = loss_fn(batch, flows, None, out, 0)
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
