"""Does the VALU-bound tracking forward hide under the HBM-bound fused flow pass when the two run on different
streams?  (C2 inputs; both are independent once the poses exist.)
    python tools/overlap_probe.py [height width]
"""
import json
import sys
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

dev = torch.device("cuda:0")
f = 150
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
flowmap_amd.set_lazy_surfaces(True)
depth_init, wlogit, flows, scene = bench.make_scene(f, h, w, dev, 0)
tracks = bench.make_tracks(f, dev, 0, scene=scene, hw=(h, w))
model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", 0.85),
                       ExtrinsicsProcrustesCfg("procrustes", 1000, False)), num_frames=f, image_shape=(h, w)).to(dev)
model.backbone.depth.data = depth_init
model.backbone.weights.data = wlogit
batch = Batch(torch.zeros((1, f, 3, 1, 1), device=dev).expand(1, f, 3, h, w))
flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(mode):
    out = model(batch, flows, 0)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    if mode == "flow":
        total = flow_fn(batch, flows, None, out, 0)
    elif mode == "track":
        total = track_fn(batch, flows, tracks, out, 0)
    elif mode == "serial":
        total = flow_fn(batch, flows, None, out, 0) + track_fn(batch, flows, tracks, out, 0)
    else:
        side.wait_stream(main)
        if mode == "overlap_track_first":
            with torch.cuda.stream(side):
                tr = track_fn(batch, flows, tracks, out, 0)
            fl = flow_fn(batch, flows, None, out, 0)
        else:
            fl = flow_fn(batch, flows, None, out, 0)
            with torch.cuda.stream(side):
                tr = track_fn(batch, flows, tracks, out, 0)
        main.wait_stream(side)
        total = fl + tr
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b), float(total)


res = {}
for mode in ("flow", "track", "serial", "overlap", "overlap_track_first"):
    for _ in range(4):
        run(mode)
    ts = [run(mode) for _ in range(20)]
    res[mode] = {"ms_min": round(min(t for t, _ in ts), 4), "ms_median": round(sorted(t for t, _ in ts)[10], 4), "loss": ts[-1][1]}
    print(mode, res[mode], flush=True)
print(json.dumps({"workload": f"{f} x {h} x {w}, 30 segments x 1225 tracks, forward only (gradient kernels)", **res}))
