"""How much the CPU baseline's code (oracle/flowmap_oracle.py — a restatement, `cpu_baseline.kind: "port"`) costs next to the code it stands for:
the IMPORTED reference (dcharatan/flowmap at /root/reference: its own Model + LossFlow, forward + backward) and the oracle's
``explicit_depth_step`` timed on the same inputs, same threads, interleaved.  Build container only (the GPU box has no reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_cpu_calibration.py [--full]      -> tests/golden/cpu_calibration.json

``bench.py`` quotes the ratio next to the port's time it measures on the GPU box's host cores (``cpu_baseline.port_over_reference``,
``reference_equivalent_value``).  TEST INFRASTRUCTURE, like everything under oracle/.
"""

from __future__ import annotations

import argparse
import json
import os
import platform
import statistics
import sys
import time
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("FLOWMAP_REFERENCE", "/root/reference"))
sys.path[:0] = [str(HERE / "refstubs"), str(REF), str(HERE.parent)]

import torch  # noqa: E402

from flowmap.dataset.types import Batch  # noqa: E402
from flowmap.flow.flow_predictor import Flows  # noqa: E402
from flowmap.loss import get_losses  # noqa: E402
from flowmap.loss.loss_flow import LossFlowCfg  # noqa: E402
from flowmap.loss.mapping.mapping_huber import MappingHuberCfg  # noqa: E402
from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg  # noqa: E402
from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg  # noqa: E402
from flowmap.model.model import Model, ModelCfg  # noqa: E402

from oracle import flowmap_oracle as orc  # noqa: E402


def legs(f, h, w, points=1000, focal=0.85):
    """(reference step, port step): forward + backward of the flow loss on the same i.i.d. inputs."""
    depth, wlogit, fl = orc.synth_iid(f, h, w, seed=0)
    cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), IntrinsicsRegressedCfg("regressed", focal),
                   ExtrinsicsProcrustesCfg("procrustes", points, False), True)
    model = Model(cfg, num_frames=f, image_shape=(h, w))  # the reference's Model, unmodified
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    batch = Batch(torch.zeros((1, f, 3, 1, 1)).expand(1, f, 3, h, w), torch.arange(f)[None], ["s"], ["d"])
    flows = Flows(fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)
    (loss_fn,) = get_losses([LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01))])

    def reference():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(batch, flows, None, model(batch, flows, 0), 0)
        loss.backward()
        return float(loss.detach())

    d, wl, fo = depth.clone().requires_grad_(True), wlogit.clone().requires_grad_(True), torch.tensor(focal, requires_grad=True)

    def port():
        d.grad = wl.grad = fo.grad = None
        total, _, _ = orc.explicit_depth_step(d, wl, fo, fl, (h, w), num_points=points)
        total.backward()
        return float(total.detach())

    return reference, port


def measure(f, h, w, rounds):
    reference, port = legs(f, h, w)
    losses = (reference(), port())  # warm-up of both (allocator, page faults)
    t_ref, t_port = [], []
    for _ in range(rounds):  # interleaved: a noisy neighbour on the host hits both legs alike
        t0 = time.perf_counter()
        reference()
        t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        port()
        t_port.append(time.perf_counter() - t0)
    ref_s, port_s = statistics.median(t_ref), statistics.median(t_port)
    return {"frames": f, "height": h, "width": w, "rounds": rounds, "reference_s_per_iter": ref_s, "port_s_per_iter": port_s,
            "port_over_reference": port_s / ref_s, "reference_s_all": t_ref, "port_s_all": t_port,
            "loss_reference": losses[0], "loss_port": losses[1], "loss_rel_diff": abs(losses[0] - losses[1]) / abs(losses[0])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--full", action="store_true", help="also 150 x 720x1280 (the metric's size: ~40 GB and ~40 s per leg and round)")
    ap.add_argument("--out", default=str(HERE.parent / "tests" / "golden" / "cpu_calibration.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    sizes = [(16, 256, 256, 9), (16, 720, 1280, 5)] + ([(150, 720, 1280, 3)] if args.full else [])
    rows = []
    for f, h, w, rounds in sizes:
        rows.append(measure(f, h, w, rounds))
        print(json.dumps({k: v for k, v in rows[-1].items() if not k.endswith("_all")}), flush=True)
    big = rows[-1]
    out = {
        "what": "seconds per forward + backward of the flow loss (explicit depth, regressed intrinsics, Procrustes P = 1000, huber 0.01) on the build "
                "container's host cores: the imported reference (dcharatan/flowmap, its own Model + LossFlow) and oracle/flowmap_oracle.py's "
                "explicit_depth_step on the same i.i.d. inputs, interleaved, medians",
        "made_by": "oracle/make_cpu_calibration.py" + (" --full" if args.full else ""),
        "threads": args.threads, "host": platform.processor() or platform.machine(), "torch": torch.__version__,
        "port_over_reference": big["port_over_reference"], "port_over_reference_measured_at": [big["frames"], big["height"], big["width"]],
        "sizes": rows,
    }
    Path(args.out).write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
