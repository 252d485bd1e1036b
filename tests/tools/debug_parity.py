"""Debug helper (GPU box): run the fused path stage by stage on the GPU and on the host
test double with identical inputs, print the relative difference of every intermediate."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from flowmap_amd import _lib, _ops  # noqa: E402
from helpers import build_host_sim  # noqa: E402
from oracle import flowmap_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def stages(depth, k, weights, flows, idx, dev):
    out = {}
    d = depth.to(dev)[None].contiguous().requires_grad_(True)
    kk = k.to(dev).contiguous().requires_grad_(True)
    w = weights.to(dev)[None].contiguous().requires_grad_(True)
    fl = [x.to(dev).contiguous() for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)]
    i = idx.to(dev)
    out["kinv"] = _ops.intrinsics_inverse(kk.detach())
    rel_b, _ = _ops.ProcrustesFit.apply(d, kk, None, w, fl[1], i)
    out["t_bwd"] = rel_b.detach()
    ext = _ops.PoseChain.apply(rel_b)
    out["ext"] = ext.detach()
    rf, rb = _ops.RelativePoses.apply(ext)
    out["rel_f"], out["rel_b"] = rf.detach(), rb.detach()
    norm = _ops.flow_valid_norm(fl[2], fl[3], 1000.0)
    out["norm"] = norm.detach().clone()
    loss = _ops.FlowLossFused.apply(d, kk, rf, rb, fl[0], fl[1], fl[2], fl[3], norm, 0, 0.01, False, 0)
    out["loss"] = loss.detach()
    loss.backward()
    out["g_depth"], out["g_k"], out["g_w"] = d.grad, kk.grad, w.grad
    return out


def main():
    f, h, w, p = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (4, 720, 1280, 1000)
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=f + h)
    k = orc.focal_to_k(torch.tensor(0.85), (h, w)).expand(1, f, 3, 3).contiguous()
    weights = (100 * wlogit).sigmoid()
    idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64)
    gpu = stages(depth, k, weights, flows, idx, "cuda:0")
    torch.cuda.synchronize()
    _lib.set_library_for_testing(build_host_sim())
    cpu = stages(depth, k, weights, flows, idx, "cpu")
    for key in gpu:
        print(f"{key:8s} gpu-vs-hostsim rel {rel(gpu[key], cpu[key]):.3e}   |gpu| {float(gpu[key].double().norm()):.6e}")
    print("loss gpu", float(gpu["loss"]), "cpu", float(cpu["loss"]), "norm gpu", gpu["norm"].tolist(), "cpu", cpu["norm"].tolist())


if __name__ == "__main__":
    main()
