"""The reference path's op sequence (the oracle = its PyTorch restatement) executed by stock
PyTorch-ROCm on the same MI355X: the informative third column next to the CPU baseline and
flowmap_amd.  Run through gpurun; prints one JSON line.

    python tests/tools/torch_gpu_reference_ops.py [--frames 150 --height 720 --width 1280 --iters 5]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from oracle import flowmap_oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--points", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=0,
                    help="frames per call: the video evaluated as consecutive windows of this many frames (sharing their boundary frame), one "
                         "forward + backward each.  Round 4: the whole-video call aborts with a GPU memory-access fault from 32 frames @ 720p on "
                         "(profiles/r04_stock_pytorch_rocm_fault_32_and_150_frames.txt; 16 frames run), so the 150-frame figure is the sum of its 16-frame windows")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    f, h, w = args.frames, args.height, args.width
    g = torch.Generator(device=dev).manual_seed(1)
    depth = (1.10 + 0.05 * torch.rand((f, h, w), device=dev, generator=g)).requires_grad_(True)
    wlogit = (0.01 * torch.randn((f - 1, h, w), device=dev, generator=g)).requires_grad_(True)
    focal = torch.tensor(0.85, device=dev, requires_grad=True)
    flows = orc.OFlows(0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g), 0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g),
                       torch.rand((1, f - 1, h, w), device=dev, generator=g), torch.rand((1, f - 1, h, w), device=dev, generator=g))

    def step():
        for p in (depth, wlogit, focal):
            p.grad = None
        if args.chunk <= 1 or args.chunk >= f:
            total, _, _ = orc.explicit_depth_step(depth, wlogit, focal, flows, (h, w), num_points=args.points)
            total.backward()
            return total
        value = 0.0
        for lo in range(0, f - 1, args.chunk - 1):  # windows [lo, lo + chunk) share their boundary frame: every pair once
            hi = min(lo + args.chunk, f)
            part = orc.OFlows(*(x[:, lo : hi - 1] for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
            total, _, _ = orc.explicit_depth_step(depth[lo:hi], wlogit[lo : hi - 1], focal, part, (h, w), num_points=args.points)
            total.backward()
            value = value + total.detach()
        return value

    step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print(json.dumps({"what": "reference op sequence (oracle) on stock PyTorch-ROCm, same GPU, fwd+bwd", "frames": f, "height": h, "width": w,
                      "procrustes_points": args.points, "frames_per_call": args.chunk if 1 < args.chunk < f else f, "ms_per_step": dt * 1e3, "iters_per_sec": 1.0 / dt,
                      "peak_memory_gb": torch.cuda.max_memory_allocated() / 1e9, "loss": float(loss.detach())}))


if __name__ == "__main__":
    main()
