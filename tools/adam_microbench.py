"""Time fm_adam_step on the C1 parameter shapes (run through gpurun).

    python tools/adam_microbench.py [--frames 150 --height 720 --width 1280 --iters 20]
Prints one JSON line: ms and GB/s for the dense-gradient tensor (depth), the sparse-gradient
tensor (weight logits: P points per pair), both through FusedAdam, and torch.optim.Adam
(foreach default and fused=True) on the same tensors.
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import flowmap_amd  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--points", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", action="store_true", help="also time every build_variants/libfm_adam_*.so (fm_adam_step only)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    f, h, w = args.frames, args.height, args.width
    g = torch.Generator(device=dev).manual_seed(0)
    out = {}
    for name, shape, sparse in (("depth", (f, h, w), False), ("weights", (f - 1, h, w), True)):
        grad = torch.randn(shape, device=dev, generator=g) * 1e-3
        if sparse:
            idx = torch.linspace(0, h * w - 1, args.points, device=dev).long()
            keep = torch.zeros((h * w,), device=dev)
            keep[idx] = 1
            grad = grad * keep.reshape(1, h, w)
        n = grad.numel()
        for label, make in (("fused_hip", lambda p: flowmap_amd.FusedAdam([p], lr=3e-5)),
                            ("torch_foreach", lambda p: torch.optim.Adam([p], lr=3e-5)),
                            ("torch_fused", lambda p: torch.optim.Adam([p], lr=3e-5, fused=True))):
            p = torch.randn(shape, device=dev, generator=g).requires_grad_(True)
            p.grad = grad
            opt = make(p)
            ms = timed(opt.step, args.iters)
            out[f"{name}_{label}"] = {"ms": ms, "GBps_at_28B": 28 * n / ms / 1e6}
            del opt, p
    if args.variants:
        import ctypes
        import glob

        from flowmap_amd import _lib

        shape = (f, h, w)
        grad = torch.randn(shape, device=dev, generator=g) * 1e-3
        p, m, v = (torch.randn(shape, device=dev, generator=g).abs() for _ in range(3))
        libs = {"shipped": str(_lib.LIB_PATH)}
        libs.update({Path(q).stem.replace("libfm_", ""): q for q in sorted(glob.glob(str(Path(__file__).resolve().parent.parent / "build_variants" / "libfm_adam_*.so")))})
        for name, path in libs.items():
            fn = ctypes.CDLL(path).fm_adam_step
            fn.argtypes = _lib.SIGNATURES["fm_adam_step"]
            fn.restype = ctypes.c_int
            st = torch.cuda.current_stream().cuda_stream
            ms = timed(lambda: fn(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 3, 3e-5, 0.9, 0.999, 1e-8, 0.0, st), args.iters)
            out[f"variant_{name}"] = {"ms": ms, "GBps_at_28B": 28 * p.numel() / ms / 1e6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
