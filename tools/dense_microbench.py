"""A/B timing of the dense Procrustes entry points (fm_procrustes_fit with indices = NULL,
fm_procrustes_scatter_dense) across build variants of fm_procrustes.hip (build_variants/*.so, made by
SRC=fm_procrustes.hip tools/build_variants.sh name:"-DFLAG"), i.i.d. and smooth flows, C1-sized inputs.
    python tools/dense_microbench.py [frames]
"""
import ctypes
import glob
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib  # noqa: E402

dev = "cuda:0"
f = int(sys.argv[1]) if len(sys.argv) > 1 else 150
h, w = 720, 1280
g = torch.Generator(device=dev).manual_seed(0)
depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g)
logit = 0.01 * torch.randn((1, f - 1, h, w), device=dev, generator=g)
flows = {"iid": 0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)}
low = 0.01 * torch.randn((f - 1, 2, h // 40, w // 40), device=dev, generator=g)
flows["smooth"] = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).permute(0, 2, 3, 1)[None].contiguous()
low = 0.01 * torch.randn((f - 1, 2, 3, 5), device=dev, generator=g)  # a few pixels of variation across a tile: what a real camera motion gives
flows["gentle"] = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).permute(0, 2, 3, 1)[None].contiguous()
if len(sys.argv) > 2:
    flows = {k: v for k, v in flows.items() if k in sys.argv[2].split(",")}
fx = 0.85 * (h * w) ** 0.5
k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
kinv = torch.linalg.inv(k).contiguous()
pairs = f - 1
stats = torch.empty((pairs, 16), dtype=torch.float64, device=dev)
t_bwd = torch.empty((1, pairs, 4, 4), device=dev)
t_fwd = torch.empty_like(t_bwd)
aux = torch.empty((pairs, 40), dtype=torch.float64, device=dev)
pair_grad = torch.empty((pairs, 20), dtype=torch.float64, device=dev)
g_t = torch.randn((1, pairs, 4, 4), device=dev, generator=g)
g_depth = torch.zeros_like(depth)
g_w = torch.empty_like(logit)
kinv_acc = torch.zeros((f, 9), dtype=torch.float64, device=dev)
consts = torch.empty((pairs, 40), dtype=torch.float64, device=dev)

libs = {"shipped": str(_lib.LIB_PATH)}
for p in sorted(glob.glob(str(ROOT / "build_variants" / "*.so"))):
    libs[Path(p).stem.replace("libfm_", "")] = p


def bind(path):
    lib = ctypes.CDLL(path)
    for name in ("fm_procrustes_fit", "fm_pose_solve_bwd", "fm_procrustes_dense_tiles", "fm_procrustes_dense_plan", "fm_procrustes_scatter_dense"):
        fn = getattr(lib, name)
        fn.argtypes = _lib.SIGNATURES[name]
        fn.restype = ctypes.c_int
    return lib


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)


P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
out = {}
shipped_stats = {}  # the fit's fp64 moments per flow kind from the shipped library: every variant must reproduce them
for name, path in libs.items():
    lib = bind(path)
    for kind, flow in flows.items():
        tiles = ctypes.c_int(0)
        lib.fm_procrustes_dense_tiles(h, w, ctypes.addressof(tiles))
        counts = torch.zeros((pairs * tiles.value,), dtype=torch.int32, device=dev)
        assert lib.fm_procrustes_dense_plan(P(flow), 1, f, h, w, P(counts), None, None, st) == 0
        first = torch.zeros((counts.numel() + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=first[1:])
        entries = torch.empty((int(first[-1].item()),), dtype=torch.int32, device=dev)
        counts.zero_()
        assert lib.fm_procrustes_dense_plan(P(flow), 1, f, h, w, P(counts), P(first), P(entries), st) == 0

        def fit():
            assert lib.fm_procrustes_fit(P(depth), P(kinv), None, P(flow), P(logit), 100.0, None, h * w, 1, 1, f, h, w, P(stats), P(t_bwd), P(t_fwd), P(aux), st) == 0

        fit()
        if name == "shipped":
            shipped_stats[kind] = stats.clone()
        stats_gap = float(((stats - shipped_stats[kind]).abs().amax(0) / shipped_stats[kind].abs().amax(0).clamp_min(1e-300)).max()) if kind in shipped_stats else None
        assert lib.fm_pose_solve_bwd(P(g_t), None, P(t_bwd), P(aux), pairs, P(pair_grad), None, 0, st) == 0

        def fused():
            assert lib.fm_procrustes_scatter_dense(P(depth), P(kinv), P(flow), P(logit), 100.0, 1, f, h, w, P(aux), P(pair_grad), P(g_depth), P(g_w), None, None, P(consts), st) == 0

        def later_only():
            assert lib.fm_procrustes_scatter_dense(P(depth), P(kinv), P(flow), P(logit), 100.0, 1, f, h, w, P(aux), P(pair_grad), None, P(g_w), None, None, P(consts), st) == 0

        def both():
            assert lib.fm_procrustes_scatter_dense(P(depth), P(kinv), P(flow), P(logit), 100.0, 1, f, h, w, P(aux), P(pair_grad), P(g_depth), P(g_w), P(first), P(entries), P(consts), st) == 0

        a, b, c, d = timed(fit), timed(later_only), timed(both), timed(fused)
        g_depth.zero_()
        both()
        planned = g_depth.clone()
        g_depth.zero_()
        fused()
        gap = float((g_depth - planned).norm() / planned.norm())
        out[f"{name}/{kind}"] = {"fit_ms": round(a, 3), "weights_only_ms": round(b, 3), "later_plus_taps_ms": round(c, 3), "fused_ms": round(d, 3),
                                 "fused_vs_planned": gap, "moments_vs_shipped": stats_gap, "entries_per_pixel": round(entries.numel() / (pairs * h * w), 4)}
        print(name, kind, out[f"{name}/{kind}"], flush=True)
print(json.dumps(out))
