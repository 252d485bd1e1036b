"""Where does the time go INSIDE the one-block-per-frame Procrustes backward (fm_procrustes_bwd_planned)?  A variant of
fm_procrustes.hip built with -DFM_PHASE_CLOCKS stamps wall_clock64 (100 MHz) at the kernel's phase boundaries; this tool runs
it on C1-sized inputs (or `frames height width`) and prints the median phase durations over the blocks.
    SRC=fm_procrustes.hip tools/build_variants.sh clocks:-DFM_PHASE_CLOCKS && python tools/phase_clocks.py [frames height width]
"""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib, _ops  # noqa: E402

dev = "cuda:0"
f, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (150, 720, 1280)
p = 1000
g = torch.Generator(device=dev).manual_seed(0)
depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g)
logit = 0.01 * torch.randn((1, f - 1, h, w), device=dev, generator=g)
flow = 0.003 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)
fx = 0.85 * (h * w) ** 0.5
k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
kinv = torch.linalg.inv(k).contiguous()
idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64).to(dev)
pairs = f - 1
work = torch.zeros((pairs * 16 + (pairs + 2) // 2 + 1,), dtype=torch.float64, device=dev)
t_bwd, t_fwd = torch.empty((1, pairs, 4, 4), device=dev), torch.empty((1, pairs, 4, 4), device=dev)
aux = torch.empty((pairs, 40), dtype=torch.float64, device=dev)
ext = torch.empty((1, f, 4, 4), device=dev)
corr = torch.empty((pairs * p, 8), device=dev)
lib = _lib.library()
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
_ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)
_taps0 = _ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)[5]
assert lib.fm_procrustes_fit_chain(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, f, h, w, P(work), P(t_bwd), P(t_fwd), P(aux), P(ext), P(corr), P(_taps0), st) == 0
_ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)
pixels, first, vectors, weights, frame_first, _taps = _ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)
g_t = torch.randn((1, pairs, 4, 4), device=dev, generator=g)
g_depth = torch.zeros_like(depth)
g_w = torch.zeros_like(logit)
g_k = torch.empty((1, f, 3, 3), device=dev)

variant = ctypes.CDLL(str(ROOT / "build_variants" / "libfm_clocks.so"))
variant.fm_procrustes_bwd_planned.argtypes = _lib.SIGNATURES["fm_procrustes_bwd_planned"]
variant.fm_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]


def launch(library):
    assert library.fm_procrustes_bwd_planned(P(corr), P(kinv), 100.0, p, 1, f, h, w, P(aux), P(t_bwd), P(g_t), None, P(pixels),
                                             P(first), P(vectors), P(weights), P(frame_first), P(g_depth), P(g_w), P(g_k), 0, st) == 0


for lib_, name in ((lib, "product"), (variant, "clocked variant")):
    for _ in range(3):
        launch(lib_)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record()
        launch(lib_)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print(name, "kernel, events around the launch: median", round(ts[10] * 1e3, 1), "us")

blocks = min(f, 256)
slots = 12
buf = (ctypes.c_longlong * (blocks * slots))()
launch(variant)
torch.cuda.synchronize()
assert variant.fm_debug_phase_clocks(ctypes.addressof(buf), blocks) == slots
stamps = torch.tensor(list(buf), dtype=torch.float64).reshape(blocks, slots)[1 : blocks - 1] * 0.01  # us; interior frames (both roles)
names = ["entry -> records + first plan batch issued", "pose-solve backward (thread 0)", "dL/dKinv of the role (thread 0)", "wait at barrier 1",
         "per-correspondence gradients -> LDS", "wait at barrier 2", "gather: first batch", "gather: rest"]
pairs_ = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
out = {"workload": f"{f} x {h} x {w}, P = {p}; medians over the {blocks - 2} interior blocks, microseconds (wall_clock64, 10 ns ticks)",
       "touched pixels per frame": int((frame_first[2] - frame_first[1]).item())}
for name, (a, b) in zip(names, pairs_):
    out[name] = round(float((stamps[:, b] - stamps[:, a]).median()), 2)
out["block lifetime (entry -> exit)"] = round(float((stamps[:, 8] - stamps[:, 0]).median()), 2)
out["first entry -> last exit over all blocks"] = round(float(stamps[:, 8].max() - stamps[:, 0].min()), 2)
print(json.dumps(out))
