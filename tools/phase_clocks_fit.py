"""Where does the time go INSIDE the one-block-per-pair Procrustes fit (fm_procrustes_fit_chain, P = 1000)?  The -DFM_PHASE_CLOCKS
variant of fm_procrustes.hip stamps wall_clock64 (100 MHz) at the kernel's phase boundaries (thread 0 of each block).
    SRC=fm_procrustes.hip tools/build_variants.sh clocks:-DFM_PHASE_CLOCKS && python tools/phase_clocks_fit.py [frames height width]
"""
import ctypes
import json
import sys
from pathlib import Path

import numpy as np
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
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
_ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)
taps = _ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)[5]
scratch = torch.empty((256 << 20) // 4, device=dev)  # written between launches: the gathers start from cold lines, as in a step

variant = ctypes.CDLL(str(ROOT / "build_variants" / "libfm_clocks.so"))
variant.fm_procrustes_fit_chain.argtypes = _lib.SIGNATURES["fm_procrustes_fit_chain"]
variant.fm_debug_phase_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]


def launch(library, with_ext=True):
    assert library.fm_procrustes_fit_chain(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, f, h, w, P(work), P(t_bwd), P(t_fwd), P(aux),
                                           P(ext) if with_ext else None, P(corr), P(taps), st) == 0


for lib_, name in ((_lib.library(), "product"), (variant, "clocked variant")):
    for with_ext in (True, False):
        times = []
        for _ in range(12):
            scratch.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            launch(lib_, with_ext)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) * 1e3)
        print(f"{name}{'' if with_ext else ' without the pose chain'}: events around the launch: median {np.median(times[2:]):.1f} us")
scratch.fill_(1.0)
launch(variant)
torch.cuda.synchronize()
n = min(pairs, 256)
out = np.zeros((n, 12), dtype=np.int64)
assert variant.fm_debug_phase_clocks(out.ctypes.data, n) == 12
t0 = out[:, 0].min()
d = lambda a, b: float(np.median((out[:, b] - out[:, a]) / 100.0))  # noqa: E731
last = int(np.argmax(out[:, 4]))
print(json.dumps({
    "workload": f"{f} x {h} x {w}, P = {p}; medians over the first {n} blocks, microseconds (wall_clock64, 10 ns ticks)",
    "block entry after the first block's": float(np.median((out[:, 0] - t0) / 100.0)),
    "entry -> thread 0's gathers returned, moments formed": d(0, 1),
    "wave sums -> LDS -> added (2 barriers)": d(1, 2),
    "moments_finish + pose_solve_one (thread 0, fp64)": d(2, 3),
    "fence + counter + barrier": d(3, 4),
    "block lifetime without the chain": d(0, 4),
    "the last block: pose chain": float((out[last, 5] - out[last, 4]) / 100.0),
    "first entry -> chain done": float((out[last, 5] - t0) / 100.0),
}))
