"""Where do the ~100 us of fm_procrustes_scatter go at C1?  Time it with outputs disabled."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib, _ops
from flowmap_amd._lib import call, ptr

dev = "cuda:0"
f, h, w, P = 150, 720, 1280, 1000
g = torch.Generator(device=dev).manual_seed(0)
depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g)
fb = 0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)
wl = 0.01 * torch.randn((1, f - 1, h, w), device=dev, generator=g)
fx = 0.85 * (h * w) ** 0.5
k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
kinv = _ops.intrinsics_inverse(k)
idx = torch.linspace(0, h * w - 1, P, dtype=torch.int64).to(dev)
pairs = f - 1
stats = torch.empty((pairs, 16), dtype=torch.float64, device=dev)
t_bwd = torch.empty((1, pairs, 4, 4), device=dev)
aux = torch.empty((pairs, 40), dtype=torch.float64, device=dev)
pg = torch.empty((pairs, 20), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
call("fm_procrustes_stats", ptr(depth), ptr(kinv), None, ptr(fb), ptr(wl), 100.0, ptr(idx), P, 1, 1, f, h, w, ptr(stats), st)
call("fm_pose_solve", ptr(stats), pairs, ptr(t_bwd), None, ptr(aux), st)
g_t = torch.randn((1, pairs, 4, 4), device=dev, generator=g)
call("fm_pose_solve_bwd", ptr(g_t), None, ptr(t_bwd), ptr(aux), pairs, ptr(pg), None, 0, st)
gd = torch.zeros_like(depth)
gw = torch.zeros_like(wl)
ka = torch.zeros((f, 9), dtype=torch.float64, device=dev)


def run(name, a, b, c, reps=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        call("fm_procrustes_scatter", ptr(depth), ptr(kinv), None, ptr(fb), ptr(wl), 100.0, ptr(idx), P, 1, 1, f, h, w, ptr(aux), ptr(pg),
             ptr(a), None, ptr(b), ptr(c), None, None, st)
    e.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {s.elapsed_time(e) / reps * 1e3:8.1f} us")


for _ in range(2):
    run("all outputs", gd, gw, ka)
    run("no depth atomics", None, gw, ka)
    run("no weight atomics", gd, None, ka)
    run("no kinv accumulation", gd, gw, None)
    run("nothing", None, None, None)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    call("fm_procrustes_stats", ptr(depth), ptr(kinv), None, ptr(fb), ptr(wl), 100.0, ptr(idx), P, 1, 1, f, h, w, ptr(stats), st)
e.record(); torch.cuda.synchronize()
print("stats (2 passes + memset)", s.elapsed_time(e) / 20 * 1e3, "us")
