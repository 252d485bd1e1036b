"""A/B timing of fm_flow_loss_fused across build variants (build_variants/*.so), GRAD on/off
and items-per-thread, interleaved in one process on identical C1-sized inputs."""
import ctypes
import glob
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib  # noqa: E402

P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
dev = "cuda:0"
f, h, w = 150, 720, 1280
g = torch.Generator(device=dev).manual_seed(0)
depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g)
ff = 0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)
fb = 0.01 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)
mf = torch.rand((1, f - 1, h, w), device=dev, generator=g)
mb = torch.rand((1, f - 1, h, w), device=dev, generator=g)
fx = 0.85 * (h * w) ** 0.5
k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
kinv = torch.linalg.inv(k).contiguous()
t = torch.eye(4, device=dev).repeat(1, f - 1, 1, 1)
t[..., :3, 3] = 0.01 * torch.randn((1, f - 1, 3), device=dev, generator=g)
t = t.contiguous()
norm = torch.tensor([1e-3, 1.0], device=dev)
gd = torch.empty_like(depth)
acc = torch.zeros((f * 2 * 20,), dtype=torch.float64, device=dev)  # the kernel adds into it (timing only: never finalised)
sc = (h * w) ** 0.5
algo = h * w * (8 * f + 24 * (f - 1))

libs = {"shipped": str(_lib.LIB_PATH)}
for p in sorted(glob.glob(str(ROOT / "build_variants" / "*.so"))):
    libs[Path(p).stem.replace("libfm_", "")] = p
fns = {}
for name, path in libs.items():
    fn = ctypes.CDLL(path).fm_flow_loss_fused
    fn.argtypes = _lib.SIGNATURES["fm_flow_loss_fused"]
    fn.restype = I
    fns[name] = fn


def launch(fn, grad, ipt):
    st = torch.cuda.current_stream().cuda_stream
    return fn(depth.data_ptr(), k.data_ptr(), kinv.data_ptr(), t.data_ptr(), t.data_ptr(), ff.data_ptr(), fb.data_ptr(), mf.data_ptr(),
              mb.data_ptr(), packed.data_ptr() if PACKED[0] else None, norm.data_ptr() if grad else None, 1, f, h, w, 0, 0.01, w / sc, h / sc, gd.data_ptr() if grad else None,
              acc.data_ptr(), ipt, st)


pack = ctypes.CDLL(libs["shipped"]).fm_flow_pack_inputs
pack.argtypes = _lib.SIGNATURES["fm_flow_pack_inputs"]
packed = torch.empty((f, (h * w // 4 + 63) // 64, 6, 64, 4), device=dev)
assert pack(ff.data_ptr(), fb.data_ptr(), mf.data_ptr(), mb.data_ptr(), 1, f, h, w, packed.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
PACKED = [False]
IPT = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (2, 4, 6, 8)
PKS = (True,) if len(sys.argv) > 1 else (False, True)
configs = [(n, True, i, pk) for n in fns for i in IPT for pk in PKS] + [("shipped", False, 4, False), ("shipped", False, 4, True)]
times = {c: [] for c in configs}
for rnd in range(6):
    for c in configs:
        name, grad, ipt, PACKED[0] = c
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            assert launch(fns[name], grad, ipt) == 0
        e.record()
        torch.cuda.synchronize()
        if rnd > 0:
            times[c].append(s.elapsed_time(e) / 3)
for c, v in times.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"{c[0]:8s} grad={int(c[1])} ipt={c[2]:2d} packed={int(c[3])}  median {med:.4f} ms  min {v[0]:.4f}  -> {algo / med / 1e6:.0f} GB/s algorithmic (grad=1 bytes)")
