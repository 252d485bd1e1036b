"""Stress the one-launch sparse fit (fm_procrustes_fit_chain): many back-to-back launches on changing inputs, each
compared with the memset + three-launch form (fm_procrustes_fit + fm_pose_chain_fwd).  The last-block election and the
cross-XCD visibility of the poses are the things under test.
    python tools/fit_stress.py [iterations]
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib  # noqa: E402

CASES = ((150, 180, 240, 1000), (150, 360, 640, 1000), (37, 96, 128, 3000), (2, 64, 64, 50), (150, 720, 1280, 1000))


def _ptr(t):
    return None if t is None else t.data_ptr()


def run(iters: int, dev: str = "cuda:0", verbose: bool = True) -> float:
    """Max |extrinsics - reference| over all launches; asserts agreement to 1e-4 and a clean workspace."""
    lib = _lib.library()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    worst = 0.0
    for f, h, w, p in CASES:
        pairs = f - 1
        fx = 0.85 * (h * w) ** 0.5
        k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
        kinv = torch.linalg.inv(k).contiguous()
        idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64).to(dev)
        logit = 0.01 * torch.randn((1, pairs, h, w), device=dev, generator=g)
        flow = 0.003 * torch.randn((1, pairs, h, w, 2), device=dev, generator=g)
        stats = torch.empty((pairs, 16), dtype=torch.float64, device=dev)
        work = torch.zeros((pairs * 16 + (pairs + 2) // 2 + 1,), dtype=torch.float64, device=dev)
        aux, aux2 = (torch.empty((pairs, 40), dtype=torch.float64, device=dev) for _ in range(2))
        tb, tf, tb2, tf2 = (torch.empty((1, pairs, 4, 4), device=dev) for _ in range(4))
        ext, ext2 = (torch.empty((1, f, 4, 4), device=dev) for _ in range(2))
        n = iters if h * w < 500000 else max(iters // 10, 20)
        depth = None
        for it in range(n):
            depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g) if it % 50 == 0 else depth * (1 + 1e-4)
            ext.fill_(float("nan"))
            assert lib.fm_procrustes_fit_chain(_ptr(depth), _ptr(kinv), None, _ptr(flow), _ptr(logit), 100.0, _ptr(idx), p, 1, f, h, w,
                                               _ptr(work), _ptr(tb), _ptr(tf), _ptr(aux), _ptr(ext), None, None, st) == 0
            assert lib.fm_procrustes_fit(_ptr(depth), _ptr(kinv), None, _ptr(flow), _ptr(logit), 100.0, _ptr(idx), p, 1, 1, f, h, w,
                                         _ptr(stats), _ptr(tb2), _ptr(tf2), _ptr(aux2), st) == 0
            assert lib.fm_pose_chain_fwd(_ptr(tb2), 1, pairs, _ptr(ext2), st) == 0
            if it % 25 == 0 or it == n - 1:
                err = float((ext - ext2).abs().max())
                assert err == err and err < 1e-4, (f, h, w, p, it, err)
                worst = max(worst, err)
                assert float(work.abs().max()) == 0.0, "the workspace was not left clean"
        if verbose:
            print(f"{f} x {h} x {w}, P = {p}: {n} launches, max |ext - reference| so far {worst:.2e}", flush=True)
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    print("ok")
