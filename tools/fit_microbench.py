"""Where do the microseconds of the sparse Procrustes fit go (P = 1000, C1-sized inputs)?  Times, with events on the
launch stream: moments alone (fm_procrustes_stats), + finish/solve (fm_procrustes_fit), + the separate pose chain
(fm_pose_chain_fwd), and the one-launch form the step uses (fm_procrustes_fit_chain).
    python tools/fit_microbench.py [frames height width]
"""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmap_amd import _lib  # noqa: E402

dev = "cuda:0"
f, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 and sys.argv[1].isdigit() else (150, 720, 1280)
p = 1000
g = torch.Generator(device=dev).manual_seed(0)
depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=dev, generator=g)
logit = 0.01 * torch.randn((1, f - 1, h, w), device=dev, generator=g)
flow = 0.003 * torch.randn((1, f - 1, h, w, 2), device=dev, generator=g)
fx = 0.85 * (h * w) ** 0.5
k = torch.tensor([[fx / w, 0, 0.5], [0, fx / h, 0.5], [0, 0, 1.0]], device=dev).expand(1, f, 3, 3).contiguous()
kinv = torch.linalg.inv(k).contiguous()
idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64).to(dev)
if "--contiguous" in sys.argv:
    # round 6: what the fit would cost if its operands were COALESCED (the bound on a compact image of the fit's taps): the P points are
    # P consecutive pixels of a row and the flow is zero, so a wave's gathers of weights / later depth / tap depths fall into the lines its neighbours read
    idx = torch.arange(p, dtype=torch.int64, device=dev) + (h // 2) * w
    flow.zero_()
pairs = f - 1
stats = torch.empty((pairs, 16), dtype=torch.float64, device=dev)
work = torch.zeros((pairs * 16 + (pairs + 2) // 2 + 1,), dtype=torch.float64, device=dev)
t_bwd = torch.empty((1, pairs, 4, 4), device=dev)
t_fwd = torch.empty_like(t_bwd)
aux = torch.empty((pairs, 40), dtype=torch.float64, device=dev)
ext = torch.empty((1, f, 4, 4), device=dev)
from flowmap_amd import _ops  # noqa: E402

_ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)
tap_records = _ops._procrustes_scatter_plan(idx, flow, 1, f, h, w)[5]
corr = torch.empty((pairs * p, 8), device=dev)
lib = _lib.library()
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ts[len(ts) // 2] * 1e3, 1)


def moments():
    assert lib.fm_procrustes_stats(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, 1, f, h, w, P(stats), st) == 0


def fit():
    assert lib.fm_procrustes_fit(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, 1, f, h, w, P(stats), P(t_bwd), P(t_fwd), P(aux), st) == 0


def fit_then_chain():
    fit()
    assert lib.fm_pose_chain_fwd(P(t_bwd), 1, pairs, P(ext), st) == 0


def chain_only():
    assert lib.fm_pose_chain_fwd(P(t_bwd), 1, pairs, P(ext), st) == 0


def fused():
    assert lib.fm_procrustes_fit_chain(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, f, h, w, P(work), P(t_bwd), P(t_fwd), P(aux), P(ext), None, None, st) == 0


def fused_no_chain():
    assert lib.fm_procrustes_fit_chain(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, f, h, w, P(work), P(t_bwd), P(t_fwd), P(aux), None, None, None, st) == 0


def fused_planned():  # the overfit loop from its second step on: static tap records in, correspondence records out
    assert lib.fm_procrustes_fit_chain(P(depth), P(kinv), None, P(flow), P(logit), 100.0, P(idx), p, 1, f, h, w, P(work), P(t_bwd), P(t_fwd), P(aux), P(ext),
                                       P(corr), P(tap_records), st) == 0


def fused_then_chain():
    fused_no_chain()
    assert lib.fm_pose_chain_fwd(P(t_bwd), 1, pairs, P(ext), st) == 0


out = {"indices": "P consecutive pixels, zero flow (coalesced operands)" if "--contiguous" in sys.argv else "linspace over the image (the reference's selection: every operand a cold line)",
       "workload": f"{f} x {h} x {w}, P = {p}; median of 30, microseconds incl. launch gaps between the launches of one call",
       "moments (memset + 1 launch)": timed(moments), "moments + finish/solve (memset + 2 launches)": timed(fit),
       "pose chain alone (1 launch)": timed(chain_only), "fit then chain (memset + 3 launches)": timed(fit_then_chain),
       "fit_chain (1 launch)": timed(fused), "fit_chain with static tap records + correspondence records out (1 launch)": timed(fused_planned),
       "fit_chain without the chain (1 launch)": timed(fused_no_chain), "fit_chain without the chain, then chain (2 launches)": timed(fused_then_chain)}
print(json.dumps(out))
