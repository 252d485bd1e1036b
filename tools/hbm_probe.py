"""What this box's HBM delivers to plain streaming kernels (context for the roofline fractions in DESIGN.md / bench.py, which are
quoted against the 8 TB/s peak): a device-to-device copy (read N + write N), a fill (write N) and a read-only reduction of 2 GiB,
timed with events over 20 repetitions.    python tools/hbm_probe.py"""
import json

import torch

dev = "cuda:0"
n = 2 * 2**30 // 4
src = torch.rand((n,), device=dev)
dst = torch.empty_like(src)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


out = {}
t = timed(lambda: dst.copy_(src))
out["copy_GBps"] = round(2 * n * 4 / t / 1e9, 1)
t = timed(lambda: dst.fill_(1.0))
out["fill_GBps"] = round(n * 4 / t / 1e9, 1)
t = timed(lambda: src.sum())
out["sum_GBps"] = round(n * 4 / t / 1e9, 1)
t = timed(lambda: torch.add(src, dst, out=dst))
out["add_inplace_GBps"] = round(3 * n * 4 / t / 1e9, 1)
print(json.dumps(out))
