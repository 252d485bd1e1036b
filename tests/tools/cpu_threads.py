"""Which thread count gives the best CPU baseline on this host? (bounded sweep)"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import flowmap_oracle as orc
f, h, w = 4, 720, 1280
depth, wlogit, flows = orc.synth_iid(f, h, w, seed=0)
depth.requires_grad_(True); wlogit.requires_grad_(True)
focal = torch.tensor(0.85, requires_grad=True)
def step():
    for p in (depth, wlogit, focal):
        p.grad = None
    total, _, _ = orc.explicit_depth_step(depth, wlogit, focal, flows, (h, w), num_points=1000)
    total.backward()
for t in [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(t)
    step()
    t0 = time.perf_counter(); step(); step(); dt = (time.perf_counter() - t0) / 2
    print(f"threads {t}: {dt:.3f} s/iter for {f} frames @720p", flush=True)
