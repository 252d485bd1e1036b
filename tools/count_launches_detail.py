import os, sys as _s; _s.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, runpy, collections
sys.argv = ["bench.py", "--cpu-frames", "0", "--steps", "5", "--warmup", "3"]
import bench
orig = bench.count_launches
def counted(step, device):
    import torch
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(); torch.cuda.synchronize(device)
    names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
    print(collections.Counter(n[:90] for n in names), file=sys.stderr)
    return orig(step, device)
bench.count_launches = counted
bench.main()
