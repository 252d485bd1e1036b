"""HBM streaming ceilings on this GPU for the read:write mixes of our kernels (through gpurun):
7:1 = fused flow loss, 4:3 = Adam, 1:1 = copy, n:0 = read-only.  Prints GB/s per mix."""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

import torch

here = Path(__file__).resolve().parent
lib_path = here / "libbw_probe.so"
if not lib_path.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(here / "bw_probe.hip"), "-o", str(lib_path)], check=True)
lib = ctypes.CDLL(str(lib_path))
lib.bw_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
quads = 150 * 720 * 1280 // 4  # one C1 depth-sized stream
src = torch.rand((7, quads * 4), device="cuda")
dst = torch.empty((3, quads * 4), device="cuda")
out = {}
for r, w in ((7, 0), (1, 1), (2, 1), (4, 3), (7, 1), (-6, 1)):
    for nt in (1, 0):
        for blocks in (256 * 4, 256 * 8, 256 * 16, 256 * 64):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                assert lib.bw_probe(src.data_ptr(), dst.data_ptr(), quads, r, w, nt, blocks, st) == 0
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.bw_probe(src.data_ptr(), dst.data_ptr(), quads, r, w, nt, blocks, st)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            out[f"r{r}w{w}_nt{nt}_b{blocks}"] = round(((7 if r == -6 else r) + w) * quads * 16 / ms / 1e6)
print(json.dumps(out))
