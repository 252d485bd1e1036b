"""Where does the time go INSIDE track_pairs?  Rebuilds the library in place with fm_track.hip compiled -DFM_TRACK_CLOCKS (run this on
the GPU box through gpurun: nothing is written back), runs the C2 bench in this process and reads the per-wave phase clocks of the
last launch: prologue (sampling), per target frame: scalar constants / residual terms / reduction + store, epilogue (source role).
    python tools/track_clocks.py"""
import ctypes
import runpy
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import flowmap_amd.build as b  # noqa: E402

b.FILE_FLAGS["fm_track.hip"] = ["-DFM_TRACK_CLOCKS"] + sys.argv[1:]
b.build_library(force=True, verbose=False)
from flowmap_amd import _lib  # noqa: E402

import os  # noqa: E402

sys.argv = ["bench.py", "--config", "c2", "--cpu-frames", "0", "--steps", "5", "--warmup", "2", "--sustained-steps", "0"] + (
    ["--no-tap-exchange"] if os.environ.get("FLOWMAP_NO_TAP_EXCHANGE") else [])
try:
    runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
except SystemExit:
    pass
lib = _lib.library()
lib.fm_debug_track_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = 2048
out = np.zeros((n, 8), dtype=np.int64)
assert lib.fm_debug_track_clocks(out.ctypes.data, n) == 8
out = out[out[:, 5] > 0]
names = ["prologue", "targets: scalar constants", "targets: terms", "targets: reduce + store", "epilogue", "whole wave"]
print(f"{len(out)} waves; median / p90 per wave in us (100 MHz clock); targets per wave median {np.median(out[:, 6]):.0f}")
for i, name in enumerate(names):
    us = out[:, i] / 100.0
    print(f"  {name:28s} {np.median(us):8.2f} {np.percentile(us, 90):8.2f}")
hw, xcc = out[:, 7] & 0xFFFFFFFF, (out[:, 7] >> 32) & 0xF
simd = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)  # (XCC, SE, SH, CU, SIMD)
per_simd = np.unique(simd, return_counts=True)[1]
per_cu = np.unique(simd >> 4, return_counts=True)[1]
print(f"  waves per SIMD that ran any: {dict(zip(*np.unique(per_simd, return_counts=True)))} over {len(per_simd)} SIMDs; waves per CU: {dict(zip(*np.unique(per_cu, return_counts=True)))} over {len(per_cu)} CUs")
terms = out[:, 2] / 100.0
for n in sorted(set(per_simd)):
    sel = np.isin(simd, np.unique(simd)[per_simd == n])
    print(f"    SIMDs with {n} waves: terms median {np.median(terms[sel]):.1f} us, whole wave median {np.median(out[sel, 5]) / 100.0:.1f} us")
per_target = out[:, 1:4].sum(axis=1) / np.maximum(out[:, 6], 1) / 100.0
print(f"  per target iteration         {np.median(per_target):8.3f} us")
