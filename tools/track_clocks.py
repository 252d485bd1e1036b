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
begin = (out[:, 6] >> 8).astype(np.float64) / 100.0  # us
out[:, 6] &= 0xFF
begin -= begin.min()
end = begin + out[:, 5] / 100.0
print(f"waves began over {begin.max():.1f} us (median {np.median(begin):.1f}, p90 {np.percentile(begin, 90):.1f}); last wave ended at {end.max():.1f} us; "
      f"ends: median {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} p99 {np.percentile(end, 99):.1f}")
order = np.argsort(-end)[:12]
print("  the waves that ended last: (work item, targets, began, prologue, terms, reduce, epilogue, ended)")
for i in order:
    print(f"    {i:5d} {out[i, 6]:3d} {begin[i]:7.1f} {out[i, 0] / 100:7.1f} {out[i, 2] / 100:7.1f} {out[i, 3] / 100:7.1f} {out[i, 4] / 100:7.1f} {end[i]:7.1f}")
for lo in range(0, len(out), max(len(out) // 8, 1)):
    sel = slice(lo, lo + max(len(out) // 8, 1))
    print(f"    work items {lo:4d}+: targets {np.median(out[sel, 6]):3.0f} began {np.median(begin[sel]):6.1f} prologue {np.median(out[sel, 0]) / 100:6.1f} terms {np.median(out[sel, 2]) / 100:6.1f} "
          f"(p90 {np.percentile(out[sel, 2], 90) / 100:6.1f}) whole {np.median(out[sel, 5]) / 100:6.1f} ended {np.median(end[sel]):6.1f} (max {end[sel].max():6.1f})")
by_t = {}
for t in np.unique(out[:, 6]):
    sel = out[:, 6] == t
    by_t[int(t)] = (int(sel.sum()), round(float(np.median(out[sel, 2]) / 100), 1), round(float(np.percentile(out[sel, 2], 90) / 100), 1))
print("  terms by number of targets (waves, median us, p90):", by_t)
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
# what separates the fast waves from the slow ones?  per-target time of the terms phase against: point group, XCC, the SIMD partner
tt = out[:, 2] / np.maximum(out[:, 6], 1) / 100.0
slow = tt > 0.5 * (np.percentile(tt, 10) + np.percentile(tt, 95))
print(f"  terms per target: p10 {np.percentile(tt, 10):.2f} median {np.median(tt):.2f} p95 {np.percentile(tt, 95):.2f} us; 'slow' waves: {slow.sum()} of {len(tt)}")
work = np.arange(len(out))
grp = work % 10
print("    by point group (work % 10): " + " ".join(f"{g}:{np.median(tt[grp == g]):.2f}/{slow[grp == g].mean():.2f}" for g in range(10)))
print("    by XCC: " + " ".join(f"{x}:{np.median(tt[xcc == x]):.2f}/{slow[xcc == x].mean():.2f}" for x in np.unique(xcc)))
ids = np.unique(simd)
pair_kinds = {"both slow": 0, "one slow": 0, "none slow": 0, "alone slow": 0, "alone fast": 0}
for sid in ids:
    sel = np.nonzero(simd == sid)[0]
    if len(sel) == 1:
        pair_kinds["alone slow" if slow[sel[0]] else "alone fast"] += 1
    else:
        k = int(slow[sel].sum())
        pair_kinds["both slow" if k == len(sel) else ("one slow" if k else "none slow")] += 1
print("    SIMDs by their waves:", pair_kinds)
cu = simd >> 4
cu_slow = {int(c): int(slow[cu == c].sum()) for c in np.unique(cu)}
hist = np.unique(list(cu_slow.values()), return_counts=True)
print("    slow waves per CU (count of CUs):", dict(zip(hist[0].tolist(), hist[1].tolist())))
tile = work // 10
first_tile_of_targets = {}
for t in np.unique(out[:, 6]):
    sel = out[:, 6] == t
    tiles = np.unique(tile[sel])
    print(f"    targets {int(t):2d}: tiles {tiles.min()}..{tiles.max()}: per tile median terms/target " + " ".join(f"{np.median(tt[sel & (tile == x)]):.2f}" for x in tiles[:12]))
per_target = out[:, 1:4].sum(axis=1) / np.maximum(out[:, 6], 1) / 100.0
print(f"  per target iteration         {np.median(per_target):8.3f} us")
