"""What follows the optimisation — SURVEY.md §8f rank 4.

* ``world_point_cloud``: the point cloud ``export_to_colmap`` builds frame by frame
  (flowmap/export/colmap.py:86-101: unproject, homogenise, camera-to-world, colours to
  point-major), as ONE launch (fm_world_points) instead of F × (unproject + einsum + 2 copies).
* ``write_ply``: the vertex layout of colmap.py:31-53 (x y z nx ny nz red green blue, binary
  little-endian — what 3D Gaussian Splatting reads), written with numpy alone.
* ``compute_ate``: flowmap/misc/ate.py:7-25 (scipy Procrustes alignment, RMS over all
  coordinates) — tiny, stays on the host.
The COLMAP camera/image records (colmap.py:114-, third_party/colmap) are file IO and stay with
the reference.
"""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _ops
from ._lib import call, check_device, ptr, stream_for


def world_point_cloud(depths: Tensor, intrinsics: Tensor, extrinsics: Tensor, colors: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """depths (F,H,W), intrinsics (F,3,3), extrinsics (F,4,4) camera-to-world, colors
    (F,3,H,W) -> world points (F·H·W,3) and colours (F·H·W,3), frames concatenated in order
    (colmap.py:86-101 with ``exports.*[0]``)."""
    check_device(depths, intrinsics, extrinsics, *(() if colors is None else (colors,)))
    if depths.dim() != 3:
        raise RuntimeError("flowmap_amd: depths must be (frame, height, width)")
    f, h, w = depths.shape
    if tuple(intrinsics.shape) != (f, 3, 3) or tuple(extrinsics.shape) != (f, 4, 4):
        raise RuntimeError("flowmap_amd: intrinsics / extrinsics do not match the depths")
    if colors is not None and tuple(colors.shape) != (f, 3, h, w):
        raise RuntimeError("flowmap_amd: colors must be (frame, 3, height, width)")
    with torch.no_grad():
        depths, extrinsics = _ops._f32c(depths, "depths"), _ops._f32c(extrinsics, "extrinsics")
        kinv = _ops.intrinsics_inverse(_ops._f32c(intrinsics, "intrinsics"))
        colors = None if colors is None else _ops._f32c(colors, "colors")
        xyz = torch.empty((f * h * w, 3), dtype=torch.float32, device=depths.device)
        rgb = None if colors is None else torch.empty_like(xyz)
        with _ops._guard(depths.device):
            call("fm_world_points", ptr(depths), ptr(kinv), ptr(extrinsics), ptr(colors), f, h, w, ptr(xyz), ptr(rgb), stream_for(depths))
    return xyz, rgb


_PLY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                       ("red", "u1"), ("green", "u1"), ("blue", "u1")])


def write_ply(path: Path, xyz: np.ndarray, rgb: np.ndarray) -> None:
    """colmap.py:31-53: zero normals, colours scaled by 255 and truncated to uint8."""
    xyz = np.asarray(xyz, dtype=np.float32)
    vertices = np.zeros(xyz.shape[0], dtype=_PLY_DTYPE)
    vertices["x"], vertices["y"], vertices["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    scaled = (np.asarray(rgb) * 255).astype(np.float32)
    vertices["red"], vertices["green"], vertices["blue"] = scaled[:, 0], scaled[:, 1], scaled[:, 2]
    kinds = {"<f4": "float", "|u1": "uchar"}
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {xyz.shape[0]}"]
    header += [f"property {kinds[_PLY_DTYPE[name].str]} {name}" for name in _PLY_DTYPE.names]
    header.append("end_header")
    with open(path, "wb") as handle:
        handle.write(("\n".join(header) + "\n").encode("ascii"))
        handle.write(vertices.tobytes())


def read_ply(path: Path) -> Tuple[np.ndarray, np.ndarray]:
    """Inverse of write_ply for the layout above (colmap.py:18-28)."""
    with open(path, "rb") as handle:
        count = None
        while True:
            line = handle.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                count = int(line.split()[-1])
            if line == "end_header":
                break
        vertices = np.frombuffer(handle.read(), dtype=_PLY_DTYPE, count=count)
    xyz = np.stack([vertices["x"], vertices["y"], vertices["z"]], axis=1)
    rgb = np.stack([vertices["red"], vertices["green"], vertices["blue"]], axis=1) / 255.0
    return xyz, rgb


def compute_ate(gt: Tensor, predicted: Tensor):
    """flowmap/misc/ate.py:7-25 -> (ate, aligned_gt, aligned_predicted)."""
    from scipy import spatial

    a, b, _ = spatial.procrustes(gt.detach().cpu().numpy(), predicted.detach().cpu().numpy())
    a = torch.tensor(a, dtype=torch.float32, device=gt.device)
    b = torch.tensor(b, dtype=torch.float32, device=predicted.device)
    return ((a - b) ** 2).mean() ** 0.5, a, b
