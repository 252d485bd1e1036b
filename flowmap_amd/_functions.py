"""Function-level operators on explicit point sets — ``torch.autograd.Function``s over the C ABI through ctypes: Unproject, Reproject,
BilinearSample, RobustMapping (flowmap/model/projection.py, flowmap/loss/mapping) and AlignRigid (flowmap/model/procrustes.py:7-51).
The per-step fused operators are in _ops.py / csrc/fm_torch.cpp."""

from __future__ import annotations

import ctypes
import warnings
import weakref
from typing import Optional

import torch
from torch import Tensor

from ._lib import call, check_device, ptr, stream_for, torch_ops  # noqa: F401
from ._base import AUX_STRIDE, PAIR_GRAD_STRIDE, STAT_STRIDE, TRACK_TILE, _f32c, _guard  # noqa: F401


def intrinsics_inverse(k):
    from ._ops import intrinsics_inverse as impl  # (K^-1 is kept on K: _ops.intrinsics_inverse; imported late — _ops imports this module)

    return impl(k)


# --------------------------------------------------------------------------------------
# Function-level building blocks on explicit point sets
# --------------------------------------------------------------------------------------


class Unproject(torch.autograd.Function):
    """unproject (flowmap/model/projection.py:76-90) for G groups of N points:
    xy (N,2) shared or (G,N,2); z (G,N); k (G,3,3) -> (G,N,3)."""

    @staticmethod
    def forward(ctx, xy, z, k):
        dev = check_device(xy, z, k)
        xy, z, k = _f32c(xy, "coordinates"), _f32c(z, "z"), _f32c(k, "intrinsics")
        if xy.requires_grad:
            raise RuntimeError("flowmap_amd: gradients w.r.t. image coordinates are not supported")
        g, n = z.shape
        shared = xy.dim() == 2
        kinv = intrinsics_inverse(k)
        out = torch.empty((g, n, 3), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_unproject_fwd", ptr(xy), 0 if shared else n * 2, ptr(z), ptr(kinv), g, n, ptr(out), stream_for(z))
        ctx.save_for_backward(xy, z, kinv)
        ctx.shared = shared
        return out

    @staticmethod
    def backward(ctx, g_out):
        xy, z, kinv = ctx.saved_tensors
        g, n = z.shape
        g_out = _f32c(g_out, "grad")
        g_z = torch.empty_like(z) if ctx.needs_input_grad[1] else None
        need_k = ctx.needs_input_grad[2]
        acc = torch.empty((g, 9), dtype=torch.float64, device=z.device) if need_k else None
        g_k = None
        with _guard(z.device):
            st = stream_for(z)
            call("fm_unproject_bwd", ptr(xy), 0 if ctx.shared else n * 2, ptr(z), ptr(kinv), ptr(g_out), g, n, ptr(g_z), ptr(acc), st)
            if need_k:
                g_k = torch.empty_like(kinv)
                call("fm_intrinsics_inverse_bwd", ptr(acc), ptr(kinv), g, ptr(g_k), 0, st)
        return None, g_z, g_k


class Reproject(torch.autograd.Function):
    """reproject_points (flowmap/model/projection.py:116-134): xyz (G,N,3), T (G,4,4),
    K (G,3,3) -> xy (G,N,2)."""

    @staticmethod
    def forward(ctx, xyz, t, k):
        dev = check_device(xyz, t, k)
        xyz, t, k = _f32c(xyz, "points"), _f32c(t, "transformations"), _f32c(k, "intrinsics")
        g, n, _ = xyz.shape
        out = torch.empty((g, n, 2), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_reproject_fwd", ptr(xyz), ptr(t), ptr(k), g, n, ptr(out), stream_for(xyz))
        ctx.save_for_backward(xyz, t, k)
        return out

    @staticmethod
    def backward(ctx, g_xy):
        xyz, t, k = ctx.saved_tensors
        g, n, _ = xyz.shape
        g_xy = _f32c(g_xy, "grad")
        g_xyz = torch.empty_like(xyz) if ctx.needs_input_grad[0] else None
        g_t = torch.empty_like(t)
        g_k = torch.empty_like(k)
        acc = torch.empty((g, 18), dtype=torch.float64, device=xyz.device)
        with _guard(xyz.device):
            call("fm_reproject_bwd", ptr(xyz), ptr(t), ptr(k), ptr(g_xy), g, n, ptr(g_xyz), ptr(g_t), ptr(g_k), ptr(acc),
                 stream_for(xyz))
        return g_xyz, g_t if ctx.needs_input_grad[1] else None, g_k if ctx.needs_input_grad[2] else None


class BilinearSample(torch.autograd.Function):
    """F.grid_sample(bilinear, border, align_corners=False) of a channels-last image
    (G,H,W,C) at normalised coordinates (G,P,2) in (0,1) -> (G,P,C)
    (flowmap/model/projection.py:235-241,266-272)."""

    @staticmethod
    def forward(ctx, img, xy):
        dev = check_device(img, xy)
        img, xy = _f32c(img, "image"), _f32c(xy, "coordinates")
        if xy.requires_grad:
            raise RuntimeError("flowmap_amd: gradients w.r.t. sampling coordinates are not supported")
        g, h, w, c = img.shape
        p = xy.shape[1]
        out = torch.empty((g, p, c), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_bilinear_sample_fwd", ptr(img), ptr(xy), g, h, w, c, p, ptr(out), stream_for(img))
        ctx.save_for_backward(xy)
        ctx.dims = (g, h, w, c, p)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (xy,) = ctx.saved_tensors
        g, h, w, c, p = ctx.dims
        g_out = _f32c(g_out, "grad")
        g_img = torch.zeros((g, h, w, c), dtype=torch.float32, device=xy.device)
        with _guard(xy.device):
            call("fm_bilinear_sample_bwd", ptr(g_out), ptr(xy), g, h, w, c, p, ptr(g_img), stream_for(xy))
        return g_img, None


class RobustMapping(torch.autograd.Function):
    """Mapping.forward (flowmap/loss/mapping/mapping.py:35-43) on (n,2) pairs."""

    @staticmethod
    def forward(ctx, a, b, kind, delta, ax, ay):
        dev = check_device(a, b)
        a, b = _f32c(a, "a"), _f32c(b, "b")
        n = a.shape[0]
        out = torch.empty((n,), dtype=torch.float32, device=dev)
        with _guard(dev):
            call("fm_mapping_fwd", ptr(a), ptr(b), n, kind, float(delta), float(ax), float(ay), ptr(out), stream_for(a))
        ctx.save_for_backward(a, b)
        ctx.cfg = (kind, float(delta), float(ax), float(ay))
        return out

    @staticmethod
    def backward(ctx, g_out):
        a, b = ctx.saved_tensors
        kind, delta, ax, ay = ctx.cfg
        g_out = _f32c(g_out, "grad")
        g_a = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        g_b = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        with _guard(a.device):
            call("fm_mapping_bwd", ptr(a), ptr(b), ptr(g_out), a.shape[0], kind, delta, ax, ay, ptr(g_a), ptr(g_b), stream_for(a))
        return g_a, g_b, None, None, None, None


class AlignRigid(torch.autograd.Function):
    """align_rigid (flowmap/model/procrustes.py:7-51): p, q (G,P,3), w (G,P) -> (G,4,4)."""

    @staticmethod
    def forward(ctx, p, q, w):
        dev = check_device(p, q, w)
        p, q, w = _f32c(p, "p"), _f32c(q, "q"), _f32c(w, "weights")
        g, n, _ = p.shape
        stats = torch.empty((g, STAT_STRIDE), dtype=torch.float64, device=dev)
        t = torch.empty((g, 4, 4), dtype=torch.float32, device=dev)
        aux = torch.empty((g, AUX_STRIDE), dtype=torch.float64, device=dev)
        with _guard(dev):
            st = stream_for(p)
            call("fm_align_rigid_stats", ptr(p), ptr(q), ptr(w), g, n, ptr(stats), st)
            call("fm_pose_solve", ptr(stats), g, ptr(t), None, ptr(aux), st)
        ctx.save_for_backward(p, q, w, t, aux)
        return t

    @staticmethod
    def backward(ctx, g_t):
        p, q, w, t, aux = ctx.saved_tensors
        g, n, _ = p.shape
        g_t = _f32c(g_t, "grad")
        pair_grad = torch.empty((g, PAIR_GRAD_STRIDE), dtype=torch.float64, device=p.device)
        g_p = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        g_q = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        g_w = torch.empty_like(w) if ctx.needs_input_grad[2] else None
        with _guard(p.device):
            st = stream_for(p)
            call("fm_pose_solve_bwd", ptr(g_t), None, ptr(t), ptr(aux), g, ptr(pair_grad), None, 0, st)
            call("fm_align_rigid_bwd", ptr(p), ptr(q), ptr(w), g, n, ptr(aux), ptr(pair_grad), ptr(g_p), ptr(g_q), ptr(g_w), st)
        return g_p, g_q, g_w
