"""CPU oracle for FlowMap's reprojection / flow-consistency inner loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``flowmap_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / the timed CPU baseline ("port").

What it is: an independent restatement, in plain PyTorch CPU ops, of the reference
algorithm (dcharatan/flowmap).  Gradients come from ``torch.autograd`` walking the
restated op chain, exactly as the reference obtains them.  Every function names the
reference ``file:line`` it follows.  dtype-generic: run it in fp32 (the reference's
precision) or fp64 (tie-breaker for noise-floor questions).

Parity pin: the reference ships no tests / golden vectors, so the
pin is the reference itself, imported in the build container by
``oracle/make_golden.py`` (which writes ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks this restatement against those fixtures.

Conventions: normalised image coordinates, x right / y down in (0,1); extrinsics are
camera-to-world; ``b`` batch, ``F`` frames, ``H×W`` pixels, ``P`` points.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn.functional as tnf
from torch import Tensor

EPS_PROJECT = 1e-5  # flowmap/model/projection.py:52
INF_PROJECT = 1e8  # flowmap/model/projection.py:53


# --------------------------------------------------------------------------------------
# Plain containers mirroring the reference's hot-path input types
# --------------------------------------------------------------------------------------


@dataclass
class OFlows:
    """flowmap/flow/flow_predictor.py:16-21"""

    forward: Tensor  # (b, F-1, H, W, 2)
    backward: Tensor  # (b, F-1, H, W, 2)
    forward_mask: Tensor  # (b, F-1, H, W)
    backward_mask: Tensor  # (b, F-1, H, W)


@dataclass
class OTracks:
    """flowmap/tracking/track_predictor.py:13-20"""

    xy: Tensor  # (b, f, P, 2)
    visibility: Tensor  # (b, f, P) bool
    start_frame: int


# --------------------------------------------------------------------------------------
# Geometry (flowmap/model/projection.py)
# --------------------------------------------------------------------------------------


def pixel_grid(shape: Sequence[int], device="cpu", dtype=torch.float32):
    """flowmap/model/projection.py:93-113 (``sample_image_grid``).

    Returns ``xy`` of shape (*shape, len(shape)) holding pixel-centre coordinates with
    the LAST axis fastest in component 0 (x = (col+0.5)/W, y = (row+0.5)/H for 2-D),
    plus the int64 (row, col) index grid.  The reference evaluates (int64 + 0.5) / n
    in fp32; we do the same then cast.
    """
    axes = [torch.arange(n, device=device) for n in shape]
    idx = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)
    centres = [((a + 0.5) / n).to(dtype) for a, n in zip(axes, shape)]
    grids = torch.meshgrid(*centres, indexing="ij")
    xy = torch.stack(list(reversed(grids)), dim=-1)
    return xy, idx


def append_one(x: Tensor) -> Tensor:
    """flowmap/model/projection.py:11-15 (``homogenize_points``)."""
    return torch.cat((x, torch.ones_like(x[..., :1])), dim=-1)


def append_zero(x: Tensor) -> Tensor:
    """flowmap/model/projection.py:18-22 (``homogenize_vectors``)."""
    return torch.cat((x, torch.zeros_like(x[..., :1])), dim=-1)


def matvec(m: Tensor, v: Tensor) -> Tensor:
    """flowmap/model/projection.py:25-30 (``transform_rigid``): m[..., i, j] v[..., j].

    As a broadcasting einsum, which is how the reference states it: ATen folds the dimensions over which ``m`` is constant (the
    pixels of a frame) into ONE (3x3)·(3xN) product per frame.  ``m @ v[..., None]`` — this function until round 6 — expands ``m`` to
    one tiny matrix per pixel instead: the same numbers, 8x the time in ``bmm`` and 2x per step, i.e. a CPU baseline half as fast as
    the code it stands for."""
    return torch.einsum("...ij,...j->...i", m, v)


def lift(xy: Tensor, z: Tensor, k: Tensor) -> Tensor:
    """flowmap/model/projection.py:76-90 (``unproject``): (K^-1 [x,y,1]) * z."""
    rays = matvec(torch.linalg.inv(k), append_one(xy))
    return rays * z.unsqueeze(-1)


def pinhole(points: Tensor, k: Tensor, eps: float = EPS_PROJECT, inf: float = INF_PROJECT) -> Tensor:
    """flowmap/model/projection.py:49-58 (``project_camera_space``).

    Divide ALL components by (z + eps), clamp non-finite values, multiply by the full
    K, drop the last row.  Note the homogeneous component is z/(z+eps), not 1.
    """
    q = points / (points[..., -1:] + eps)
    q = torch.nan_to_num(q, nan=0.0, posinf=inf, neginf=-inf)
    return matvec(k, q)[..., :-1]


def world_to_image(points: Tensor, cam2world: Tensor, k: Tensor, eps: float = EPS_PROJECT):
    """flowmap/model/projection.py:61-73 (``project``)."""
    cam = matvec(torch.linalg.inv(cam2world), append_one(points))[..., :-1]
    return pinhole(cam, k, eps=eps), cam[..., -1] >= 0


def warp_points(xyz: Tensor, rel: Tensor, k: Tensor) -> Tensor:
    """flowmap/model/projection.py:116-134 (``reproject_points``)."""
    moved = matvec(rel, append_one(xyz))[..., :3]
    return pinhole(moved, k)


def _pad_pose_dims(m: Tensor, n_grid: int) -> Tensor:
    # (b, f, i, j) -> (b, f, 1 × n_grid, i, j); projection.py:156-157,178-179
    return m.reshape(*m.shape[:2], *([1] * n_grid), *m.shape[2:])


def forward_flow_positions(surfaces: Tensor, e: Tensor, k: Tensor) -> Tensor:
    """flowmap/model/projection.py:143-162 (``compute_forward_flow``)."""
    rel = torch.linalg.inv(e[:, 1:]) @ e[:, :-1]
    g = surfaces.ndim - 3
    return warp_points(surfaces[:, :-1], _pad_pose_dims(rel, g), _pad_pose_dims(k[:, 1:], g))


def backward_flow_positions(surfaces: Tensor, e: Tensor, k: Tensor) -> Tensor:
    """flowmap/model/projection.py:165-184 (``compute_backward_flow``)."""
    rel = torch.linalg.inv(e[:, :-1]) @ e[:, 1:]
    g = surfaces.ndim - 3
    return warp_points(surfaces[:, 1:], _pad_pose_dims(rel, g), _pad_pose_dims(k[:, :-1], g))


def chain_poses(rel: Tensor) -> Tensor:
    """flowmap/model/projection.py:187-210 (``get_extrinsics``): P_0 = I, P_k = P_{k-1} rel_{k-1}.

    The reference hard-codes fp32 for the identity (:204); we follow the input dtype so
    the fp64 tie-breaker runs are genuinely fp64.
    """
    steps = rel.shape[-3]
    cur = torch.eye(4, dtype=rel.dtype, device=rel.device).expand(*rel.shape[:-3], 4, 4).contiguous()
    out = [cur]
    for i in range(steps):
        cur = cur @ rel[..., i, :, :]
        out.append(cur)
    return torch.stack(out, dim=-3)


def bilinear_border(img_bfhwc: Tensor, xy_bfp: Tensor) -> Tensor:
    """Shared by projection.py:235-242 and :266-273.

    ``F.grid_sample(mode="bilinear", padding_mode="border", align_corners=False)`` of a
    channels-last (b, f, H, W, C) image at normalised (0,1) coordinates (b, f, P, 2).
    Returns (b, f, P, C).
    """
    b, f, h, w, c = img_bfhwc.shape
    p = xy_bfp.shape[2]
    out = tnf.grid_sample(
        img_bfhwc.reshape(b * f, h, w, c).permute(0, 3, 1, 2),
        (xy_bfp * 2 - 1).reshape(b * f, p, 1, 2),
        mode="bilinear",
        padding_mode="border",
        align_corners=False,
    )
    return out.reshape(b, f, c, p).permute(0, 1, 3, 2)


def rigid_fit(p: Tensor, q: Tensor, w: Tensor) -> Tensor:
    """flowmap/model/procrustes.py:7-51 (``align_rigid``): T with T·p ≈ q, weighted.

    Centroids use weights normalised with +1e-8 (:23-25); the covariance uses the RAW
    weights (:32); reflection fix on the last singular direction (:35-39).
    """
    wn = w / (w.sum(dim=-1, keepdim=True) + 1e-8)
    pc = (wn.unsqueeze(-1) * p).sum(dim=-2)
    qc = (wn.unsqueeze(-1) * q).sum(dim=-2)
    p0 = p - pc.unsqueeze(-2)
    q0 = q - qc.unsqueeze(-2)
    cov = (q0 * w.unsqueeze(-1)).transpose(-1, -2) @ p0
    u, _, vt = torch.linalg.svd(cov)
    flip = torch.eye(3, dtype=p.dtype, device=p.device).expand(*p.shape[:-2], 3, 3).contiguous()
    flip[..., 2, 2] = (torch.linalg.det(u) * torch.linalg.det(vt)).sign()
    rot = u @ flip @ vt
    trans = qc - matvec(rot, pc)
    out = torch.eye(4, dtype=p.dtype, device=p.device).expand(*p.shape[:-2], 4, 4).contiguous()
    out[..., :3, :3] = rot
    out[..., :3, 3] = trans
    return out


def fit_poses(surfaces: Tensor, bwd_flow: Tensor, bwd_weight: Tensor, indices: Tensor) -> Tensor:
    """flowmap/model/projection.py:213-252 (``align_surfaces``)."""
    b, f, h, w, _ = surfaces.shape
    xy, _ = pixel_grid((h, w), surfaces.device, surfaces.dtype)
    later_pts = surfaces[:, 1:].reshape(b, f - 1, h * w, 3)[:, :, indices]
    where = (xy + bwd_flow).reshape(b, f - 1, h * w, 2)[:, :, indices]
    earlier_pts = bilinear_border(surfaces[:, :-1], where)
    wts = bwd_weight.reshape(b, f - 1, h * w)[..., indices]
    return chain_poses(rigid_fit(later_pts, earlier_pts, wts))


def track_positions(surfaces: Tensor, e: Tensor, k: Tensor, tracks: OTracks):
    """flowmap/model/projection.py:255-298 (``compute_track_flow``)."""
    pts = bilinear_border(surfaces, tracks.xy)  # (b, fs, P, 3)
    rel = torch.linalg.inv(e)[:, None, :, None] @ e[:, :, None, None]  # (b, fs, ft, 1, 4, 4)
    target = warp_points(pts[:, :, None], rel, k[:, None, :, None])  # (b, fs, ft, P, 2)
    src = tracks.xy[:, :, None]
    vis = tracks.visibility[:, :, None] & tracks.visibility[:, None, :]
    inside_src = (src >= 0).all(-1) & (src < 1).all(-1)
    inside_tgt = (target >= 0).all(-1) & (target < 1).all(-1)
    return target, vis & inside_src & inside_tgt


# --------------------------------------------------------------------------------------
# Robust mappings (flowmap/loss/mapping/*)
# --------------------------------------------------------------------------------------


def aspect_scale(v: Tensor, hw) -> Tensor:
    """flowmap/loss/mapping/mapping.py:9-24 (``fix_aspect_ratio``)."""
    h, w = hw
    s = (h * w) ** 0.5
    return v * torch.tensor((w / s, h / s), dtype=v.dtype, device=v.device)


def robust(a: Tensor, b: Tensor, hw, kind: str = "huber", delta: float = 0.01) -> Tensor:
    """mapping.py:35-43 + mapping_huber.py:19-34 / mapping_l1.py:16-20 / mapping_l2.py:16-21."""
    d = aspect_scale(a, hw) - aspect_scale(b, hw)
    if kind == "huber":
        n = d.norm(dim=-1)
        return tnf.huber_loss(n, torch.zeros_like(n), reduction="none", delta=delta) / delta
    if kind == "l1":
        return d.norm(dim=-1)
    if kind == "l2":
        return 0.5 * (d * d).sum(dim=-1)
    raise ValueError(kind)


# --------------------------------------------------------------------------------------
# Losses (flowmap/loss/loss_flow.py, loss_tracking.py, loss.py)
# --------------------------------------------------------------------------------------


def _or_one(v):
    # loss_flow.py:70 / loss_tracking.py:61: ``valid_sum or 1``
    return v if bool(v != 0) else 1


def flow_loss(surfaces, e, k, flows: OFlows, hw, kind="huber", delta=0.01) -> Tensor:
    """flowmap/loss/loss_flow.py:31-70 (unweighted)."""
    xy, _ = pixel_grid(hw, surfaces.device, surfaces.dtype)
    f_term = robust(forward_flow_positions(surfaces, e, k) - xy, flows.forward, hw, kind, delta)
    b_term = robust(backward_flow_positions(surfaces, e, k) - xy, flows.backward, hw, kind, delta)
    num = (f_term * flows.forward_mask).sum() + (b_term * flows.backward_mask).sum()
    den = flows.forward_mask.sum() + flows.backward_mask.sum()
    return num / _or_one(den)


def tracking_loss(surfaces, e, k, tracks: Sequence[OTracks], hw, kind="huber", delta=0.01) -> Tensor:
    """flowmap/loss/loss_tracking.py:28-61 (unweighted): one global ratio over all segments."""
    num = 0
    den = 0
    for seg in tracks:
        f = seg.xy.shape[1]
        s = seg.start_frame
        tgt, vis = track_positions(surfaces[:, s : s + f], e[:, s : s + f], k[:, s : s + f], seg)
        term = robust(tgt, seg.xy[:, None], hw, kind, delta) * vis
        num = num + term.sum()
        den = den + vis.sum()
    return num / _or_one(den)


# --------------------------------------------------------------------------------------
# Caller glue (model.py:54-90, intrinsics/common.py:6-20, extrinsics_procrustes.py:23-59,
# backbone_explicit_depth.py:34-41) — enough to restate one optimisation step
# --------------------------------------------------------------------------------------


def focal_to_k(focal: Tensor, hw) -> Tensor:
    """flowmap/model/intrinsics/common.py:6-20 (``focal_lengths_to_intrinsics``)."""
    h, w = hw
    f = focal * (h * w) ** 0.5
    k = torch.eye(3, dtype=focal.dtype, device=focal.device)
    k[:2, 2] = 0.5
    k = k.broadcast_to(*f.shape, 3, 3).contiguous()
    k[..., 0, 0] = f / w
    k[..., 1, 1] = f / h
    return k


def procrustes_indices(hw, num_points: Optional[int], device="cpu", randomize=False, generator=None):
    """flowmap/model/extrinsics/extrinsics_procrustes.py:34-51."""
    n = hw[0] * hw[1]
    if num_points is None:
        return torch.arange(n, dtype=torch.int64, device=device)
    if randomize:
        return torch.randint(0, n, (num_points,), dtype=torch.int64, device=device, generator=generator)
    return torch.linspace(0, n - 1, num_points, dtype=torch.int64, device=device)


@dataclass
class OStepOutput:
    depths: Tensor
    surfaces: Tensor
    intrinsics: Tensor
    extrinsics: Tensor
    weights: Tensor


def model_forward(depth: Tensor, weights: Tensor, k: Tensor, flows: OFlows, indices: Tensor) -> OStepOutput:
    """flowmap/model/model.py:54-90 with depth/weights/K already produced by the
    (out-of-scope) backbone and intrinsics modules.  depth (b,F,H,W), weights
    (b,F-1,H,W), k (b,F,3,3)."""
    b, f, h, w = depth.shape
    xy, _ = pixel_grid((h, w), depth.device, depth.dtype)
    surfaces = lift(xy, depth, k[:, :, None, None])
    e = fit_poses(surfaces, flows.backward, weights, indices)
    return OStepOutput(depth, surfaces, k, e, weights)


def explicit_depth_step(
    depth_param: Tensor,
    weight_param: Tensor,
    focal: Tensor,
    flows: OFlows,
    hw,
    *,
    num_points: Optional[int] = 1000,
    tracks: Optional[Sequence[OTracks]] = None,
    flow_weight: float = 1000.0,
    track_weight: float = 100.0,
    kind: str = "huber",
    delta: float = 0.01,
    weight_sensitivity: float = 100.0,
):
    """One optimisation step's forward as ``ModelWrapperOverfit.training_step`` runs it
    (model_wrapper_overfit.py:51-62) with BackboneExplicitDepth
    (backbone_explicit_depth.py:34-41), IntrinsicsRegressed (intrinsics_regressed.py:33-41),
    ExtrinsicsProcrustes and the enabled losses (loss.py:31-47).  Returns
    (total, dict of per-loss values, OStepOutput).  depth_param (F,H,W), weight_param
    (F-1,H,W), focal 0-dim.
    """
    f = depth_param.shape[0]
    depth = depth_param[None]
    weights = (weight_sensitivity * weight_param).sigmoid()[None]
    k = focal_to_k(focal, hw).expand(1, f, 3, 3)
    idx = procrustes_indices(hw, num_points, depth.device)
    out = model_forward(depth, weights, k, flows, idx)
    parts = {"flow": flow_weight * flow_loss(out.surfaces, out.extrinsics, k, flows, hw, kind, delta)}
    if tracks is not None:
        parts["tracking"] = track_weight * tracking_loss(out.surfaces, out.extrinsics, k, tracks, hw, kind, delta)
    total = sum(parts.values())
    return total, parts, out


# --------------------------------------------------------------------------------------
# Seeded synthetic inputs.  Pure functions of (seed, sizes).
# --------------------------------------------------------------------------------------


def synth_iid(f: int, h: int, w: int, seed: int = 0, dtype=torch.float32):
    """i.i.d. inputs as used for the survey's CPU baseline: depth
    U(1.10,1.15), flows N(0,0.01²), masks U(0,1), weight logits N(0,0.01²)."""
    g = torch.Generator().manual_seed(seed)
    depth = (1.10 + 0.05 * torch.rand((f, h, w), generator=g)).to(dtype)
    wlogit = (0.01 * torch.randn((f - 1, h, w), generator=g)).to(dtype)
    flows = OFlows(
        (0.01 * torch.randn((1, f - 1, h, w, 2), generator=g)).to(dtype),
        (0.01 * torch.randn((1, f - 1, h, w, 2), generator=g)).to(dtype),
        torch.rand((1, f - 1, h, w), generator=g).to(dtype),
        torch.rand((1, f - 1, h, w), generator=g).to(dtype),
    )
    return depth, wlogit, flows


def synth_scene(f: int, h: int, w: int, seed: int = 0, focal: float = 0.85, depth_noise: float = 0.05, device="cpu"):
    """A geometrically consistent scene: static bumpy surface seen by a smoothly moving
    camera.  Ground-truth depth per frame is produced by fixed-point ray casting
    against a height field; ground-truth flows come from the oracle's own
    reprojection.  Returns dict with gt depth/poses/K, Flows, and a noisy init depth.
    All fp32 outputs (generated in fp64) on the CPU.  ``device``: where the generation runs
    (the full-size fixtures of BASELINE.json configs[1..2] take minutes on host cores and
    seconds on the GPU; the random draws are made on the CPU either way, so a scene is a
    function of (seed, sizes) up to the rounding of the device's elementary functions).
    """
    g = torch.Generator().manual_seed(seed)
    dt = torch.float64
    hw = (h, w)
    k = focal_to_k(torch.tensor(focal, dtype=dt), hw).to(device)
    xy, _ = pixel_grid(hw, device=device, dtype=dt)
    rays = matvec(torch.linalg.inv(k), append_one(xy))  # (h, w, 3), z component 1

    # camera path: small translations + rotations
    t_axis = torch.linspace(0, 1, f, dtype=dt)
    ph = torch.rand(6, generator=g, dtype=dt) * 2 * math.pi
    trans = torch.stack(
        [0.25 * torch.sin(2 * math.pi * t_axis + ph[0]), 0.10 * torch.sin(4 * math.pi * t_axis + ph[1]), 0.15 * t_axis],
        dim=-1,
    )
    ang = torch.stack(
        [0.04 * torch.sin(2 * math.pi * t_axis + ph[2]), 0.06 * torch.sin(2 * math.pi * t_axis + ph[3]), 0.02 * torch.sin(2 * math.pi * t_axis + ph[4])],
        dim=-1,
    )

    def rot(a):
        cx, cy, cz = torch.cos(a)
        sx, sy, sz = torch.sin(a)
        rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=dt)
        ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=dt)
        rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=dt)
        return rz @ ry @ rx

    e = torch.eye(4, dtype=dt).repeat(f, 1, 1)
    for i in range(f):
        e[i, :3, :3] = rot(ang[i])
        e[i, :3, 3] = trans[i]
    e = (torch.linalg.inv(e[0])[None] @ e).to(device)  # first pose = identity, like get_extrinsics

    def height(xw, yw):  # world surface z = height(x, y)
        return 2.0 + 0.25 * torch.sin(1.7 * xw + 0.3) * torch.cos(1.3 * yw - 0.2) + 0.1 * torch.sin(3.1 * xw * yw)

    depth = torch.empty((f, h, w), dtype=dt, device=device)
    for i in range(f):
        r = e[i, :3, :3]
        c = e[i, :3, 3]
        d = torch.full((h, w), 2.0, dtype=dt, device=device)
        for _ in range(40):
            pw = matvec(r, rays * d[..., None]) + c
            # move along the ray so that the point lands on the surface
            err = height(pw[..., 0], pw[..., 1]) - pw[..., 2]
            d = d + err / r[2, 2].clamp_min(0.5)
        depth[i] = d

    surfaces = lift(xy, depth[None], k.expand(1, f, 1, 1, 3, 3))
    kk = k.expand(1, f, 3, 3)
    fwd = forward_flow_positions(surfaces, e[None], kk) - xy
    bwd = backward_flow_positions(surfaces, e[None], kk) - xy

    def inside(pos):
        return ((pos >= 0).all(-1) & (pos < 1).all(-1)).to(dt)

    flows = OFlows(
        fwd.float().cpu(), bwd.float().cpu(), inside(fwd + xy).float().cpu(), inside(bwd + xy).float().cpu()
    )
    del fwd, bwd, surfaces
    smooth = tnf.interpolate(
        torch.randn((1, f, max(h // 16, 2), max(w // 16, 2)), generator=g, dtype=dt).to(device), size=(h, w), mode="bilinear",
        align_corners=False,
    )[0]
    init = depth * (1 + depth_noise * smooth)
    return {
        "depth_gt": depth.float().cpu(),
        "depth_init": init.float().cpu(),
        "extrinsics_gt": e.float().cpu(),
        "intrinsics_gt": k.float().cpu(),
        "focal": focal,
        "flows": flows,
    }


def synth_tracks(
    f: int, h: int, w: int, scene=None, seed: int = 0, interval: int = 5, radius: int = 20, grid: int = 35, p_visible: float = 0.9
):
    """Track segments laid out as ``generate_video_tracks`` does
    (flowmap/tracking/__init__.py:49-70): a segment around every ``interval``-th frame,
    ±radius, ``grid``² query points on the middle frame.  With a ``scene`` the tracks
    are oracle projections of the true surface points; otherwise i.i.d. jitter around
    the query grid.  visibility = in-frame ∧ Bernoulli(p_visible).
    """
    g = torch.Generator().manual_seed(seed + 1000)
    out = []
    for mid in range(0, f, interval):
        s, e_ = max(0, mid - radius), min(f, mid + radius + 1)
        n = e_ - s
        lin = (torch.arange(grid, dtype=torch.float32) + 0.5) / grid
        q = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), dim=-1).reshape(-1, 2)  # (P, 2) xy
        if scene is not None:
            hw = (h, w)
            kk = scene["intrinsics_gt"].double()
            ee = scene["extrinsics_gt"].double()
            xy, _ = pixel_grid(hw, dtype=torch.float64)
            surf = lift(xy, scene["depth_gt"].double()[None, mid : mid + 1], kk.expand(1, 1, 1, 1, 3, 3))
            pts = bilinear_border(surf, q.double()[None, None])  # (1,1,P,3)
            rel = torch.linalg.inv(ee[s:e_]) @ ee[mid]
            xyt = warp_points(pts[0, 0][None], rel[:, None], kk[None, None]).float()  # (n, P, 2)
        else:
            drift = 0.003 * torch.randn((n, q.shape[0], 2), generator=g).cumsum(0)
            xyt = q[None] + drift - drift[mid - s]
        inside = (xyt >= 0).all(-1) & (xyt < 1).all(-1)
        vis = inside & (torch.rand(inside.shape, generator=g) < p_visible)
        out.append(OTracks(xyt[None].contiguous(), vis[None].contiguous(), s))
    return out


def _tap_pixels(xy01: Tensor, h: int, w: int):
    """The (row, col) of the <= 4 pixels ``grid_sample(bilinear, border, align_corners=False)``
    reads for normalised coordinates xy01 (..., 2), as (rows (..., 4), cols (..., 4), inside (..., 4))."""
    ix = (xy01[..., 0].double() * w - 0.5).clamp(0, w - 1)
    iy = (xy01[..., 1].double() * h - 0.5).clamp(0, h - 1)
    x0, y0 = ix.floor().long(), iy.floor().long()
    cols = torch.stack([x0, x0 + 1, x0, x0 + 1], dim=-1)
    rows = torch.stack([y0, y0, y0 + 1, y0 + 1], dim=-1)
    return rows, cols, (cols < w) & (rows < h)


def procrustes_touched(hw, indices: Tensor, bwd_flow: Tensor) -> Tensor:
    """Pixels of dL/ddepth the Procrustes fit writes to (projection.py:226-242): the sampled pixel
    of every later frame and the bilinear taps of its flowed position in the earlier frame.
    bwd_flow (1, F-1, H, W, 2) -> bool (F, H, W).  For the masked comparisons of the parity tests."""
    h, w = hw
    pairs = bwd_flow.shape[1]
    mask = torch.zeros((pairs + 1, h * w), dtype=torch.bool)
    xy, _ = pixel_grid(hw)
    where = (xy + bwd_flow[0].cpu().float()).reshape(pairs, h * w, 2)[:, indices.cpu()]
    rows, cols, inside = _tap_pixels(where, h, w)
    flat = (rows * w + cols).clamp(0, h * w - 1)
    for i in range(pairs):
        mask[i + 1, indices.cpu()] = True
        mask[i, flat[i][inside[i]]] = True
    return mask.reshape(pairs + 1, h, w)


def tracks_touched(hw, frames: int, tracks: Sequence[OTracks]) -> Tensor:
    """Pixels of dL/ddepth the tracking loss writes to (projection.py:266-272): the bilinear taps of
    every visible, in-frame track point.  -> bool (F, H, W)."""
    h, w = hw
    mask = torch.zeros((frames, h * w), dtype=torch.bool)
    for seg in tracks:
        xy = seg.xy[0].cpu().float()  # (f, P, 2)
        live = seg.visibility[0].cpu() & (xy >= 0).all(-1) & (xy < 1).all(-1)
        rows, cols, inside = _tap_pixels(xy, h, w)
        flat = (rows * w + cols).clamp(0, h * w - 1)
        keep = inside & live[..., None]
        for j in range(xy.shape[0]):
            mask[seg.start_frame + j, flat[j][keep[j]]] = True
    return mask.reshape(frames, h, w)


# --------------------------------------------------------------------------------------
# ATE (flowmap/misc/ate.py:7-25): RMS over all coordinates after scipy's Procrustes
# alignment (translation, uniform scale, rotation/reflection; both sets normalised).
# --------------------------------------------------------------------------------------


def ate(gt_positions: Tensor, predicted_positions: Tensor) -> float:
    from scipy import spatial

    a, b, _ = spatial.procrustes(gt_positions.detach().cpu().double().numpy(), predicted_positions.detach().cpu().double().numpy())
    return float(((torch.tensor(a, dtype=torch.float32) - torch.tensor(b, dtype=torch.float32)) ** 2).mean() ** 0.5)


# --------------------------------------------------------------------------------------
# IntrinsicsSoftmin candidate sweep (flowmap/model/intrinsics/intrinsics_softmin.py:85-131)
# --------------------------------------------------------------------------------------


def softmin_intrinsics(depths: Tensor, weights: Tensor, bwd_flow: Tensor, candidates: Tensor, indices: Tensor, hw) -> Tensor:
    """depths (b,>=2,H,W), weights (b,>=1,H,W), bwd_flow (b,>=1,H,W,2), focal candidates
    (n), indices (P) -> softmin-blended intrinsics (b,3,3).  Every candidate aligns frames
    0/1 by Procrustes on the sampled points and is scored by the weighted L1 error of the
    pose-induced backward flow (:92-121); softmin with temperature 10 (:123-127)."""
    b = depths.shape[0]
    n = candidates.shape[0]
    h, w = hw
    k = focal_to_k(candidates.to(depths.dtype), hw)  # (n,3,3)
    xy, _ = pixel_grid(hw, depths.device, depths.dtype)
    rep = lambda t: t[:, None].expand(b, n, *t.shape[1:]).reshape(b * n, *t.shape[1:])  # noqa: E731
    surfaces = lift(xy, rep(depths[:, :2]), k[None, :, None, None, None].expand(b, n, 2, 1, 1, 3, 3).reshape(b * n, 2, 1, 1, 3, 3))
    e = fit_poses(surfaces, rep(bwd_flow[:, :1]), rep(weights[:, :1]), indices)
    pts = surfaces.reshape(b * n, 2, h * w, 3)[:, :, indices]
    kk = k[None, :, None].expand(b, n, 2, 3, 3).reshape(b * n, 2, 3, 3)
    back = backward_flow_positions(pts, e, kk).reshape(b, n, -1, 2)
    flow = back - xy.reshape(h * w, 2)[indices]
    flow_gt = bwd_flow[:, :1].reshape(b, 1, h * w, 2)[:, :, indices]
    wts = weights[:, :1].reshape(b, 1, h * w, 1)[:, :, indices]
    err = ((flow - flow_gt) * wts).abs().sum(dim=(2, 3))
    soft = tnf.softmin((err - err.min(dim=1, keepdim=True).values) * 10, dim=1)
    return (k[None] * soft[:, :, None, None]).sum(dim=1)


# --------------------------------------------------------------------------------------
# Flow post-processing — flowmap/flow/flow_predictor.py:39-102
# --------------------------------------------------------------------------------------


def consistency_mask(videos: Tensor, flow: Tensor) -> Tensor:
    """flow_predictor.py:59-80.  videos (b,f,3,h,w), flow (b,f-1,h,w,2) -> (b,f-1,h,w):
    (1 - max_c |source - bilinear_zeros(target, xy + flow)|)^8."""
    b, f, _, h, w = videos.shape
    src = videos[:, :-1].reshape(b * (f - 1), 3, h, w)
    tgt = videos[:, 1:].reshape(b * (f - 1), 3, h, w)
    xy, _ = pixel_grid((h, w), videos.device, videos.dtype)
    pos = xy + flow.reshape(b * (f - 1), h, w, 2)
    sampled = tnf.grid_sample(tgt, pos * 2 - 1, mode="bilinear", padding_mode="zeros", align_corners=False)
    delta = (src - sampled).abs().max(dim=1).values
    return ((1 - delta) ** 8).reshape(b, f - 1, h, w)


def resize_bilinear(x: Tensor, shape) -> Tensor:
    """F.interpolate(bilinear, align_corners=False) over the last two dims of (n,c,h,w)
    (flow_predictor.py:45,55)."""
    return tnf.interpolate(x, tuple(shape), mode="bilinear", align_corners=False)


def cropped_shapes(original_hw, image_shape, patch_size: int, multiplier: int = 1):
    """get_image_shape + compute_patch_cropped_shape (flowmap/misc/cropping.py:31-40,85-96):
    -> (resize target, patch-cropped shape), both times ``multiplier`` for the flow network's input
    (cropping.py:114-125)."""
    if isinstance(image_shape, tuple):
        h, w = image_shape
    else:
        oh, ow = original_hw
        scale = (image_shape / (oh * ow)) ** 0.5
        h, w = round(oh * scale), round(ow * scale)
    h, w, patch = h * multiplier, w * multiplier, patch_size * multiplier
    return (h, w), ((h // patch) * patch, (w // patch) * patch)


def crop_and_resize(videos: Tensor, intrinsics, image_shape, patch_size: int, multiplier: int = 1):
    """crop_and_resize_batch_for_model (multiplier 1) / _for_flow (cropping.py:99-125): bilinear
    resize of (b,f,3,h,w) videos, centre crop to whole patches, and the matching change of the
    normalised focal lengths (cropping.py:54-70)."""
    b, f, c, h, w = videos.shape
    resized, cropped = cropped_shapes((h, w), image_shape, patch_size, multiplier)
    full = resize_bilinear(videos.reshape(b * f, c, h, w), resized).reshape(b, f, c, *resized)
    row, col = (resized[0] - cropped[0]) // 2, (resized[1] - cropped[1]) // 2
    out = full[..., row : row + cropped[0], col : col + cropped[1]]
    if intrinsics is not None:
        intrinsics = intrinsics.clone()
        intrinsics[..., 0, 0] *= resized[1] / cropped[1]
        intrinsics[..., 1, 1] *= resized[0] / cropped[0]
    return out, intrinsics, resized


def bidirectional_flows(videos: Tensor, predictor, shape) -> OFlows:
    """flow_predictor.py:82-102 around an arbitrary ``predictor(videos) -> raw flow``."""

    def one_direction(v):
        raw = predictor(v)
        b, p, h, w, _ = raw.shape
        mask = consistency_mask(v, raw)
        flow = resize_bilinear(raw.permute(0, 1, 4, 2, 3).reshape(b * p, 2, h, w), shape).reshape(b, p, 2, *shape).permute(0, 1, 3, 4, 2)
        return flow, resize_bilinear(mask.reshape(b * p, 1, h, w), shape).reshape(b, p, *shape)

    fwd, fwd_mask = one_direction(videos)
    bwd, bwd_mask = one_direction(videos.flip(dims=(1,)))
    return OFlows(fwd, bwd.flip(dims=(1,)), fwd_mask, bwd_mask.flip(dims=(1,)))


def synth_video(f: int, h: int, w: int, seed: int = 0) -> Tensor:
    """Smooth random video (1,f,3,h,w) in [0,1] whose frames drift slowly, so that a
    difference-based stand-in predictor yields small, spatially varying flows."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((f * 3, 1, max(h // 6, 2), max(w // 6, 2)), generator=g)
    base = tnf.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
    return base.reshape(1, f, 3, h, w).contiguous()


def standin_predictor(videos: Tensor) -> Tensor:
    """Deterministic stand-in for the optical-flow network (RAFT is out of scope): element-wise
    ops only, so CPU and GPU agree bit for bit.  (b,f,3,h,w) -> (b,f-1,h,w,2)."""
    d = (videos[:, 1:, :2] - videos[:, :-1, :2]) * 0.25 + (videos[:, :-1, 2:3] - 0.5) * 0.05
    return d.permute(0, 1, 3, 4, 2).contiguous()


# --------------------------------------------------------------------------------------
# Export — flowmap/export/colmap.py:86-101
# --------------------------------------------------------------------------------------


def world_point_cloud(depths: Tensor, k: Tensor, extrinsics: Tensor, colors: Tensor):
    """depths (F,H,W), k (F,3,3), extrinsics (F,4,4), colors (F,3,H,W) -> (F·H·W,3) world
    points and (F·H·W,3) colours, frame after frame."""
    f, h, w = depths.shape
    xy, _ = pixel_grid((h, w), depths.device, depths.dtype)
    pts, cols = [], []
    for i in range(f):
        xyz = lift(xy, depths[i], k[i])
        world = matvec(extrinsics[i], append_one(xyz))[..., :3]
        pts.append(world.reshape(h * w, 3))
        cols.append(colors[i].permute(1, 2, 0).reshape(h * w, 3))
    return torch.cat(pts), torch.cat(cols)
