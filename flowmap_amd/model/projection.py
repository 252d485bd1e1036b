"""Drop-in for ``flowmap.model.projection`` (flowmap/model/projection.py) backed by the
HIP kernels in libflowmap_hip.so.

Same public names, positional order, shapes and gradient behaviour as the reference;
every function cites the reference lines it stands in for.  Differences, all opt-in or
invisible to results:

* ``unproject`` may return a :class:`LazySurfaces` (see ``set_lazy_surfaces``): a view
  of (depth, intrinsics) that the fused consumers in this package read directly, so the
  1.66 GB (B,F,H,W,3) tensor of config C1 never touches HBM.  Anything else that
  touches it (torch functions, attribute access, indexing beyond frame slices)
  materialises the real tensor on the spot.
* tensors must be fp32 and live on the GPU; there is no CPU path (RuntimeError).
* flows / track coordinates are constants: gradients w.r.t. them are not implemented.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import _ops
from .._lib import check_device

# --------------------------------------------------------------------------------------
# small tensor helpers (no kernels needed: pure indexing / concatenation)
# --------------------------------------------------------------------------------------


def homogenize_points(points: Tensor) -> Tensor:
    """flowmap/model/projection.py:11-15"""
    return torch.cat([points, torch.ones_like(points[..., :1])], dim=-1)


def homogenize_vectors(vectors: Tensor) -> Tensor:
    """flowmap/model/projection.py:18-22"""
    return torch.cat([vectors, torch.zeros_like(vectors[..., :1])], dim=-1)


earlier = lambda x: x[:, :-1]  # noqa: E731   flowmap/model/projection.py:139
later = lambda x: x[:, 1:]  # noqa: E731     flowmap/model/projection.py:140

_grid_cache: dict = {}


def sample_image_grid(shape: Tuple[int, ...], device: torch.device = torch.device("cpu")):
    """flowmap/model/projection.py:93-113.  Returns (xy float grid, ij int64 indices).

    The grid is a constant of (shape, device); it is built once with the reference's
    arithmetic ((int64 + 0.5) / length in fp32) and cached — the reference rebuilds it
    three times per step.  The fused kernels never read it (they derive coordinates from
    the thread index); it exists for API parity and as the identity token that lets
    ``unproject`` recognise the canonical grid.
    """
    device = torch.device(device)
    key = (tuple(int(s) for s in shape), str(device))
    hit = _grid_cache.get(key)
    if hit is not None:
        return hit
    # Built on the host and copied once: torch's GPU division differs from the CPU's by an
    # ulp on some pixel centres, and the kernels (which derive coordinates from the thread
    # index with an IEEE divide) follow the CPU values the oracle and golden vectors use.
    indices = [torch.arange(length) for length in shape]
    stacked = torch.stack(torch.meshgrid(*indices, indexing="ij"), dim=-1).to(device)
    coords = [(idx + 0.5) / length for idx, length in zip(indices, shape)]
    coords = list(reversed(coords))
    xy = torch.stack(torch.meshgrid(*coords, indexing="xy"), dim=-1).to(device)
    if len(_grid_cache) > 16:
        _grid_cache.clear()
    _grid_cache[key] = (xy, stacked)
    return xy, stacked


# --------------------------------------------------------------------------------------
# lazy surfaces
# --------------------------------------------------------------------------------------

_LAZY = False


def set_lazy_surfaces(enabled: bool) -> None:
    """When enabled, ``unproject(canonical grid, depths (b,f,h,w), K (b,f,1,1,3,3))``
    returns a LazySurfaces instead of writing (b,f,h,w,3) floats to HBM."""
    global _LAZY
    _LAZY = bool(enabled)


def lazy_surfaces_enabled() -> bool:
    return _LAZY


class LazySurfaces:
    """Camera-space surfaces defined by (depths, intrinsics) but not stored.

    Quacks like the (b, f, h, w, 3) tensor for what the hot path's callers read
    (``shape``, ``device``, ``dtype``, ``ndim``, frame slicing ``[:, s:e]``); any other
    use transparently materialises the real tensor (with autograd intact).
    """

    def __init__(self, depths: Tensor, intrinsics: Tensor):
        self.depths = depths  # (b, f, h, w)
        self.intrinsics = intrinsics  # (b, f, 3, 3)
        self._dense: Optional[Tensor] = None

    # -- cheap metadata ---------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size((*self.depths.shape, 3))

    @property
    def device(self):
        return self.depths.device

    @property
    def dtype(self):
        return self.depths.dtype

    @property
    def ndim(self):
        return 5

    def dim(self):
        return 5

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    # -- materialisation ----------------------------------------------------------------
    def materialize(self) -> Tensor:
        if self._dense is None:
            b, f, h, w = self.depths.shape
            xy, _ = sample_image_grid((h, w), self.depths.device)
            self._dense = _unproject_dense(xy, self.depths, self.intrinsics.reshape(b, f, 1, 1, 3, 3))
        return self._dense

    def __getitem__(self, item):
        # frame slices keep laziness: surfaces[:, s:e]  (loss_tracking.py:48-49)
        if (isinstance(item, tuple) and len(item) == 2 and isinstance(item[0], slice) and item[0] == slice(None)
                and isinstance(item[1], slice)):
            return LazySurfaces(self.depths[item], self.intrinsics[item])
        return self.materialize()[item]

    def __getattr__(self, name):
        # anything we did not anticipate: behave like the real tensor
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        conv = lambda a: a.materialize() if isinstance(a, LazySurfaces) else a  # noqa: E731
        args = tuple(conv(a) for a in args)
        kwargs = {k: conv(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    def __repr__(self):
        return f"LazySurfaces(shape={tuple(self.shape)}, device={self.device})"


def _dense(surfaces) -> Tensor:
    return surfaces.materialize() if isinstance(surfaces, LazySurfaces) else surfaces


class LazyWeights:
    """Correspondence weights sigmoid(sensitivity · logits) (BackboneExplicitDepth,
    flowmap/model/backbone/backbone_explicit_depth.py:38-41) that are not stored: only
    the P sampled pixels per pair are ever read (0.1 % at 720p with P = 1000), so
    ``align_surfaces`` applies the sigmoid inside its gather and writes the logit gradient
    directly.  Any other use materialises the real (b, f-1, h, w) tensor."""

    _fm_lazy = "logits"  # (flowmap_amd/_reference.py asks a lazy value for its device through this attribute, without evaluating it)

    def __init__(self, logits: Tensor, sensitivity: float, lazy_slices: bool = True):
        self.logits = logits  # (b, f-1, h, w)
        self.sensitivity = float(sensitivity)
        # frame slices stay lazy only for consumers that understand a LazyWeights (this package's IntrinsicsSoftmin / align_surfaces); the
        # reference's own IntrinsicsSoftmin hands `weights[:, :1]` to einops (intrinsics_softmin.py:120), which needs a real tensor
        self.lazy_slices = bool(lazy_slices)
        self._dense: Optional[Tensor] = None

    @property
    def shape(self):
        return self.logits.shape

    @property
    def device(self):
        return self.logits.device

    @property
    def dtype(self):
        return self.logits.dtype

    def materialize(self) -> Tensor:
        if self._dense is None:
            self._dense = (self.sensitivity * self.logits).sigmoid()
        return self._dense

    @property
    def ndim(self):
        return self.logits.ndim

    def dim(self):
        return self.logits.dim()

    def size(self, i=None):
        return self.logits.shape if i is None else self.logits.shape[i]

    def __getitem__(self, item):
        # frame slices keep laziness: weights[:, :1]  (intrinsics_softmin.py:100,120 read the first pair only)
        if (isinstance(item, tuple) and len(item) == 2 and isinstance(item[0], slice) and item[0] == slice(None)
                and isinstance(item[1], slice)):
            if self.lazy_slices:
                return LazyWeights(self.logits[item], self.sensitivity)
            if self._dense is None:  # the sigmoid of the frames asked for, not of the whole (b, f-1, h, w) tensor
                return (self.sensitivity * self.logits[item]).sigmoid()
        return self.materialize()[item]

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.ones_like, torch.zeros_like, torch.empty_like) and args and isinstance(args[0], LazyWeights):
            # model/model.py:67-68 (`use_correspondence_weights: false`) asks for the shape only: no sigmoid over (f-1, h, w) for that
            return func(args[0].logits.detach(), *args[1:], **kwargs)
        conv = lambda a: a.materialize() if isinstance(a, (LazyWeights, LazySurfaces)) else a  # noqa: E731
        return func(*tuple(conv(a) for a in args), **{k: conv(v) for k, v in kwargs.items()})

    def __repr__(self):
        return f"LazyWeights(shape={tuple(self.shape)}, sensitivity={self.sensitivity})"


class LazyExtrinsics:
    """The camera-to-world chain of ``get_extrinsics`` (flowmap/model/projection.py:187-210) over the fit's relative poses, NOT evaluated yet.

    ``align_surfaces`` returns one while gradients are being recorded (a training step) and nothing has asked for the chain in an earlier
    step: the fused flow loss reads the relative poses (``_fm_relative_poses``), so a flow-only step never chains them — in the fit's launch
    that is the LAST block's one wave working for ~7 us at 149 poses while the rest of the GPU waits (``profiles/r06_fit_microbench.jsonl``).
    Anything else — an attribute, an index, a torch function, this package's tracking loss — evaluates the chain (one launch,
    ``fm_pose_chain_fwd``, differentiable through the fit's poses) and notes on the constant flow tensor that the chain is wanted: from the
    next step on the fit's own launch produces it again, as in rounds 1-5.  Under ``torch.no_grad()`` (validation, ``Model.export``: the
    reference's ``ModelExports`` checks its fields) the module returns the tensor as before; so does the function-level ``align_surfaces``."""

    _fm_lazy = "_rel"  # (flowmap_amd/_reference.py asks a lazy value for its device through this attribute, without evaluating it)

    def __init__(self, rel: Tensor, rel_inv: Tensor, flow_tensor: Optional[Tensor] = None):
        self._rel = rel  # (b, f-1, 4, 4): later camera -> earlier camera, the factors of the chain
        self._fm_relative_poses = (rel_inv, rel)
        self._flow_tensor = flow_tensor
        self._dense: Optional[Tensor] = None

    @property
    def shape(self):
        b, pairs = self._rel.shape[:2]
        return torch.Size((b, pairs + 1, 4, 4))

    @property
    def device(self):
        return self._rel.device

    @property
    def dtype(self):
        return self._rel.dtype

    @property
    def ndim(self):
        return 4

    def dim(self):
        return 4

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def materialize(self) -> Tensor:
        if self._dense is None:
            self._dense = _ops.PoseChain.apply(self._rel)
            self._dense._fm_relative_poses = self._fm_relative_poses
            if self._flow_tensor is not None:  # something reads the chain in this optimisation: the fit's launch chains it from the next step on
                self._flow_tensor.__dict__["_fm_extrinsics_wanted"] = True
        return self._dense

    def __getitem__(self, item):
        return self.materialize()[item]

    def __len__(self):
        return int(self._rel.shape[0])

    def __iter__(self):
        return iter(self.materialize())

    # torch.as_tensor / numpy.asarray of the value (the array protocols: a detached view of the evaluated chain)
    def __dlpack__(self, *args, **kwargs):
        return self.materialize().detach().__dlpack__(*args, **kwargs)

    def __dlpack_device__(self):
        return self._rel.__dlpack_device__()

    def __array__(self, dtype=None, copy=None):
        out = self.materialize().detach().cpu().numpy()
        return out if dtype is None else out.astype(dtype)

    # torch.save / pickle / copy.deepcopy: the VALUE travels — the evaluated chain as a plain, detached tensor
    def __reduce_ex__(self, protocol):
        return self.materialize().detach().__reduce_ex__(protocol)

    def __deepcopy__(self, memo):
        return self.materialize().detach().clone()

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        conv = lambda a: a.materialize() if isinstance(a, (LazyExtrinsics, LazyWeights, LazySurfaces)) else a  # noqa: E731
        return func(*tuple(conv(a) for a in args), **{k: conv(v) for k, v in kwargs.items()})

    def __repr__(self):
        return f"LazyExtrinsics(shape={tuple(self.shape)}, device={self.device}, evaluated={self._dense is not None})"


def _dense_extrinsics(extrinsics):
    return extrinsics.materialize() if isinstance(extrinsics, LazyExtrinsics) else extrinsics


# --------------------------------------------------------------------------------------
# unproject
# --------------------------------------------------------------------------------------


def _pad_shape(shape, nd):
    return (1,) * (nd - len(shape)) + tuple(shape)


def _split_lead(out_shape, *mat_batch_shapes):
    """Leading dims over which the small matrices vary ("groups") vs trailing dims over
    which they are constant ("points")."""
    nd = len(out_shape)
    n_lead = 0
    for i in range(nd):
        if any(_pad_shape(ms, nd)[i] != 1 for ms in mat_batch_shapes):
            n_lead = i + 1
    lead, pts = tuple(out_shape[:n_lead]), tuple(out_shape[n_lead:])
    g = 1
    for s_ in lead:
        g *= s_
    n = 1
    for s_ in pts:
        n *= s_
    return n_lead, lead, pts, g, n


def _flatten_mats(m: Tensor, nd: int, n_lead: int, lead, g: int) -> Tensor:
    r, c = m.shape[-2:]
    ms = _pad_shape(m.shape[:-2], nd)
    return m.reshape(ms[:n_lead] + (r, c)).expand(*lead, r, c).reshape(g, r, c)


def _unproject_dense(coordinates: Tensor, z: Tensor, intrinsics: Tensor) -> Tensor:
    """Materialising unproject through fm_unproject_fwd.  Handles the reference's
    broadcast patterns: coordinates (*#batch, 2), z (*#batch), intrinsics (*#batch, 3, 3)
    where the intrinsics are constant over the trailing ("grid") dims."""
    check_device(coordinates, z, intrinsics)
    out_shape = torch.broadcast_shapes(coordinates.shape[:-1], z.shape, intrinsics.shape[:-2])
    nd = len(out_shape)
    n_lead, lead, pts, g, n = _split_lead(out_shape, intrinsics.shape[:-2])
    k = _flatten_mats(intrinsics, nd, n_lead, lead, g)
    zz = z.expand(out_shape).reshape(g, n)
    cshape = _pad_shape(coordinates.shape[:-1], nd)
    if all(s_ == 1 for s_ in cshape[:n_lead]):
        xy = coordinates.reshape(cshape[n_lead:] + (2,)).expand(*pts, 2).reshape(n, 2)
    else:
        xy = coordinates.expand(*out_shape, 2).reshape(g, n, 2)
    return _ops.Unproject.apply(xy, zz, k).reshape(*out_shape, 3)


def _squeeze_intrinsics(intrinsics: Tensor) -> Tensor:
    """(b,f,1,1,3,3) -> (b,f,3,3) without leaving work for autograd: the caller's own tensor when
    the argument is ``K[:, :, None, None]`` of a contiguous (b,f,3,3) K (model/model.py:69-74 —
    gradients then reach K directly and every consumer of the step sees ONE tensor object),
    otherwise a reshape (a view; indexing ``[:, :, 0, 0]`` would cost two select_backward
    zero-fill + copy launches per step)."""
    b, f = intrinsics.shape[:2]
    base = intrinsics._base
    if (
        base is not None
        and tuple(base.shape) == (b, f, 3, 3)
        and base.is_contiguous()
        and base.data_ptr() == intrinsics.data_ptr()
        and base.requires_grad == intrinsics.requires_grad
    ):
        return base
    return intrinsics.reshape(b, f, 3, 3)


def unproject(coordinates: Tensor, z: Tensor, intrinsics: Tensor):
    """flowmap/model/projection.py:76-90: (K⁻¹·[x,y,1])·z."""
    if (
        _LAZY
        and z.dim() == 4
        and intrinsics.dim() == 6
        and tuple(intrinsics.shape[2:4]) == (1, 1)
        and tuple(intrinsics.shape[:2]) == tuple(z.shape[:2])
        and coordinates is sample_image_grid(tuple(z.shape[2:]), z.device)[0]
    ):
        check_device(z, intrinsics)
        return LazySurfaces(z, _squeeze_intrinsics(intrinsics))
    return _unproject_dense(coordinates, z, intrinsics)


def unproject_dense(coordinates: Tensor, z: Tensor, intrinsics: Tensor) -> Tensor:
    """``unproject`` that always returns a real tensor (never a LazySurfaces)."""
    return _unproject_dense(coordinates, z, intrinsics)


# --------------------------------------------------------------------------------------
# rigid transforms / projection on explicit points
# --------------------------------------------------------------------------------------


def _group_points(points: Tensor, *mats: Tensor):
    """Broadcast points (*batch, d) against matrices (*batch, r, c) and flatten to groups:
    returns (points (G,N,d), [mats (G,r,c)], out_batch_shape)."""
    bshape = torch.broadcast_shapes(points.shape[:-1], *[m.shape[:-2] for m in mats])
    nd = len(bshape)
    n_lead, lead, pts, g, n = _split_lead(bshape, *[m.shape[:-2] for m in mats])
    p = points.expand(*bshape, points.shape[-1]).reshape(g, n, points.shape[-1])
    return p, [_flatten_mats(m, nd, n_lead, lead, g) for m in mats], bshape


def transform_rigid(homogeneous_coordinates: Tensor, transformation: Tensor) -> Tensor:
    """flowmap/model/projection.py:25-30 (tiny generic matvec; kept in torch)."""
    return (transformation @ homogeneous_coordinates.unsqueeze(-1)).squeeze(-1)


def transform_cam2world(homogeneous_coordinates: Tensor, extrinsics: Tensor) -> Tensor:
    """flowmap/model/projection.py:33-38"""
    return transform_rigid(homogeneous_coordinates, _dense_extrinsics(extrinsics))


def transform_world2cam(homogeneous_coordinates: Tensor, extrinsics: Tensor) -> Tensor:
    """flowmap/model/projection.py:41-46"""
    return transform_rigid(homogeneous_coordinates, torch.linalg.inv_ex(_dense_extrinsics(extrinsics), check_errors=False)[0])


_EYE_CACHE: dict = {}


def _eye44(device) -> Tensor:
    key = str(device)
    if key not in _EYE_CACHE:
        _EYE_CACHE[key] = torch.eye(4, dtype=torch.float32, device=device)
    return _EYE_CACHE[key]


def reproject_points(xyz: Tensor, relative_transformations: Tensor, intrinsics: Tensor) -> Tensor:
    """flowmap/model/projection.py:116-134."""
    check_device(xyz, relative_transformations, intrinsics)
    p, (t, k), bshape = _group_points(xyz, relative_transformations, intrinsics)
    return _ops.Reproject.apply(p, t, k).reshape(*bshape, 2)


def project_camera_space(points: Tensor, intrinsics: Tensor, epsilon: float = 1e-5, infinity: float = 1e8) -> Tensor:
    """flowmap/model/projection.py:49-58 (3-D points; the kernel fixes eps=1e-5, inf=1e8
    like every caller in the reference)."""
    if epsilon != 1e-5 or infinity != 1e8 or points.shape[-1] != 3:
        raise NotImplementedError("flowmap_amd.project_camera_space supports 3-D points with epsilon=1e-5, infinity=1e8")
    ident = _eye44(points.device)
    return reproject_points(points, ident, intrinsics)


def project(points: Tensor, extrinsics: Tensor, intrinsics: Tensor, epsilon: float = 1e-5):
    """flowmap/model/projection.py:61-73: world -> image, plus the in-front test."""
    if epsilon != 1e-5:
        raise NotImplementedError("flowmap_amd.project supports epsilon=1e-5")
    inv = torch.linalg.inv_ex(_dense_extrinsics(extrinsics), check_errors=False)[0]
    cam_z = (inv[..., 2, :3] * points).sum(-1) + inv[..., 2, 3]
    return reproject_points(points, inv, intrinsics), cam_z >= 0


# --------------------------------------------------------------------------------------
# flow-induced positions
# --------------------------------------------------------------------------------------


def _flow_positions(surfaces, extrinsics: Tensor, intrinsics: Tensor, forward: bool) -> Tensor:
    surfaces, extrinsics = _dense(surfaces), _dense_extrinsics(extrinsics)
    check_device(surfaces, extrinsics, intrinsics)
    rel_f, rel_b = _ops.RelativePoses.apply(extrinsics)
    b, f = surfaces.shape[:2]
    grid = surfaces.shape[2:-1]
    n = 1
    for s in grid:
        n *= s
    if forward:
        src, rel, k = surfaces[:, :-1], rel_f, intrinsics[:, 1:]
    else:
        src, rel, k = surfaces[:, 1:], rel_b, intrinsics[:, :-1]
    out = _ops.Reproject.apply(src.reshape(b * (f - 1), n, 3), rel.reshape(b * (f - 1), 4, 4), k.reshape(b * (f - 1), 3, 3))
    return out.reshape(b, f - 1, *grid, 2)


def compute_forward_flow(surfaces, extrinsics: Tensor, intrinsics: Tensor) -> Tensor:
    """flowmap/model/projection.py:143-162: positions of all surface points after the
    forward flow induced by the poses.  surfaces (b,f,*grid,3) -> (b,f-1,*grid,2)."""
    return _flow_positions(surfaces, extrinsics, intrinsics, True)


def compute_backward_flow(surfaces, extrinsics: Tensor, intrinsics: Tensor) -> Tensor:
    """flowmap/model/projection.py:165-184."""
    return _flow_positions(surfaces, extrinsics, intrinsics, False)


# --------------------------------------------------------------------------------------
# poses
# --------------------------------------------------------------------------------------


def get_extrinsics(inverse_relative_transformations: Tensor) -> Tensor:
    """flowmap/model/projection.py:187-210: P_0 = I, P_k = P_{k-1}·T_{k-1} — one launch
    instead of a Python loop of F-1 matmuls."""
    check_device(inverse_relative_transformations)
    return _ops.PoseChain.apply(inverse_relative_transformations)


def align_surfaces(surfaces, backward_flows: Tensor, backward_weights: Tensor, indices: Tensor) -> Tensor:
    """flowmap/model/projection.py:213-252: per-pair weighted Procrustes on flow-induced
    correspondences, chained into camera-to-world extrinsics (b,f,4,4)."""
    return _align_surfaces(surfaces, backward_flows, backward_weights, indices, lazy_ok=False)


def _align_surfaces(surfaces, backward_flows: Tensor, backward_weights: Tensor, indices: Tensor, lazy_ok: bool):
    """``lazy_ok`` (ExtrinsicsProcrustes.forward, i.e. Model.forward -> ModelOutput.extrinsics): the chain may come back as a LazyExtrinsics;
    the function-level API above always returns the tensor."""
    idx = indices
    if isinstance(surfaces, LazySurfaces):
        h, w = surfaces.depths.shape[2:]
    else:
        h, w = surfaces.shape[2:4]
    if idx is not None and idx.numel() == h * w and getattr(idx, "_fm_is_arange", False):
        idx = None  # dense: the kernel derives the index from the thread id
    sens = 0.0
    if isinstance(backward_weights, LazyWeights):
        if backward_weights.sensitivity != 0.0:
            backward_weights, sens = backward_weights.logits, backward_weights.sensitivity
        else:
            backward_weights = backward_weights.materialize()
    if isinstance(surfaces, LazySurfaces):
        # the chain only when something reads it (LazyExtrinsics): a training step (gradients recorded) on lazy surfaces, in an optimisation
        # where nothing has asked for the extrinsics so far.  (Inside a hipGraph capture too: a reader inside the captured region becomes part of
        # the graph, and the captured steps of this package — GraphedStep, GraphedShardedStep, training.py — hand no ModelOutput out of it.)
        lazy = (lazy_ok and _ops.options.lazy_extrinsics and torch.is_grad_enabled() and idx is not None and torch.is_tensor(backward_flows)
                and not backward_flows.__dict__.get("_fm_extrinsics_wanted", False))
        rel, rel_inv, extrinsics = _ops.ProcrustesFit.apply_chained(surfaces.depths, surfaces.intrinsics, None, backward_weights, backward_flows, idx, sens,
                                                                    want_extrinsics=not lazy)
        if lazy and extrinsics is None:
            return LazyExtrinsics(rel, rel_inv, backward_flows)
    else:
        rel, rel_inv, extrinsics = _ops.ProcrustesFit.apply_chained(None, None, surfaces, backward_weights, backward_flows, idx, sens)
    if extrinsics is None:  # (dense fits chain the poses in a launch of their own)
        extrinsics = _ops.PoseChain.apply(rel)
    # The per-pair poses the chain was built from: later(E)⁻¹·earlier(E) and its inverse up to
    # rounding (the chain cancels).  The fused flow loss reads them from here
    # instead of re-deriving them from the chain (two launches + three in the backward).
    extrinsics._fm_relative_poses = (rel_inv, rel)
    return extrinsics


# --------------------------------------------------------------------------------------
# tracks
# --------------------------------------------------------------------------------------


def compute_track_flow(surfaces, extrinsics: Tensor, intrinsics: Tensor, tracks):
    """flowmap/model/projection.py:255-298: reproject every track point from every source
    frame into every target frame.  Returns (xy_target (b,fs,ft,p,2), visibility
    (b,fs,ft,p) bool)."""
    surfaces, extrinsics = _dense(surfaces), _dense_extrinsics(extrinsics)
    check_device(surfaces, extrinsics, intrinsics, tracks.xy)
    b, f, h, w, _ = surfaces.shape
    p = tracks.xy.shape[2]
    xyz = _ops.BilinearSample.apply(surfaces.reshape(b * f, h, w, 3), tracks.xy.reshape(b * f, p, 2)).reshape(b, f, p, 3)
    rel = _ops.AllPairsPoses.apply(extrinsics)  # (b, fs, ft, 4, 4) = inv(E_ft) @ E_fs
    pts = xyz[:, :, None].expand(b, f, f, p, 3).reshape(b * f * f, p, 3)
    k = intrinsics[:, None].expand(b, f, f, 3, 3).reshape(b * f * f, 3, 3)
    xy_target = _ops.Reproject.apply(pts, rel.reshape(b * f * f, 4, 4), k).reshape(b, f, f, p, 2)
    xy_source = tracks.xy[:, :, None]
    visibility = tracks.visibility[:, :, None] & tracks.visibility[:, None]
    source_in = (xy_source >= 0).all(dim=-1) & (xy_source < 1).all(dim=-1)
    target_in = (xy_target >= 0).all(dim=-1) & (xy_target < 1).all(dim=-1)
    return xy_target, visibility & source_in & target_in
