"""Generate tests/golden/*.npz by running the REFERENCE itself (dcharatan/flowmap,
mounted read-only at /root/reference) on seeded synthetic inputs.

Runs only in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference ships no tests, so these fixtures ARE the parity pin:
``tests/test_oracle_golden.py`` checks ``oracle/flowmap_oracle.py`` against them, and
the ``-m gpu`` tests check the HIP path against both.  Inputs are stored next to the
outputs so the fixtures are self-contained.  Everything is computed by the reference's
own functions in fp32 (its native precision); a second copy of the scalar results is
produced in fp64 (same code, inputs up-cast, hard-coded fp32 ``eye`` patched) to record
the reference's own noise floor.
"""

from __future__ import annotations

import os
import sys
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("FLOWMAP_REFERENCE", "/root/reference"))
sys.path[:0] = [str(HERE / "refstubs"), str(REF), str(HERE.parent)]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from flowmap.dataset.types import Batch  # noqa: E402
from flowmap.flow.flow_predictor import Flows  # noqa: E402
from flowmap.loss import get_losses  # noqa: E402
from flowmap.loss.loss_flow import LossFlowCfg  # noqa: E402
from flowmap.loss.loss_tracking import LossTrackingCfg  # noqa: E402
from flowmap.loss.mapping import get_mapping  # noqa: E402
from flowmap.loss.mapping.mapping_huber import MappingHuberCfg  # noqa: E402
from flowmap.loss.mapping.mapping_l1 import MappingL1Cfg  # noqa: E402
from flowmap.loss.mapping.mapping_l2 import MappingL2Cfg  # noqa: E402
from flowmap.model import projection as rp  # noqa: E402
from flowmap.model.backbone.backbone_explicit_depth import BackboneExplicitDepthCfg  # noqa: E402
from flowmap.model.extrinsics.extrinsics_procrustes import ExtrinsicsProcrustesCfg  # noqa: E402
from flowmap.model.intrinsics.intrinsics_regressed import IntrinsicsRegressedCfg  # noqa: E402
from flowmap.model.model import Model, ModelCfg  # noqa: E402
from flowmap.model.procrustes import align_rigid  # noqa: E402
from flowmap.tracking.track_predictor import Tracks  # noqa: E402

from oracle import flowmap_oracle as orc  # noqa: E402  (input generators only)

OUT = Path(os.environ.get("FLOWMAP_GOLDEN_OUT", HERE.parent / "tests" / "golden"))
OUT.mkdir(parents=True, exist_ok=True)
torch.set_num_threads(8)


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    np.savez_compressed(OUT / f"{name}.npz", **{k: npy(v) for k, v in arrays.items()})
    size = (OUT / f"{name}.npz").stat().st_size
    print(f"  wrote {name}.npz ({size / 1024:.1f} KiB)")


def mapping_cfg(kind, delta=0.01):
    return {"huber": MappingHuberCfg("huber", delta), "l1": MappingL1Cfg("l1"), "l2": MappingL2Cfg("l2")}[kind]


def run_step(depth, wlogit, focal, flows, hw, num_points, tracks=None, kind="huber", dtype=torch.float32):
    """The reference's Model + losses, driven like ModelWrapperOverfit.training_step."""
    f = depth.shape[0]
    cfg = ModelCfg(
        BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
        IntrinsicsRegressedCfg("regressed", float(focal)),
        ExtrinsicsProcrustesCfg("procrustes", num_points, False),
        True,
    )
    model = Model(cfg, num_frames=f, image_shape=hw)
    model.backbone.depth.data = depth.clone()
    model.backbone.weights.data = wlogit.clone()
    if dtype == torch.float64:
        model = model.double()
    batch = Batch(torch.zeros((1, f, 3, *hw), dtype=dtype), torch.arange(f)[None], ["s"], ["d"])
    rflows = Flows(*(x.to(dtype) for x in (flows.forward, flows.backward, flows.forward_mask, flows.backward_mask)))
    rtracks = None
    loss_cfgs = [LossFlowCfg(0, 1000.0, "flow", mapping_cfg(kind))]
    if tracks is not None:
        rtracks = [Tracks(t.xy.to(dtype), t.visibility, t.start_frame) for t in tracks]
        loss_cfgs.append(LossTrackingCfg(0, 100.0, "tracking", mapping_cfg(kind)))
    losses = get_losses(loss_cfgs)
    out = model(batch, rflows, 0)
    parts = [fn(batch, rflows, rtracks, out, 0) for fn in losses]
    total = sum(parts)
    total.backward()
    return {
        "total": total,
        "loss_flow": parts[0],
        "loss_tracking": parts[1] if tracks is not None else torch.zeros(()),
        "extrinsics": out.extrinsics,
        "intrinsics": out.intrinsics,
        "g_depth": model.backbone.depth.grad,
        "g_wlogit": model.backbone.weights.grad,
        "g_focal": model.intrinsics.focal_length.grad,
    }


class fp64_reference:
    """Patch the reference's hard-coded fp32 identities (procrustes.py:46,48,
    projection.py:204, intrinsics/common.py:14) so the same code runs in fp64."""

    def __enter__(self):
        self._eye = torch.eye

        def eye(*a, **k):
            if k.get("dtype") == torch.float32:
                k["dtype"] = torch.float64
            return self._eye(*a, **k)

        torch.eye = eye

        # sample_image_grid always yields fp32 (projection.py:104-111); up-cast it in
        # every module that bound the name at import time.
        import flowmap.loss.loss_flow as lf
        import flowmap.model.model as mm

        self._grid = rp.sample_image_grid

        def grid64(*a, **k):
            xy, ij = self._grid(*a, **k)
            return xy.double(), ij

        self._mods = (rp, lf, mm)
        for m in self._mods:
            m.sample_image_grid = grid64
        return self

    def __exit__(self, *exc):
        torch.eye = self._eye
        for m in self._mods:
            m.sample_image_grid = self._grid


def tracks_arrays(tracks):
    d = {"n_segments": np.int64(len(tracks))}
    for i, t in enumerate(tracks):
        d[f"trk{i}_xy"] = t.xy
        d[f"trk{i}_vis"] = t.visibility
        d[f"trk{i}_start"] = np.int64(t.start_frame)
    return d


def gold_steps():
    print("step fixtures")
    # 1. i.i.d. inputs, flow loss only, P=100 subsample  (C0/C1 shape family, tiny)
    f, h, w = 8, 24, 32
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=11)
    r32 = run_step(depth, wlogit, 0.85, flows, (h, w), 100)
    with fp64_reference():
        r64 = run_step(depth.double(), wlogit.double(), 0.85, flows, (h, w), 100, dtype=torch.float64)
    save(
        "step_iid_flow",
        depth=depth, wlogit=wlogit, focal=np.float32(0.85), num_points=np.int64(100),
        fwd=flows.forward, bwd=flows.backward, fwd_mask=flows.forward_mask, bwd_mask=flows.backward_mask,
        **{k: v for k, v in r32.items()}, **{f"f64_{k}": v for k, v in r64.items()},
    )

    # 2. consistent scene, flow + tracking, all pixels in Procrustes (num_points=None)
    f, h, w = 7, 20, 28
    sc = orc.synth_scene(f, h, w, seed=5)
    tracks = orc.synth_tracks(f, h, w, scene=sc, seed=5, interval=3, radius=2, grid=6)
    wlogit = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(3))
    r32 = run_step(sc["depth_init"], wlogit, 0.8, sc["flows"], (h, w), None, tracks)
    with fp64_reference():
        r64 = run_step(sc["depth_init"].double(), wlogit.double(), 0.8, sc["flows"], (h, w), None, tracks, dtype=torch.float64)
    fl = sc["flows"]
    save(
        "step_scene_flow_tracking",
        depth=sc["depth_init"], wlogit=wlogit, focal=np.float32(0.8), num_points=np.int64(-1),
        fwd=fl.forward, bwd=fl.backward, fwd_mask=fl.forward_mask, bwd_mask=fl.backward_mask,
        **tracks_arrays(tracks), **r32, **{f"f64_{k}": v for k, v in r64.items()},
    )

    # 3. i.i.d. inputs, flow + tracking with i.i.d. tracks, l1 and l2 mappings, odd sizes
    f, h, w = 5, 13, 17
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=21)
    tracks = orc.synth_tracks(f, h, w, scene=None, seed=2, interval=2, radius=2, grid=5)
    for kind in ("l1", "l2"):
        r32 = run_step(depth, wlogit, 0.7, flows, (h, w), 50, tracks, kind=kind)
        save(
            f"step_iid_{kind}_odd",
            depth=depth, wlogit=wlogit, focal=np.float32(0.7), num_points=np.int64(50),
            fwd=flows.forward, bwd=flows.backward, fwd_mask=flows.forward_mask, bwd_mask=flows.backward_mask,
            **tracks_arrays(tracks), **r32,
        )


def rand_pose(g, n, scale=0.2):
    a = scale * torch.randn((n, 3), generator=g)
    th = a.norm(dim=-1, keepdim=True).clamp_min(1e-9)
    ax = a / th
    kx = torch.zeros((n, 3, 3))
    kx[:, 0, 1], kx[:, 0, 2], kx[:, 1, 0] = -ax[:, 2], ax[:, 1], ax[:, 2]
    kx[:, 1, 2], kx[:, 2, 0], kx[:, 2, 1] = -ax[:, 0], -ax[:, 1], ax[:, 0]
    r = torch.eye(3) + torch.sin(th)[..., None] * kx + (1 - torch.cos(th))[..., None] * (kx @ kx)
    e = torch.eye(4).repeat(n, 1, 1)
    e[:, :3, :3] = r
    e[:, :3, 3] = scale * torch.randn((n, 3), generator=g)
    return e


def gold_functions():
    print("function-level fixtures")
    g = torch.Generator().manual_seed(77)

    # general K (skew, off-centre principal point) per frame
    def rand_k(b, f):
        k = torch.eye(3).repeat(b, f, 1, 1)
        k[..., 0, 0] = 0.8 + 0.3 * torch.rand((b, f), generator=g)
        k[..., 1, 1] = 0.9 + 0.3 * torch.rand((b, f), generator=g)
        k[..., 0, 1] = 0.02 * torch.randn((b, f), generator=g)
        k[..., 0, 2] = 0.5 + 0.05 * torch.randn((b, f), generator=g)
        k[..., 1, 2] = 0.5 + 0.05 * torch.randn((b, f), generator=g)
        return k

    # --- sample_image_grid / unproject ------------------------------------------------
    b, f, h, w = 2, 3, 6, 9
    xy, ij = rp.sample_image_grid((h, w))
    z = 1 + torch.rand((b, f, h, w), generator=g)
    k = rand_k(b, f)
    z.requires_grad_(True)
    k.requires_grad_(True)
    surf = rp.unproject(xy, z, k[:, :, None, None])
    cot = torch.randn(surf.shape, generator=g)
    (surf * cot).sum().backward()
    save("fn_unproject", xy=xy, ij=ij, z=z, k=k, cot=cot, surfaces=surf, g_z=z.grad, g_k=k.grad)

    # --- compute_forward_flow / compute_backward_flow ---------------------------------
    surfaces = surf.detach().clone().requires_grad_(True)
    e = rand_pose(g, b * f).reshape(b, f, 4, 4).requires_grad_(True)
    k2 = k.detach().clone().requires_grad_(True)
    out_f = rp.compute_forward_flow(surfaces, e, k2)
    out_b = rp.compute_backward_flow(surfaces, e, k2)
    cf = torch.randn(out_f.shape, generator=g)
    cb = torch.randn(out_b.shape, generator=g)
    ((out_f * cf).sum() + (out_b * cb).sum()).backward()
    save(
        "fn_flow_positions",
        surfaces=surfaces, extrinsics=e, intrinsics=k2, cot_f=cf, cot_b=cb, xy_fwd=out_f, xy_bwd=out_b,
        g_surfaces=surfaces.grad, g_extrinsics=e.grad, g_intrinsics=k2.grad,
    )

    # 1-D grid variant (intrinsics_softmin.py:105-109 passes (bn, f, P, 3))
    s1 = torch.randn((2, 3, 11, 3), generator=g) + torch.tensor([0, 0, 3.0])
    out1 = rp.compute_backward_flow(s1, e.detach(), k2.detach())
    save("fn_flow_positions_1d", surfaces=s1, extrinsics=e, intrinsics=k2, xy_bwd=out1)

    # --- project_camera_space edge cases ----------------------------------------------
    pts = torch.tensor(
        [[0.3, -0.2, 2.0], [0.1, 0.1, -1e-5], [0.0, 0.0, -1e-5], [1.0, 2.0, -0.5], [0.2, 0.1, 1e-30], [-0.4, 0.3, 0.0]],
        dtype=torch.float32,
    )
    kk = rand_k(1, 1)[0, 0]
    save("fn_project_edge", points=pts, k=kk, xy=rp.project_camera_space(pts, kk))

    # --- reproject_points with broadcasting -------------------------------------------
    xyz = torch.randn((4, 7, 3), generator=g) + torch.tensor([0, 0, 2.5])
    rel = rand_pose(g, 4)[:, None]
    save("fn_reproject", xyz=xyz, rel=rel, k=kk, xy=rp.reproject_points(xyz, rel, kk))

    # --- project (world → image) --------------------------------------------------------
    ext = rand_pose(g, 4)[:, None]
    pxy, front = rp.project(xyz, ext, kk)
    save("fn_project", xyz=xyz, extrinsics=ext, k=kk, xy=pxy, in_front=front)

    # --- get_extrinsics -------------------------------------------------------------------
    rel = rand_pose(g, 2 * 6).reshape(2, 6, 4, 4).requires_grad_(True)
    chain = rp.get_extrinsics(rel)
    cc = torch.randn(chain.shape, generator=g)
    (chain * cc).sum().backward()
    save("fn_get_extrinsics", rel=rel, cot=cc, extrinsics=chain, g_rel=rel.grad)

    # --- align_rigid: generic, reflection-prone (noisy, near-planar), zero weights ----
    cases = {}
    for name, n_pts, noise, planar in (("generic", 40, 0.01, False), ("noisy_planar", 12, 0.3, True), ("few", 4, 0.05, False)):
        p = torch.randn((5, n_pts, 3), generator=g)
        if planar:
            p[..., 2] *= 0.01
        t = rand_pose(g, 5, scale=0.8)
        q = (t[:, None, :3, :3] @ p[..., None])[..., 0] + t[:, None, :3, 3] + noise * torch.randn(p.shape, generator=g)
        wt = torch.rand((5, n_pts), generator=g)
        p.requires_grad_(True)
        q.requires_grad_(True)
        wt.requires_grad_(True)
        res = align_rigid(p, q, wt)
        cot = torch.randn(res.shape, generator=g)
        (res * cot).sum().backward()
        cases.update({f"{name}_p": p, f"{name}_q": q, f"{name}_w": wt, f"{name}_cot": cot, f"{name}_T": res,
                      f"{name}_g_p": p.grad, f"{name}_g_q": q.grad, f"{name}_g_w": wt.grad})
        # the SAME reference function in fp64 (round 4): what its own fp32 gradients above are worth — the tests hold a gradient to
        # max(1e-4, 2 x the reference's fp32-vs-fp64 gap) of this truth
        with fp64_reference():
            p64, q64, w64 = (x.detach().double().requires_grad_(True) for x in (p, q, wt))
            (align_rigid(p64, q64, w64) * cot.double()).sum().backward()
        cases.update({f"{name}_f64_g_p": p64.grad, f"{name}_f64_g_q": q64.grad, f"{name}_f64_g_w": w64.grad})
    save("fn_align_rigid", **cases)

    # --- align_surfaces: b=2, subsampled + repeated indices, flows pushing samples off-image
    b, f, h, w = 2, 4, 10, 12
    xy, _ = rp.sample_image_grid((h, w))
    z = (1.5 + 0.5 * torch.rand((b, f, h, w), generator=g)).requires_grad_(True)
    k3 = rand_k(b, f).requires_grad_(True)
    surfaces = rp.unproject(xy, z, k3[:, :, None, None])
    bflow = 0.08 * torch.randn((b, f - 1, h, w, 2), generator=g)
    wts = torch.rand((b, f - 1, h, w), generator=g).requires_grad_(True)
    idx = torch.randint(0, h * w, (37,), generator=g)
    idx[5] = idx[4]
    idx[0], idx[1] = 0, h * w - 1
    ext = rp.align_surfaces(surfaces, bflow, wts, idx)
    cot = torch.randn(ext.shape, generator=g)
    (ext * cot).sum().backward()
    with fp64_reference():
        z64, k64, w64 = (x.detach().double().requires_grad_(True) for x in (z, k3, wts))
        e64 = rp.align_surfaces(rp.unproject(rp.sample_image_grid((h, w))[0], z64, k64[:, :, None, None]), bflow.double(), w64, idx)
        (e64 * cot.double()).sum().backward()
    save("fn_align_surfaces", z=z, k=k3, bwd_flow=bflow, weights=wts, indices=idx, cot=cot,
         extrinsics=ext, g_z=z.grad, g_k=k3.grad, g_weights=wts.grad, f64_g_z=z64.grad, f64_g_k=k64.grad, f64_g_weights=w64.grad)

    # --- compute_track_flow ---------------------------------------------------------------
    b, f, h, w, p = 1, 5, 9, 11, 14
    z = (1.5 + 0.5 * torch.rand((b, f, h, w), generator=g)).requires_grad_(True)
    k4 = rand_k(b, f).requires_grad_(True)
    xy, _ = rp.sample_image_grid((h, w))
    surfaces = rp.unproject(xy, z, k4[:, :, None, None])
    e4 = rand_pose(g, f, scale=0.05).reshape(1, f, 4, 4).requires_grad_(True)
    txy = torch.rand((b, f, p, 2), generator=g) * 1.2 - 0.1  # some outside the frame
    tvis = torch.rand((b, f, p), generator=g) < 0.8
    xy_t, vis = rp.compute_track_flow(surfaces, e4, k4, Tracks(txy, tvis, 0))
    cot = torch.randn(xy_t.shape, generator=g)
    (xy_t * cot).sum().backward()
    save("fn_track_flow", z=z, k=k4, extrinsics=e4, track_xy=txy, track_vis=tvis, cot=cot,
         xy_target=xy_t, visibility=vis, g_z=z.grad, g_k=k4.grad, g_extrinsics=e4.grad)

    # --- mappings: values + grads, incl. exactly zero residual and the huber knee ------
    a = 0.02 * torch.randn((50, 2), generator=g)
    bb = 0.02 * torch.randn((50, 2), generator=g)
    a[0] = bb[0]  # n == 0
    a[1] = bb[1] + torch.tensor([0.01, 0.0])  # on/near the knee before aspect fix
    m = {}
    for kind in ("huber", "l1", "l2"):
        aa = a.clone().requires_grad_(True)
        val = get_mapping(mapping_cfg(kind)).forward(aa, bb, (9, 16))
        val.sum().backward()
        m[f"{kind}_val"] = val
        m[f"{kind}_g_a"] = aa.grad
    save("fn_mapping", a=a, b=bb, image_shape=np.array([9, 16]), **m)


def gold_flow_preprocess():
    """FlowPredictor.compute_consistency_mask / rescale_* / compute_bidirectional_flow of the
    reference around a deterministic stand-in network (oracle.standin_predictor)."""
    from flowmap.flow.flow_predictor import FlowPredictor

    class StandIn(FlowPredictor):
        def forward(self, videos):
            return orc.standin_predictor(videos)

    arrays = {}
    for tag, (f, h, w, shape) in {"a": (4, 24, 32, (6, 8)), "b": (3, 20, 28, (7, 9)), "c": (3, 12, 16, (12, 16)), "d": (2, 10, 14, (15, 21))}.items():
        videos = orc.synth_video(f, h, w, seed=40 + ord(tag))
        raw = orc.standin_predictor(videos)
        raw[0, 0, 0, 0] = torch.tensor([0.9, -0.7])  # far outside the frame: zeros padding
        raw[0, 0, 1, 1] = torch.tensor([-(1.5 / w), 0.0])  # straddles the left border

        class Fixed(FlowPredictor):
            def forward(self, v, raw=raw, videos=videos):
                return raw if torch.equal(v, videos) else orc.standin_predictor(v)

        flows = Fixed(None).compute_bidirectional_flow(Batch(videos, None, None, None), shape)
        arrays.update({f"{tag}_videos": videos, f"{tag}_raw": raw, f"{tag}_shape": np.array(shape),
                       f"{tag}_mask_full": FlowPredictor.compute_consistency_mask(videos, raw),
                       f"{tag}_forward": flows.forward, f"{tag}_backward": flows.backward,
                       f"{tag}_forward_mask": flows.forward_mask, f"{tag}_backward_mask": flows.backward_mask})
    save("fn_flow_preprocess", **arrays)


CROPPING_CASES = {
    # tag: (frames, h, w, image_shape, flow_scale_multiplier, patch_size)
    "a": (3, 30, 44, (18, 26), 4, 8),
    "b": (2, 37, 53, 400, 2, 4),  # approximate pixel count -> rounded shape
    "c": (2, 24, 32, (24, 32), 1, 8),  # identity resize, no crop
    "d": (2, 16, 20, (27, 35), 3, 5),  # upsampling
}


def gold_cropping():
    """crop_and_resize_batch_for_model / _for_flow of the reference (flowmap/misc/cropping.py)."""
    from flowmap.misc import cropping as rc

    arrays = {}
    for tag, (f, h, w, image_shape, mult, patch) in CROPPING_CASES.items():
        videos = orc.synth_video(f, h, w, seed=70 + ord(tag))
        k = torch.eye(3).repeat(1, f, 1, 1)
        k[..., 0, 0], k[..., 1, 1], k[..., 0, 2], k[..., 1, 2] = 0.8 + 0.05 * (ord(tag) - 97), 1.1, 0.5, 0.5
        cfg = rc.CroppingCfg(image_shape, mult, patch)
        batch = Batch(videos, torch.arange(f)[None], ["s"], ["d"], None, k)
        model_batch, pre_crop = rc.crop_and_resize_batch_for_model(batch, cfg)
        flow_batch = rc.crop_and_resize_batch_for_flow(batch, cfg)
        arrays.update({f"{tag}_videos": videos, f"{tag}_intrinsics": k, f"{tag}_pre_crop": np.array(pre_crop),
                       f"{tag}_model_videos": model_batch.videos, f"{tag}_flow_videos": flow_batch.videos,
                       f"{tag}_model_intrinsics": model_batch.intrinsics, f"{tag}_flow_intrinsics": flow_batch.intrinsics})
    save("fn_cropping", **arrays)


def gold_export():
    """The point-cloud loop of export_to_colmap (flowmap/export/colmap.py:86-101) executed with
    the reference's own unproject / homogenize_points (the module itself needs plyfile, which
    is not installed), and compute_ate (flowmap/misc/ate.py)."""
    from einops import einsum, rearrange

    from flowmap.misc.ate import compute_ate

    sc = orc.synth_scene(4, 14, 18, seed=9)
    depths, ext = sc["depth_gt"], sc["extrinsics_gt"]
    k = sc["intrinsics_gt"].expand(4, 3, 3).contiguous()
    k[2, 0, 2] = 0.47  # an off-centre principal point, as COLMAP intrinsics may have
    colors = torch.rand((4, 3, 14, 18), generator=torch.Generator().manual_seed(9))
    xy, _ = rp.sample_image_grid((14, 18), depths.device)
    points, cols = [], []
    for e, kk, d, rgb in zip(ext, k, depths, colors):
        xyz = rp.homogenize_points(rp.unproject(xy, d, kk))
        xyz = einsum(e, xyz, "i j, ... j -> ... i")[..., :3]
        points.append(rearrange(xyz, "h w xyz -> (h w) xyz"))
        cols.append(rearrange(rgb, "c h w -> (h w) c"))
    g = torch.Generator().manual_seed(10)
    gt = torch.randn((12, 3), generator=g)
    rot = torch.linalg.qr(torch.randn((3, 3), generator=g))[0]
    pred = 2.5 * gt @ rot.T + 0.3 + 0.02 * torch.randn((12, 3), generator=g)
    ate, a_gt, a_pred = compute_ate(gt, pred)
    save("fn_export", depths=depths, intrinsics=k, extrinsics=ext, colors=colors, points=torch.cat(points), point_colors=torch.cat(cols),
         ate_gt=gt, ate_pred=pred, ate=ate, ate_aligned_gt=a_gt, ate_aligned_pred=a_pred)


def gold_softmin():
    """IntrinsicsSoftmin.forward (intrinsics_softmin.py:63-141) with its internal
    torch.randperm replaced by a recorded permutation."""
    print("softmin fixture")
    from flowmap.model.backbone.backbone import BackboneOutput
    from flowmap.model.intrinsics import intrinsics_softmin as ism

    f, h, w = 3, 14, 18
    depth, wlogit, fl = orc.synth_iid(f, h, w, seed=31)
    g = torch.Generator().manual_seed(31)
    perm = torch.randperm(h * w, generator=g)
    cfg = ism.IntrinsicsSoftminCfg("softmin", 96, 0.5, 2.0, 16, None)
    d = depth[None].clone().requires_grad_(True)
    wt = (100 * wlogit).sigmoid()[None].clone().requires_grad_(True)
    batch = Batch(torch.zeros((1, f, 3, h, w)), torch.arange(f)[None], ["s"], ["d"])
    flows = Flows(fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)
    real_randperm = torch.randperm
    torch.randperm = lambda *a, **k: perm.clone()
    try:
        module = ism.IntrinsicsSoftmin(cfg)
        k = module.forward(batch, flows, BackboneOutput(d, wt), 0)
    finally:
        torch.randperm = real_randperm
    cot = torch.arange(9.0).reshape(3, 3) / 9
    (k[0, 0] * cot).sum().backward()
    # the same module in fp64 (round 4): the truth its own fp32 gradients are measured against
    d64, w64 = d.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True)
    flows64 = Flows(fl.forward.double(), fl.backward.double(), fl.forward_mask.double(), fl.backward_mask.double())
    grid32 = ism.sample_image_grid
    torch.randperm = lambda *a, **k: perm.clone()
    ism.sample_image_grid = lambda *a, **k: (lambda xy_ij: (xy_ij[0].double(), xy_ij[1]))(grid32(*a, **k))
    try:
        with fp64_reference():
            k64 = ism.IntrinsicsSoftmin(cfg).double().forward(batch, flows64, BackboneOutput(d64, w64), 0)
    finally:
        torch.randperm = real_randperm
        ism.sample_image_grid = grid32
    assert k64.dtype == torch.float64
    (k64[0, 0] * cot.double()).sum().backward()
    save("fn_softmin", depth=depth, weights=wt, bwd=fl.backward, indices=perm[:96], candidates=module.focal_length_candidates,
         cot=cot, intrinsics=k[0, 0], g_depth=d.grad, g_weights=wt.grad, f64_intrinsics=k64[0, 0], f64_g_depth=d64.grad, f64_g_weights=w64.grad)


if __name__ == "__main__":
    gold_steps()
    gold_functions()
    gold_softmin()
    gold_flow_preprocess()
    gold_export()
    gold_cropping()
    print("done")
