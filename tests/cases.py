"""Parity cases shared by the CPU (host test double) and GPU (HIP) test modules: each
takes the device to run flowmap_amd on and compares against the reference's golden
vectors (tests/golden, made by oracle/make_golden.py) and/or the oracle."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_close_or_reference_gap, load_golden, maxerr, t
from flowmap_amd import Tracks
from flowmap_amd.loss.mapping import get_mapping
from flowmap_amd.model import procrustes as fp
from flowmap_amd.model import projection as fm
from helpers import mapping_cfg
from oracle import flowmap_oracle as orc

TOL = 1e-4


def _leaf(a, dev):
    return t(a).to(dev).requires_grad_(True)


def case_grid_and_unproject(dev):
    g = load_golden("fn_unproject")
    h, w = g["xy"].shape[:2]
    xy, ij = fm.sample_image_grid((h, w), dev)
    assert np.array_equal(xy.cpu().numpy(), g["xy"]) and np.array_equal(ij.cpu().numpy(), g["ij"])
    z, k = _leaf(g["z"], dev), _leaf(g["k"], dev)
    s = fm.unproject(xy, z, k[:, :, None, None])
    assert isinstance(s, torch.Tensor)
    (s * t(g["cot"]).to(dev)).sum().backward()
    assert_close(s, g["surfaces"], 1e-6, what="surfaces")
    assert_close(z.grad, g["g_z"], 1e-5, what="g_z")
    assert_close(k.grad, g["g_k"], 1e-5, what="g_k")


def case_flow_positions(dev):
    g = load_golden("fn_flow_positions")
    s, e, k = _leaf(g["surfaces"], dev), _leaf(g["extrinsics"], dev), _leaf(g["intrinsics"], dev)
    f_ = fm.compute_forward_flow(s, e, k)
    b_ = fm.compute_backward_flow(s, e, k)
    ((f_ * t(g["cot_f"]).to(dev)).sum() + (b_ * t(g["cot_b"]).to(dev)).sum()).backward()
    assert_close(f_, g["xy_fwd"], 1e-5, what="xy_fwd")
    assert_close(b_, g["xy_bwd"], 1e-5, what="xy_bwd")
    assert_close(s.grad, g["g_surfaces"], TOL, what="g_surfaces")
    assert_close(e.grad, g["g_extrinsics"], TOL, what="g_extrinsics")
    assert_close(k.grad, g["g_intrinsics"], TOL, what="g_intrinsics")
    g1 = load_golden("fn_flow_positions_1d")  # 1-D grid, as IntrinsicsSoftmin calls it
    b1 = fm.compute_backward_flow(t(g1["surfaces"]).to(dev), t(g1["extrinsics"]).to(dev), t(g1["intrinsics"]).to(dev))
    assert_close(b1, g1["xy_bwd"], 1e-5, what="xy_bwd 1d")


def case_projection_edges(dev):
    g = load_golden("fn_project_edge")  # z = -eps (inf/nan clamps), behind camera, z ~ 0
    out = fm.project_camera_space(t(g["points"]).to(dev), t(g["k"]).to(dev))
    assert torch.isfinite(out).all()
    assert_close(out, g["xy"], 1e-6, what="edge projection")
    g = load_golden("fn_reproject")  # broadcasting (4,7,3) x (4,1,4,4) x (3,3)
    assert_close(fm.reproject_points(t(g["xyz"]).to(dev), t(g["rel"]).to(dev), t(g["k"]).to(dev)), g["xy"], 1e-5, what="reproject")
    g = load_golden("fn_project")
    xy, front = fm.project(t(g["xyz"]).to(dev), t(g["extrinsics"]).to(dev), t(g["k"]).to(dev))
    assert_close(xy, g["xy"], 1e-5, what="project")
    assert np.array_equal(front.cpu().numpy(), g["in_front"])


def case_pose_chain(dev):
    g = load_golden("fn_get_extrinsics")
    rel = _leaf(g["rel"], dev)
    e = fm.get_extrinsics(rel)
    (e * t(g["cot"]).to(dev)).sum().backward()
    assert_close(e, g["extrinsics"], 1e-6, what="extrinsics")
    assert_close(rel.grad, g["g_rel"], 1e-5, what="g_rel")


def case_align_rigid(dev, case):
    g = load_golden("fn_align_rigid")
    p, q, w = _leaf(g[f"{case}_p"], dev), _leaf(g[f"{case}_q"], dev), _leaf(g[f"{case}_w"], dev)
    T = fp.align_rigid(p, q, w)
    (T * t(g[f"{case}_cot"]).to(dev)).sum().backward()
    assert_close(T, g[f"{case}_T"], 1e-5, what="T")
    # gradients: against the REFERENCE's own function evaluated in fp64 (oracle/make_golden.py: fp64_reference), at 1e-4 or twice the gap the
    # reference's fp32 gradients have to it (measured from the fixture: 3e-7 .. 3e-5 on these clouds)
    for mine, name in ((p.grad, "p"), (q.grad, "q"), (w.grad, "w")):
        assert_close_or_reference_gap(mine, g[f"{case}_f64_g_{name}"], g[f"{case}_g_{name}"], TOL, what=f"g_{name}")
    # ... and against the fp64 oracle (the restatement the full-size tests use as truth)
    p64 = t(g[f"{case}_p"]).double().requires_grad_(True)
    q64 = t(g[f"{case}_q"]).double().requires_grad_(True)
    w64 = t(g[f"{case}_w"]).double().requires_grad_(True)
    (orc.rigid_fit(p64, q64, w64) * t(g[f"{case}_cot"]).double()).sum().backward()
    assert_close(p.grad, p64.grad, TOL, what="g_p vs fp64")
    assert_close(q.grad, q64.grad, TOL, what="g_q vs fp64")
    assert_close(w.grad, w64.grad, TOL, what="g_w vs fp64")


def case_align_surfaces(dev, lazy):
    g = load_golden("fn_align_surfaces")  # b=2, repeated indices, samples pushed off-image
    z, k, w = _leaf(g["z"], dev), _leaf(g["k"], dev), _leaf(g["weights"], dev)
    h, wd = z.shape[2:]
    xy, _ = fm.sample_image_grid((h, wd), dev)
    fm.set_lazy_surfaces(lazy)
    try:
        surfaces = fm.unproject(xy, z, k[:, :, None, None])
        assert isinstance(surfaces, fm.LazySurfaces) == lazy
        e = fm.align_surfaces(surfaces, t(g["bwd_flow"]).to(dev), w, t(g["indices"]).to(dev))
    finally:
        fm.set_lazy_surfaces(False)
    (e * t(g["cot"]).to(dev)).sum().backward()
    assert_close(e, g["extrinsics"], 1e-5, what="extrinsics")
    # the golden gradients are the reference's fp32 svd_backward, itself ~1e-4 from the truth on this fixture: the bar is
    # the fp64 oracle at 1e-4 — or, where the fp32 reference is further than that from it, 4x the reference's own gap
    z64, k64, w64 = (t(g[n]).double().requires_grad_(True) for n in ("z", "k", "weights"))
    xy64, _ = orc.pixel_grid((h, wd), dtype=torch.float64)
    e64 = orc.fit_poses(orc.lift(xy64, z64, k64[:, :, None, None]), t(g["bwd_flow"]).double(), w64, t(g["indices"]))
    (e64 * t(g["cot"]).double()).sum().backward()
    assert_close(e, e64, 1e-5, what="extrinsics vs fp64")
    for ours, truth, gold, what in ((z.grad, z64.grad, g["g_z"], "g_z"), (k.grad, k64.grad, g["g_k"], "g_k"), (w.grad, w64.grad, g["g_weights"], "g_weights")):
        assert_close_or_reference_gap(ours, truth, t(gold), TOL, what=what)


def case_track_flow(dev):
    g = load_golden("fn_track_flow")
    z, k, e = _leaf(g["z"], dev), _leaf(g["k"], dev), _leaf(g["extrinsics"], dev)
    h, w = z.shape[2:]
    xy, _ = fm.sample_image_grid((h, w), dev)
    surfaces = fm.unproject(xy, z, k[:, :, None, None])
    tgt, vis = fm.compute_track_flow(surfaces, e, k, Tracks(t(g["track_xy"]).to(dev), t(g["track_vis"]).to(dev), 0))
    (tgt * t(g["cot"]).to(dev)).sum().backward()
    assert_close(tgt, g["xy_target"], 1e-5, what="xy_target")
    assert np.array_equal(vis.cpu().numpy(), g["visibility"])
    assert_close(z.grad, g["g_z"], TOL, what="g_z")
    assert_close(k.grad, g["g_k"], TOL, what="g_k")
    assert_close(e.grad, g["g_extrinsics"], TOL, what="g_extrinsics")


def case_mappings(dev, kind):
    g = load_golden("fn_mapping")  # includes a zero residual and one on the huber knee
    a = _leaf(g["a"], dev)
    val = get_mapping(mapping_cfg(kind)).forward(a, t(g["b"]).to(dev), tuple(int(x) for x in g["image_shape"]))
    val.sum().backward()
    assert_close(val, g[f"{kind}_val"], 1e-6, what="value")
    assert_close(a.grad, g[f"{kind}_g_a"], 1e-6, what="grad")
    assert torch.isfinite(a.grad).all()


def case_flow_loss_batched(dev, lazy):
    """LossFlow with batch size 2 (the pretraining shape, model_wrapper_pretrain.py:60-82):
    poses fitted per batch element, one global masked mean."""
    from flowmap_amd import Batch, Flows, ModelOutput
    from flowmap_amd.loss import LossFlow, LossFlowCfg

    b, f, h, w, p = 2, 4, 12, 16, 60
    g = torch.Generator().manual_seed(5)
    depth = (1.1 + 0.1 * torch.rand((b, f, h, w), generator=g))
    weights = torch.rand((b, f - 1, h, w), generator=g)
    focal = torch.tensor([0.8, 0.9])
    k = orc.focal_to_k(focal, (h, w))[:, None].expand(b, f, 3, 3).contiguous()
    fl = orc.OFlows(
        0.01 * torch.randn((b, f - 1, h, w, 2), generator=g), 0.01 * torch.randn((b, f - 1, h, w, 2), generator=g),
        torch.rand((b, f - 1, h, w), generator=g), torch.rand((b, f - 1, h, w), generator=g),
    )
    idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64)

    # oracle (fp64)
    d64 = depth.double().requires_grad_(True)
    w64 = weights.double().requires_grad_(True)
    k64 = k.double().requires_grad_(True)
    fl64 = orc.OFlows(*(x.double() for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
    o = orc.model_forward(d64, w64, k64, fl64, idx)
    ref = 1000.0 * orc.flow_loss(o.surfaces, o.extrinsics, k64, fl64, (h, w))
    ref.backward()

    # ours
    d, wt, kk = depth.to(dev).requires_grad_(True), weights.to(dev).requires_grad_(True), k.to(dev).requires_grad_(True)
    flows = Flows(fl.forward.to(dev), fl.backward.to(dev), fl.forward_mask.to(dev), fl.backward_mask.to(dev))
    fm.set_lazy_surfaces(lazy)
    try:
        xy, _ = fm.sample_image_grid((h, w), dev)
        surfaces = fm.unproject(xy, d, kk[:, :, None, None])
        ext = fm.align_surfaces(surfaces, flows.backward, wt, idx.to(dev))
        out = ModelOutput(d, surfaces, kk, ext, wt)
        loss = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))(Batch(torch.zeros((b, f, 3, h, w), device=dev)), flows, None, out, 0)
    finally:
        fm.set_lazy_surfaces(False)
    loss.backward()
    assert_close(loss, ref, TOL, what="loss")
    assert_close(ext, o.extrinsics, TOL, what="extrinsics")
    assert_close(d.grad, d64.grad, TOL, what="g_depth")
    assert_close(wt.grad, w64.grad, 3 * TOL, what="g_weights")
    # dL/dK on i.i.d. inputs is a sum that cancels to ~1e-3 of its terms: held to the fp64 truth at 1e-4, or to twice the gap the
    # reference's own fp32 evaluation (the oracle in fp32) has on the same inputs — measured here
    d32, w32, k32 = (x.detach().clone().requires_grad_(True) for x in (depth, weights, k))
    o32 = orc.model_forward(d32, w32, k32, fl, idx)
    (1000.0 * orc.flow_loss(o32.surfaces, o32.extrinsics, k32, fl, (h, w))).backward()
    assert_close_or_reference_gap(kk.grad, k64.grad, k32.grad, TOL, what="g_k")


def case_procrustes_planned_backward(dev):
    """The sparse Procrustes backward switches from atomics to the planned gather when the same
    (indices, flows) come back: same gradients (first step: atomics; later steps: plan), repeatable; index sets with duplicates and per-step indices never get a plan."""
    from flowmap_amd import _ops

    f, h, w, points = 5, 22, 30, 90
    depth, wlogit, of = orc.synth_iid(f, h, w, seed=21)
    of.backward[0, 1, 3, 4] = torch.tensor([0.6, -0.8])  # far outside: taps clamped to the border
    k = torch.eye(3).repeat(1, f, 1, 1)
    k[..., 0, 0], k[..., 1, 1], k[..., :2, 2] = 0.9, 1.15, 0.5
    k, bwd = k.to(dev), of.backward.to(dev)
    cot = torch.randn((1, f - 1, 4, 4), generator=torch.Generator().manual_seed(2)).to(dev)

    def run(indices):
        d = depth[None].to(dev).requires_grad_(True)
        lg = wlogit[None].to(dev).requires_grad_(True)
        kk = k.clone().requires_grad_(True)
        t_bwd, t_fwd = _ops.ProcrustesFit.apply(d, kk, None, lg, bwd, indices, 100.0, 1)
        ((t_bwd + 0.5 * t_fwd) * cot).sum().backward()
        return d.grad, lg.grad, kk.grad

    idx = torch.linspace(0, h * w - 1, points).to(torch.int64).to(dev)
    before = _ops.counters["procrustes_planned"]
    first = run(idx)  # atomics (first sighting)
    assert _ops.counters["procrustes_planned"] == before
    second = run(idx)  # builds the plan
    third = run(idx)
    assert _ops.counters["procrustes_planned"] == before + 2
    for a, b_, c, name in zip(first, second, third, ("g_depth", "g_logits", "g_k")):
        assert_close(b_, a, 2e-5, abs_=1e-7, what=f"planned vs atomic {name}")
        # planned steps have no float atomics on the big tensors (the small fp64 block sums before them
        # could still differ in a last bit between launches)
        assert_close(c, b_, 1e-6, abs_=1e-9, what=f"repeat {name}")
    # round 3: the planned backward is ONE launch (fm_procrustes_bwd_planned, one workgroup per frame); the three launches it
    # replaced (pose-solve backward, per-correspondence pass, planned gather) must give the same gradients
    from flowmap_amd._lib import torch_ops

    torch_ops().set_one_launch_backward(False)
    try:
        three = run(idx)
    finally:
        torch_ops().set_one_launch_backward(True)
    assert _ops.counters["procrustes_planned"] == before + 3
    for c, t3, name in zip(third, three, ("g_depth", "g_logits", "g_k")):
        assert_close(c, t3, 1e-6, abs_=1e-9, what=f"one launch vs three {name}")
    before += 1
    dup = idx.clone()
    dup[1] = dup[0]
    for _ in range(3):
        got = run(dup)
    assert _ops.counters["procrustes_planned"] == before + 2  # duplicates: atomics every time
    for _ in range(3):
        run(idx.clone())  # a new index tensor each step (randomize_points): never planned
    assert _ops.counters["procrustes_planned"] == before + 2
    assert got[0].isfinite().all()


def case_fill_and_sparse_store(dev):
    """fm_fill_zero (any count, bounded grid) and fm_sparse_store (out[g·stride + idx[j]] = values[g][j])."""
    from flowmap_amd._lib import call, ptr, stream_for

    for count, blocks in ((1, 4), (3, 1), (4, 1), (1027, 2), (50_001, 512), (262_144, 3)):
        x = torch.full((count + 8,), 7.0, device=dev)
        call("fm_fill_zero", ptr(x), count, blocks, stream_for(x))
        assert bool((x[:count] == 0).all()) and bool((x[count:] == 7).all()), (count, blocks)
    g = torch.Generator().manual_seed(0)
    groups, points, stride = 5, 37, 211
    idx = torch.randperm(stride, generator=g)[:points].to(dev)
    values = torch.randn((groups, points), generator=g).to(dev)
    out = torch.zeros((groups, stride), device=dev)
    call("fm_sparse_store", ptr(values), ptr(idx), points, groups, stride, ptr(out), stream_for(out))
    want = torch.zeros((groups, stride), device=dev)
    want[:, idx] = values
    assert torch.equal(out, want)


def case_track_scatter_plan(dev):
    """The planned gather (fm_track_scatter_plan + fm_depth_gather) lands exactly where the atomic
    fm_track_scatter does, for visible / invisible / out-of-frame / border-clipped track points and
    for a shard that owns only some source frames; two runs of the gather are bit-identical."""
    from flowmap_amd import _ops
    from flowmap_amd._lib import call, ptr, stream_for
    from helpers import to_tracks

    f, h, w = 7, 20, 28
    otracks = orc.synth_tracks(f, h, w, seed=3, interval=3, radius=2, grid=5, p_visible=0.8)
    otracks[0].xy[0, 0, :3] = torch.tensor([[-0.2, 0.5], [0.5, 1.3], [0.999, 0.001]])  # outside, outside, on the border (clipped taps)
    otracks[1].xy[0, 1, :2] = torch.tensor([[0.5 / w, 0.5 / h], [1 - 0.4 / w, 1 - 0.4 / h]])  # exactly on / beyond the corner pixel centres
    otracks[1].visibility[0, 1, :2] = True
    tracks = to_tracks(otracks, dev)
    g = torch.Generator().manual_seed(5)
    k = torch.eye(3).repeat(f, 1, 1)
    k[:, 0, 0], k[:, 1, 1], k[:, :2, 2] = 0.9, 1.1, 0.5
    kinv = torch.linalg.inv(k).contiguous().to(dev)
    for own, frame0, f_local in ((None, 0, f), ((2, 5), 2, 4)):
        pk = _ops.PackedTracks(tracks, torch.device(dev), own)
        gws = torch.randn((pk.total, 3), generator=g).to(dev)
        flag = ((pk.vis != 0) & (pk.xy >= 0).all(-1) & (pk.xy < 1).all(-1)).to(torch.uint8)
        if own is not None:  # flags of sources this shard does not own stay 0 (track_points never visits them)
            owned = torch.zeros_like(flag)
            for sg, fr in pk.blocks.tolist():
                start, fl, p, off = pk.seg[sg].tolist()
                owned[off + fr * p : off + (fr + 1) * p] = 1
            flag = flag * owned
        scale = torch.tensor([0.37, 1.0], device=dev)
        up = torch.tensor([1.9], device=dev)
        atomic = torch.zeros((f_local, h, w), device=dev)
        call("fm_track_scatter", ptr(gws), ptr(flag), ptr(pk.xy), ptr(pk.vis), ptr(pk.seg), ptr(pk.blocks), pk.nblocks, pk.pmax, ptr(kinv),
             ptr(scale), ptr(up), h, w, frame0, ptr(atomic), stream_for(atomic))
        pixels, first, entries, weights = pk.scatter_plan(h, w)
        assert pk.scatter_plan(h, w)[0] is pixels  # planned once
        assert bool((pixels[1:] > pixels[:-1]).all()) and int(first[-1]) == entries.numel() == weights.numel()
        runs = []
        for _ in range(2):
            out = torch.full((f_local, h, w), 0.25, device=dev)  # the gather ADDS into what is there
            call("fm_depth_gather", ptr(gws), ptr(pixels), ptr(first), ptr(entries), ptr(weights), pixels.numel(), ptr(kinv),
                 ptr(scale), ptr(up), h, w, frame0, ptr(out), stream_for(out))
            runs.append(out)
        assert torch.equal(runs[0], runs[1])
        assert_close(runs[0] - 0.25, atomic, 2e-6, abs_=1e-6, what=f"gather vs atomic scatter (own={own})")
        assert 0 < int((atomic != 0).sum()) <= pixels.numel()  # (taps of weight exactly 0 are planned too)


def case_loss_gating_and_empty_tracks(dev):
    """Loss.forward's enable_after gate (loss.py:39-41) and the degenerate inputs: no track
    segments, all-invisible tracks (valid_sum or 1)."""
    from flowmap_amd import Batch, Flows, ModelOutput, Tracks
    from flowmap_amd.loss import LossTracking, LossTrackingCfg

    f, h, w = 3, 8, 10
    depth, wlogit, fl = orc.synth_iid(f, h, w, seed=2)
    d = depth[None].to(dev).requires_grad_(True)
    k = orc.focal_to_k(torch.tensor(0.85), (h, w)).expand(1, f, 3, 3).contiguous().to(dev)
    flows = Flows(*(x.to(dev) for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
    batch = Batch(torch.zeros((1, f, 3, h, w), device=dev))
    for lazy in (True, False):
        fm.set_lazy_surfaces(lazy)
        try:
            xy, _ = fm.sample_image_grid((h, w), dev)
            surfaces = fm.unproject(xy, d, k[:, :, None, None])
            ext = fm.align_surfaces(surfaces, flows.backward, torch.full((1, f - 1, h, w), 0.5, device=dev), None)
            out = ModelOutput(d, surfaces, k, ext, None)
            fn = LossTracking(LossTrackingCfg(50, 100.0, "tracking", mapping_cfg("huber")))
            gated = fn(batch, flows, None, out, 10)  # before enable_after: tracks may even be None
            assert float(gated) == 0.0 and gated.dtype == torch.float32
            invisible = [Tracks(torch.rand((1, f, 7, 2), device=dev), torch.zeros((1, f, 7), dtype=torch.bool, device=dev), 0)]
            val = fn(batch, flows, invisible, out, 60)
            assert float(val.detach()) == 0.0
            val.backward()
            assert float(fn(batch, flows, [], out, 60)) == 0.0
        finally:
            fm.set_lazy_surfaces(False)


def case_softmin_intrinsics(dev, lazy_weights):
    """The fused IntrinsicsSoftmin candidate sweep against the reference's golden output
    (its torch.randperm replaced by the recorded indices)."""
    from flowmap_amd import Batch, BackboneOutput, Flows
    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftmin, IntrinsicsSoftminCfg

    g = load_golden("fn_softmin")
    depth = t(g["depth"])[None].to(dev).requires_grad_(True)
    f, h, w = depth.shape[1:]
    idx = t(g["indices"]).to(dev)
    module = IntrinsicsSoftmin(IntrinsicsSoftminCfg("softmin", idx.numel(), 0.5, 2.0, int(g["candidates"].shape[0]), None)).to(dev)
    assert torch.allclose(module.focal_length_candidates.cpu(), t(g["candidates"]))
    module._draw_indices = lambda count, device: idx
    weights_ref = t(g["weights"])
    if lazy_weights:
        logits = (torch.logit(weights_ref.double()) / 100).float().to(dev).requires_grad_(True)
        weights = fm.LazyWeights(logits, 100.0)
    else:
        logits = None
        weights = weights_ref.to(dev).requires_grad_(True)
    bwd = t(g["bwd"]).to(dev)
    flows = Flows(bwd, bwd, torch.ones(bwd.shape[:-1], device=dev), torch.ones(bwd.shape[:-1], device=dev))
    k = module(Batch(torch.zeros((1, f, 3, h, w), device=dev)), flows, BackboneOutput(depth, weights), 0)
    assert k.shape == (1, f, 3, 3)
    (k[0, 0] * t(g["cot"]).to(dev)).sum().backward()
    # gradients against the reference module's own fp64 evaluation (oracle/make_golden.py), at 1e-4 or twice the gap of its fp32 gradients
    # (measured from the fixture: 4e-6 on depth, 6.5e-5 on the weights)
    assert_close(k[0, 0], g["intrinsics"], 1e-4, what="intrinsics")
    assert_close_or_reference_gap(depth.grad[0], g["f64_g_depth"][0], g["g_depth"][0], 1e-4, what="g_depth")
    if lazy_weights:
        # the weights were recovered from their logits in fp32 (logit -> sigmoid is not the identity to the last bit): the truth for THESE
        # weights is the chain rule on the fp64 gradient at the sigmoid the kernel evaluates
        sig = torch.sigmoid(100.0 * logits.detach().double().cpu())
        scale = 100 * sig * (1 - sig)
        assert_close_or_reference_gap(logits.grad, t(g["f64_g_weights"]).double() * scale, t(g["g_weights"]).double() * scale, 1e-4, what="g_logits")
    else:
        assert_close_or_reference_gap(weights.grad, g["f64_g_weights"], g["g_weights"], 1e-4, what="g_weights")


def case_softmin_blend(dev):
    """fm_softmin_blend_fwd/bwd (the tail of IntrinsicsSoftmin.forward) against the reference's torch
    formulation on the same errors: softmin weights, blended K for every frame, parked K^-1 and the
    gradient w.r.t. the errors; batch > 1, more candidates than lanes."""
    import torch.nn.functional as F

    from flowmap_amd import _ops
    from flowmap_amd._lib import call, ptr, stream_for
    from flowmap_amd.model.model import focal_lengths_to_intrinsics

    for b, n, frames in ((1, 60, 7), (3, 8, 2), (2, 150, 70)):
        g = torch.Generator().manual_seed(n)
        err = (torch.rand((b, n), generator=g) * 0.4 + 0.3).double()
        err[0, n // 3] = 0.05  # a clear winner in one batch entry, a flat distribution in the others
        cand = focal_lengths_to_intrinsics(torch.linspace(0.5, 2.0, n), (48, 64)).to(dev)
        err_d = err.to(dev)
        soft = torch.empty((b, n), dtype=torch.float32, device=dev)
        k = torch.empty((b, frames, 3, 3), dtype=torch.float32, device=dev)
        kinv = torch.empty_like(k)
        call("fm_softmin_blend_fwd", ptr(err_d), ptr(cand), b, n, frames, ptr(soft), ptr(k), ptr(kinv), stream_for(k))
        e32 = err.float().requires_grad_(True)
        w_ref = F.softmin((e32 - e32.min(dim=1, keepdim=True).values) * 10, dim=1)
        k_ref = (cand.cpu()[None] * w_ref[:, :, None, None]).sum(dim=1)[:, None].expand(b, frames, 3, 3)
        assert_close(soft, w_ref.detach(), 2e-6, what="soft")
        assert_close(k, k_ref.detach(), 2e-6, what="blended K")
        fresh = torch.empty_like(k)
        call("fm_intrinsics_inverse", ptr(k), b * frames, ptr(fresh), stream_for(k))
        assert torch.equal(kinv, fresh)
        cot = torch.randn((b, frames, 3, 3), generator=g)
        (k_ref * cot).sum().backward()
        g_err = torch.empty((b, n), dtype=torch.float32, device=dev)
        cot_d = cot.to(dev)
        call("fm_softmin_blend_bwd", ptr(cot_d), ptr(soft), ptr(cand), b, n, frames, ptr(g_err), stream_for(k))
        assert_close(g_err, e32.grad, 2e-5, abs_=1e-7, what="g_err")


def case_softmin_step(dev, seed=5):
    """A whole optimisation step with the softmin intrinsics module in the model
    (the reference's default for its first 1000 steps) against the fp64 oracle."""
    from flowmap_amd import Batch
    from flowmap_amd.loss import LossFlow, LossFlowCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, Model, ModelCfg
    from helpers import mapping_cfg, to_flows

    f, h, w, p_soft, p_proc, n = 4, 40, 56, 700, 600, 12
    sc = orc.synth_scene(f, h, w, seed=seed)
    depth, oflows = sc["depth_init"], sc["flows"]
    wlogit = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(seed + 1))
    gen = torch.Generator().manual_seed(seed)
    idx = torch.randperm(h * w, generator=gen)[:p_soft]
    fm.set_lazy_surfaces(True)
    try:
        cfg = ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0),
                       IntrinsicsSoftminCfg("softmin", p_soft, 0.5, 2.0, n, None),
                       ExtrinsicsProcrustesCfg("procrustes", p_proc, False))
        model = Model(cfg, num_frames=f, image_shape=(h, w))
        model.backbone.depth.data = depth.clone()
        model.backbone.weights.data = wlogit.clone()
        model = model.to(dev)
        model.intrinsics._draw_indices = lambda count, device: idx.to(device)
        batch = Batch(torch.zeros((1, f, 3, h, w), device=dev))
        out = model(batch, to_flows(oflows, dev), 0)
        loss = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))(batch, to_flows(oflows, dev), None, out, 0)
        sinks = [fm._ops.depth_sink(out.depths), out.backward_correspondence_weights.logits.__dict__["_fm_sink"]]  # the step's sinks of the two parameter views
        loss.backward()
        # depth and weight-logit gradients of the sweep joined the main buffers in place
        assert [s_.leading_in_place() for s_ in sinks] == [1, 1] and [s_.leading_dense() for s_ in sinks] == [0, 0]
    finally:
        fm.set_lazy_surfaces(False)

    dt = torch.float64
    d = depth.to(dt).requires_grad_(True)
    wl = wlogit.to(dt).requires_grad_(True)
    fl = orc.OFlows(*(x.to(dt) for x in (oflows.forward, oflows.backward, oflows.forward_mask, oflows.backward_mask)))
    weights = (100.0 * wl).sigmoid()[None]
    k = orc.softmin_intrinsics(d[None], weights, fl.backward, torch.linspace(0.5, 2.0, n), idx, (h, w))
    k = k[:, None].expand(1, f, 3, 3)
    o = orc.model_forward(d[None], weights, k, fl, orc.procrustes_indices((h, w), p_proc, "cpu"))
    ref = 1000.0 * orc.flow_loss(o.surfaces, o.extrinsics, k, fl, (h, w), "huber", 0.01)
    ref.backward()
    # the same step by the oracle in fp32 = what the reference's arithmetic delivers on these inputs: the softmin over 12
    # candidates and dL/dK behind it amplify rounding, so each quantity is held to 1e-4 of the fp64 truth or to 4x the fp32
    # reference's own measured gap, whichever is larger
    d32 = depth.detach().clone().requires_grad_(True)
    wl32 = wlogit.detach().clone().requires_grad_(True)
    weights32 = (100.0 * wl32).sigmoid()[None]
    k32 = orc.softmin_intrinsics(d32[None], weights32, oflows.backward, torch.linspace(0.5, 2.0, n), idx, (h, w))
    k32 = k32[:, None].expand(1, f, 3, 3)
    o32 = orc.model_forward(d32[None], weights32, k32, oflows, orc.procrustes_indices((h, w), p_proc, "cpu"))
    ref32 = 1000.0 * orc.flow_loss(o32.surfaces, o32.extrinsics, k32, oflows, (h, w), "huber", 0.01)
    ref32.backward()
    assert_close(out.intrinsics[0, 0], k[0, 0].detach(), 1e-4, what="intrinsics")
    assert_close_or_reference_gap(loss, ref.detach(), ref32.detach(), TOL, what="loss")
    assert_close_or_reference_gap(model.backbone.depth.grad, d.grad, d32.grad, TOL, what="g_depth")
    assert_close_or_reference_gap(model.backbone.weights.grad, wl.grad, wl32.grad, TOL, what="g_wlogit")


def case_packed_inputs(dev, hw=(18, 28)):
    """fm_flow_pack_inputs against a numpy re-layout, and the fused loss reading the packed
    copy against the same loss reading the reference layout (identical arithmetic)."""
    import numpy as np
    from flowmap_amd import _ops
    from helpers import run_ours

    h, w = hw
    f = 4
    depth, wlogit, oflows = orc.synth_iid(f, h, w, seed=11)
    focal = 0.85
    ff, fb, mf, mb = (x.to(dev) for x in (oflows.forward, oflows.backward, oflows.forward_mask, oflows.backward_mask))
    packed = _ops.packed_flow_inputs(ff, fb, mf, mb, eager=True)
    if w % 4 != 0:
        assert packed is None  # the reference layout is streamed directly
    else:
        n, quads = h * w, h * w // 4
        chunks = (quads + 63) // 64
        assert packed.shape == (f, chunks, 6, 64, 4)
        want = np.zeros((f, chunks * 64, 6, 4), np.float32)
        for fr in range(f):
            if fr < f - 1:
                want[fr, :quads, 0:2] = ff[0, fr].reshape(quads, 2, 4).cpu().numpy()
                want[fr, :quads, 2] = mf[0, fr].reshape(quads, 4).cpu().numpy()
            if fr > 0:
                want[fr, :quads, 3:5] = fb[0, fr - 1].reshape(quads, 2, 4).cpu().numpy()
                want[fr, :quads, 5] = mb[0, fr - 1].reshape(quads, 4).cpu().numpy()
        want = want.reshape(f, chunks, 64, 6, 4).transpose(0, 1, 3, 2, 4)
        assert np.array_equal(packed.cpu().numpy(), want)
        assert _ops.packed_flow_inputs(ff, fb, mf, mb, eager=True) is packed  # cached per Flows object
        mf.mul_(1.0)  # an in-place edit bumps the version: the copy is rebuilt
        assert _ops.packed_flow_inputs(ff, fb, mf, mb, eager=True) is not packed

    res = {}
    for use in (True, False):
        _ops.options.packed_inputs = use
        _ops.options.pack_on_first_sight = True  # (a single step: the loss would otherwise stream the reference layout both times)
        packs = _ops.counters["flow_packs"]
        try:
            res[use] = run_ours(depth, wlogit, focal, oflows, (h, w), 100, device=dev)
        finally:
            _ops.options.packed_inputs = True
            _ops.options.pack_on_first_sight = False
        # (on the host double `.to("cpu")` hands run_ours the very tensors packed above: the cached copy is then reused)
        assert _ops.counters["flow_packs"] - packs <= (1 if (use and w % 4 == 0) else 0)
    # identical arithmetic per residual; only the order of the float atomics differs run to run
    for key in ("total", "g_depth", "g_wlogit", "g_focal", "extrinsics"):
        assert_close(res[True][key], res[False][key], 1e-5, abs_=1e-9, what=key)


def case_fused_adam(dev, weight_decay=0.0):
    """FusedAdam against torch.optim.Adam (the optimiser the reference constructs,
    model_wrapper_overfit.py:104-105) on CPU: same trajectory, interchangeable state."""
    from flowmap_amd import FusedAdam

    g = torch.Generator().manual_seed(21)
    shapes = [(3, 17, 23), (2, 16, 24), ()]  # odd tail, vector path, 0-dim (focal length)
    init = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (10.0 ** (k % 3 - 1)) for s in shapes] for k in range(12)]
    for gs in grads:  # untouched entries, like the weights outside the Procrustes sample
        gs[0][0] = 0.0
    ref_p = [x.clone().requires_grad_(True) for x in init]
    our_p = [x.clone().to(dev).requires_grad_(True) for x in init]
    ref = torch.optim.Adam(ref_p, lr=3e-3, weight_decay=weight_decay)
    ours = FusedAdam(our_p, lr=3e-3, weight_decay=weight_decay)
    for k, gs in enumerate(grads):
        for p, q, gr in zip(ref_p, our_p, gs):
            p.grad = gr.clone()
            q.grad = gr.clone().to(dev)
        ref.step()
        ours.step()
        if k == 5:  # state round trip through the torch optimiser's format
            sd = ours.state_dict()
            ours = FusedAdam(our_p, lr=1.0)
            ours.load_state_dict(sd)
            assert ours.param_groups[0]["lr"] == 3e-3
    for p, q in zip(ref_p, our_p):
        assert_close(q.detach(), p.detach(), 2e-6, abs_=1e-7, what="param")
    for p, q in zip(ref_p, our_p):
        assert_close(ours.state[q]["exp_avg"], ref.state[p]["exp_avg"], 2e-6, abs_=1e-9, what="exp_avg")
        assert_close(ours.state[q]["exp_avg_sq"], ref.state[p]["exp_avg_sq"], 2e-6, abs_=1e-12, what="exp_avg_sq")
        assert float(ours.state[q]["step"]) == float(ref.state[p]["step"]) == len(grads)
    if weight_decay == 0.0:
        assert torch.equal(our_p[0].detach()[0].cpu(), init[0][0])  # zero gradient forever -> parameter never moves
    # our state loads into torch's Adam
    chk = torch.optim.Adam([x.detach().cpu().clone().requires_grad_(True) for x in our_p], lr=1.0)
    chk.load_state_dict(ours.state_dict())


def case_flow_preprocess(dev, tag):
    """Consistency mask, resize and direction handling of FlowPredictor against the
    reference's own outputs (golden) around the deterministic stand-in network."""
    from flowmap_amd import Batch
    from flowmap_amd.flow import FlowPredictor

    g = load_golden("fn_flow_preprocess")
    videos, raw = t(g[f"{tag}_videos"]).to(dev), t(g[f"{tag}_raw"]).to(dev)
    shape = tuple(int(x) for x in g[f"{tag}_shape"])

    class StandIn(FlowPredictor):
        def forward(self, v):
            return raw if torch.equal(v, videos) else orc.standin_predictor(v)

    tol = 2e-5  # (1-δ)^8 by squaring vs pow, coordinate rounding of the two grid_sample formulas
    assert_close(FlowPredictor.compute_consistency_mask(videos, raw), g[f"{tag}_mask_full"], tol, what="mask_full")
    flows = StandIn(None).compute_bidirectional_flow(Batch(videos), shape)
    for name in ("forward", "backward", "forward_mask", "backward_mask"):
        got = getattr(flows, name)
        assert got.shape == g[f"{tag}_{name}"].shape and got.is_contiguous()
        assert_close(got, g[f"{tag}_{name}"], tol, what=name)
    # the stand-alone helpers the fused path replaces give the same thing
    b, p = raw.shape[:2]
    assert_close(FlowPredictor.rescale_flow(raw, shape), g[f"{tag}_forward"], tol, what="rescale_flow")
    assert_close(FlowPredictor.rescale_mask(FlowPredictor.compute_consistency_mask(videos, raw), shape), g[f"{tag}_forward_mask"], tol,
                 what="rescale_mask")


def case_focal_intrinsics(dev):
    """IntrinsicsRegressed's K in one launch: bit-identical to focal_lengths_to_intrinsics
    (the reference's two roundings) spread over the frames, K^-1 left for the step's consumers,
    and the gradient of autograd through the reference formulation."""
    from flowmap_amd import _ops
    from flowmap_amd.model.model import focal_lengths_to_intrinsics

    for (h, w), lead, rep in (((720, 1280), (), (1, 150)), ((37, 53), (3,), (4,)), ((256, 256), (2, 2), (1, 3))):
        g = torch.Generator().manual_seed(h)
        focal = (0.5 + torch.rand(lead, generator=g)).to(dev).requires_grad_(True)
        k = _ops.focal_intrinsics(focal, rep, (h, w))
        ref_focal = focal.detach().clone().requires_grad_(True)
        want = focal_lengths_to_intrinsics(ref_focal, (h, w))
        want = want.reshape(*lead, *([1] * len(rep)), 3, 3).expand(*lead, *rep, 3, 3)
        assert k.shape == want.shape and k.is_contiguous()
        assert torch.equal(k, want)
        calls = dict(_ops.counters)
        kinv = _ops.intrinsics_inverse(k)  # parked by the same launch
        assert float((kinv.cpu().double() - torch.linalg.inv(k.detach().double().cpu())).abs().max()) < 1e-6
        fresh = torch.empty_like(kinv)
        from flowmap_amd._lib import call, ptr, stream_for
        call("fm_intrinsics_inverse", ptr(k.detach()), k.numel() // 9, ptr(fresh), stream_for(k))
        assert torch.equal(kinv, fresh)
        cot = torch.randn(k.shape, generator=g).to(dev)
        (k * cot).sum().backward()
        (want * cot).sum().backward()
        assert_close(focal.grad, ref_focal.grad, 2e-6, what="g_focal")
        assert calls == dict(_ops.counters)
    # a different K object at the same address is not served the parked inverse
    focal = torch.tensor(0.9, device=dev, requires_grad=True)
    k1 = _ops.focal_intrinsics(focal, (1, 4), (48, 64))
    inv1 = _ops.intrinsics_inverse(k1).clone()
    other = k1.detach().clone()
    other[..., 0, 0] *= 2
    assert not torch.equal(_ops.intrinsics_inverse(other), inv1)
    view = k1[:, :, None, None]
    assert _ops.intrinsics_inverse(view.reshape(1, 4, 3, 3)).data_ptr() == _ops.intrinsics_inverse(k1).data_ptr()


CROPPING_CASES = {  # oracle/make_golden.py: tag -> (image_shape, flow_scale_multiplier, patch_size)
    "a": ((18, 26), 4, 8),
    "b": (400, 2, 4),
    "c": ((24, 32), 1, 8),
    "d": ((27, 35), 3, 5),
}


def case_cropping(dev, tag):
    """crop_and_resize_batch_for_model / _for_flow against the reference's outputs (golden)."""
    from flowmap_amd import Batch
    from flowmap_amd.misc import cropping

    g = load_golden("fn_cropping")
    image_shape, mult, patch = CROPPING_CASES[tag]
    cfg = cropping.CroppingCfg(image_shape, mult, patch)
    videos, k = t(g[f"{tag}_videos"]).to(dev), t(g[f"{tag}_intrinsics"]).to(dev)
    batch = Batch(videos, intrinsics=k)
    model_batch, pre_crop = cropping.crop_and_resize_batch_for_model(batch, cfg)
    flow_batch = cropping.crop_and_resize_batch_for_flow(batch, cfg)
    assert tuple(pre_crop) == tuple(int(x) for x in g[f"{tag}_pre_crop"])
    for name, got in (("model", model_batch), ("flow", flow_batch)):
        want = g[f"{tag}_{name}_videos"]
        assert got.videos.shape == want.shape and got.videos.is_contiguous()
        assert_close(got.videos, want, 2e-6, what=f"{name}_videos")
        assert_close(got.intrinsics, g[f"{tag}_{name}_intrinsics"], 1e-6, what=f"{name}_intrinsics")
    assert torch.equal(batch.intrinsics, k) and batch.videos is videos  # inputs untouched
    # the unfused pieces agree with the fused pass
    resized = cropping.resize_batch(batch, tuple(pre_crop))
    two_step = cropping.patch_crop_batch(resized, patch)
    assert torch.equal(two_step.videos, model_batch.videos)
    assert_close(two_step.intrinsics, model_batch.intrinsics, 1e-6, what="two_step_intrinsics")
    none = cropping.crop_and_resize_batch_for_flow(Batch(videos), cfg)
    assert none.intrinsics is None and torch.equal(none.videos, flow_batch.videos)


def case_export(dev, tmp_path):
    """World-space point cloud (one launch) against the reference's per-frame loop, the PLY
    round trip, and compute_ate."""
    import numpy as np
    from flowmap_amd import export

    g = load_golden("fn_export")
    pts, cols = export.world_point_cloud(t(g["depths"]).to(dev), t(g["intrinsics"]).to(dev), t(g["extrinsics"]).to(dev), t(g["colors"]).to(dev))
    assert_close(pts, g["points"], 2e-6, what="points")
    assert torch.equal(cols.cpu(), t(g["point_colors"]))
    only, none = export.world_point_cloud(t(g["depths"]).to(dev), t(g["intrinsics"]).to(dev), t(g["extrinsics"]).to(dev))
    assert none is None and torch.equal(only, pts)

    path = tmp_path / "points3D.ply"
    export.write_ply(path, pts.cpu().numpy(), cols.cpu().numpy())
    head = path.read_bytes()[:400].split(b"end_header\n")[0].decode()
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n" % pts.shape[0])
    assert head.rstrip().endswith("property uchar red\nproperty uchar green\nproperty uchar blue")
    xyz, rgb = export.read_ply(path)
    assert np.array_equal(xyz, pts.cpu().numpy())
    assert np.array_equal(rgb, (cols.cpu().numpy() * 255).astype(np.uint8) / 255.0)

    ate, a_gt, a_pred = export.compute_ate(t(g["ate_gt"]).to(dev), t(g["ate_pred"]).to(dev))
    assert abs(float(ate) - float(g["ate"])) < 1e-7 and ate.device.type == torch.device(dev).type
    assert_close(a_gt, g["ate_aligned_gt"], 1e-6, what="aligned gt")
    assert_close(a_pred, g["ate_aligned_pred"], 1e-6, what="aligned predicted")


def case_random_subset(dev):
    """fm_random_subset: distinct, in range, reproducible per seed, different across seeds, and
    plausibly uniform (chi-square over 64 buckets; first-position histogram over many seeds)."""
    from flowmap_amd import _ops

    n, k = 921600, 8192
    a = _ops.random_subset(n, k, dev, seed=1234)
    assert a.dtype == torch.int64 and a.shape == (k,) and a.device.type == torch.device(dev).type
    assert int(a.min()) >= 0 and int(a.max()) < n and a.unique().numel() == k
    assert torch.equal(a, _ops.random_subset(n, k, dev, seed=1234))
    b = _ops.random_subset(n, k, dev, seed=1235)
    assert (a == b).float().mean().item() < 0.01
    # order is random too: about half of the successive differences are positive
    assert abs((a[1:] > a[:-1]).float().mean().item() - 0.5) < 0.03
    counts = torch.bincount((a.cpu() * 64) // n, minlength=64).double()
    chi2 = float(((counts - k / 64) ** 2 / (k / 64)).sum())
    assert chi2 < 120.0, chi2  # 63 dof: mean 63, 99.99th percentile ~ 115
    # small domains and count == n (a full permutation), incl. non-power-of-two sizes
    for m in (1, 2, 7, 64, 100, 1000):
        full = _ops.random_subset(m, m, dev, seed=m)
        assert torch.equal(full.sort().values.cpu(), torch.arange(m))
    firsts = torch.stack([_ops.random_subset(10, 1, dev, seed=s) for s in range(400)]).flatten().cpu()
    hist = torch.bincount(firsts, minlength=10).double()
    assert float(((hist - 40) ** 2 / 40).sum()) < 40.0  # 9 dof
    torch.manual_seed(7)
    c = _ops.random_subset(n, 16, dev)
    torch.manual_seed(7)
    assert torch.equal(c, _ops.random_subset(n, 16, dev))


def case_capturable_pieces(dev):
    """The two host values a captured step cannot carry — Adam's step number and the sampler's
    seed — live in device memory in capturable mode and behave as their host-side versions."""
    from flowmap_amd import FusedAdam, _ops

    g = torch.Generator().manual_seed(5)
    init = torch.randn((3, 10, 12), generator=g)
    grads = [torch.randn((3, 10, 12), generator=g) for _ in range(5)]
    a = init.clone().to(dev).requires_grad_(True)
    b = init.clone().to(dev).requires_grad_(True)
    plain, capt = FusedAdam([a], lr=1e-2), FusedAdam([b], lr=1e-2, capturable=True)
    for gr in grads:
        a.grad, b.grad = gr.clone().to(dev), gr.clone().to(dev)
        plain.step()
        capt.step()
    assert capt.state[b]["step"].device.type == torch.device(dev).type and float(capt.state[b]["step"]) == 5.0
    assert_close(b.detach(), a.detach(), 1e-6, abs_=1e-8, what="capturable Adam")

    previous = _ops.graph_capturable
    _ops.graph_capturable = True
    try:
        x = _ops.random_subset(5000, 64, dev)
        y = _ops.random_subset(5000, 64, dev)
        assert x.unique().numel() == 64 and y.unique().numel() == 64 and not torch.equal(x, y)  # the state advanced
        z = _ops.random_subset(5000, 64, dev, seed=3)
        assert torch.equal(z, _ops.random_subset(5000, 64, dev, seed=3))  # explicit seeds stay stateless
    finally:
        _ops.graph_capturable = previous


def _random_rigid(n, gen, angle=0.03, shift=0.05):
    """n small rigid 4x4 transforms (fp64)."""
    a = angle * torch.randn((n, 3), generator=gen, dtype=torch.float64)
    zero = torch.zeros(n, dtype=torch.float64)
    skew = torch.stack([zero, -a[:, 2], a[:, 1], a[:, 2], zero, -a[:, 0], -a[:, 1], a[:, 0], zero], dim=-1).reshape(n, 3, 3)
    out = torch.eye(4, dtype=torch.float64).repeat(n, 1, 1)
    out[:, :3, :3] = torch.linalg.matrix_exp(skew)
    out[:, :3, 3] = shift * torch.randn((n, 3), generator=gen, dtype=torch.float64)
    return out


def case_flow_fused_leaves(dev, f, h, w, packed, kind="huber", seed=0, tol=TOL):
    """The fused flow loss on its own, with the relative poses and a GENERAL per-frame K (all nine
    entries free, different per frame) held as leaves: loss, dL/ddepth, per-frame dL/dK (b,F,3,3) and
    dL/dT_fwd / dL/dT_bwd (b,F-1,4,4) against the fp64 oracle's autograd — every FRAME compared on
    its own, so the per-frame reduction chain of the kernel (per-thread fp32 partials -> DPP wave
    sum -> fp64 across waves -> fp64 atomics -> flow_finalize_frame) is pinned, not only dL/dfocal.
    loss_flow.py:31-70, projection.py:76-90,116-134,143-184."""
    from flowmap_amd import _ops

    gen = torch.Generator().manual_seed(seed)
    depth = 1.0 + 0.3 * torch.rand((1, f, h, w), generator=gen, dtype=torch.float64)
    k = orc.focal_to_k(torch.tensor(0.85, dtype=torch.float64), (h, w)).repeat(1, f, 1, 1)
    k = k + 0.02 * torch.randn((1, f, 3, 3), generator=gen, dtype=torch.float64)  # skew, free last row, per-frame
    t_fwd = _random_rigid(f - 1, gen)[None]
    t_bwd = _random_rigid(f - 1, gen)[None]  # an independent leaf, not the inverse of t_fwd
    fl = orc.OFlows(
        0.01 * torch.randn((1, f - 1, h, w, 2), generator=gen, dtype=torch.float64),
        0.01 * torch.randn((1, f - 1, h, w, 2), generator=gen, dtype=torch.float64),
        torch.rand((1, f - 1, h, w), generator=gen, dtype=torch.float64),
        torch.rand((1, f - 1, h, w), generator=gen, dtype=torch.float64),
    )
    weight = 1000.0

    # oracle, fp64 (the fp32 inputs ours sees, widened)
    leaves32 = [x.float() for x in (depth, k, t_fwd, t_bwd)]
    d64, k64, tf64, tb64 = (x.double().requires_grad_(True) for x in leaves32)
    fl64 = orc.OFlows(*(x.float().double() for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
    xy, _ = orc.pixel_grid((h, w), dtype=torch.float64)
    surfaces = orc.lift(xy, d64, k64[:, :, None, None])
    fpos = orc.warp_points(surfaces[:, :-1], tf64[:, :, None, None], k64[:, 1:, None, None])
    bpos = orc.warp_points(surfaces[:, 1:], tb64[:, :, None, None], k64[:, :-1, None, None])
    num = (orc.robust(fpos - xy, fl64.forward, (h, w), kind) * fl64.forward_mask).sum() + (
        orc.robust(bpos - xy, fl64.backward, (h, w), kind) * fl64.backward_mask).sum()
    ref = weight * num / (fl64.forward_mask.sum() + fl64.backward_mask.sum())
    ref.backward()

    # ours
    d, kk, tf, tb = (x.to(dev).requires_grad_(True) for x in leaves32)
    ff, fb, mf, mb = (x.float().to(dev) for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask))
    norm = _ops.flow_valid_norm(mf, mb, weight)
    pk = None
    if packed:
        pk = _ops.packed_flow_inputs(ff, fb, mf, mb, eager=True)
        assert pk is not None, "the packed layout needs width % 4 == 0"
    loss = _ops.FlowLossFused.apply(d, kk, tf, tb, ff, fb, mf, mb, norm, _ops.MAPPING_KINDS[kind], 0.01, False, 0, pk)
    loss.backward()

    assert_close(loss, ref, tol, what="loss")
    assert_close(d.grad, d64.grad, tol, what="dL/ddepth")
    assert maxerr(d.grad, d64.grad) <= 10 * tol, "dL/ddepth: max-abs"
    for fr in range(f):
        assert_close(d.grad[0, fr], d64.grad[0, fr], tol, what=f"dL/ddepth frame {fr}")
        assert_close(kk.grad[0, fr], k64.grad[0, fr], tol, what=f"dL/dK frame {fr}")
    for pr in range(f - 1):
        assert_close(tf.grad[0, pr, :3], tf64.grad[0, pr, :3], tol, what=f"dL/dT_fwd pair {pr}")
        assert_close(tb.grad[0, pr, :3], tb64.grad[0, pr, :3], tol, what=f"dL/dT_bwd pair {pr}")
    assert float(tf.grad[0, :, 3].abs().max()) == 0.0 and float(tb.grad[0, :, 3].abs().max()) == 0.0  # [..., :3] drops the last row


def case_ghost_terms(dev, kind="huber"):
    """fm_flow_ghost_terms (the ghost halo of frame sharding): the dL/ddepth of ONE direction of ONE pair's flow-loss term, added into a
    frame's gradient — the fused flow loss of the whole video restricted to a shard [1 .. F-2] plus the two ghost terms (the backward term
    of pair 0 at frame 1, the forward term of pair F-2 at frame F-2) gives the whole video's dL/ddepth of the shard's boundary frames."""
    from flowmap_amd import _ops
    from flowmap_amd._lib import call, ptr, stream_for

    f, h, w = 5, 12, 16
    gen = torch.Generator().manual_seed(5)
    depth = (1.0 + 0.3 * torch.rand((1, f, h, w), generator=gen)).to(dev)
    k = orc.focal_to_k(torch.tensor(0.85), (h, w)).repeat(1, f, 1, 1).to(dev)  # (shared by all frames, as the ghost halo requires)
    t_fwd, t_bwd = _random_rigid(f - 1, gen)[None].float().to(dev), _random_rigid(f - 1, gen)[None].float().to(dev)
    ff, fb = (0.01 * torch.randn((1, f - 1, h, w, 2), generator=gen)).to(dev), (0.01 * torch.randn((1, f - 1, h, w, 2), generator=gen)).to(dev)
    mf, mb = torch.rand((1, f - 1, h, w), generator=gen).to(dev), torch.rand((1, f - 1, h, w), generator=gen).to(dev)
    weight = 1000.0
    norm = _ops.flow_valid_norm(mf, mb, weight)  # the GLOBAL normaliser, as a shard uses it

    def grad_of(lo, hi):  # the fused flow loss over frames lo..hi (pairs lo..hi-1), normalised globally
        d = depth[:, lo : hi + 1].clone().requires_grad_(True)
        loss = _ops.FlowLossFused.apply(d, k[:, lo : hi + 1].contiguous(), t_fwd[:, lo:hi].contiguous(), t_bwd[:, lo:hi].contiguous(),
                                        ff[:, lo:hi].contiguous(), fb[:, lo:hi].contiguous(), mf[:, lo:hi].contiguous(), mb[:, lo:hi].contiguous(),
                                        norm, _ops.MAPPING_KINDS[kind], 0.01, False, 0, None)
        loss.backward()
        return d.grad[0]

    whole = grad_of(0, f - 1)
    shard = grad_of(1, f - 2).clone()  # frames 1 .. F-2: its boundary frames lack one term each
    assert float((shard[0] - whole[1]).abs().max()) > 1e-3 * float(whole[1].abs().max())
    kinv = _ops.intrinsics_inverse(k)
    scale = (h * w) ** 0.5
    with _ops._guard(depth.device):
        call("fm_flow_ghost_terms", ptr(depth[0, 1]), ptr(t_bwd[0, 0]), ptr(fb[0, 0]), ptr(mb[0, 0]), ptr(shard[0]),
             ptr(depth[0, f - 2]), ptr(t_fwd[0, f - 2]), ptr(ff[0, f - 2]), ptr(mf[0, f - 2]), ptr(shard[-1]),
             ptr(kinv[0, 0]), ptr(k[0, 0]), ptr(norm), None, h, w, _ops.MAPPING_KINDS[kind], 0.01, w / scale, h / scale, stream_for(depth))
    for name, mine, ref in (("first", shard[0], whole[1]), ("last", shard[-1], whole[f - 2]), ("interior", shard[1], whole[2])):
        err = float((mine - ref).abs().max())
        assert err <= 2e-6 * float(ref.abs().max()), (name, err)  # (the same per-pixel arithmetic; the two terms are summed in another order)


def case_dense_procrustes(dev, h, w, flow_sigma, f=4):
    """`num_points: null` (every pixel a correspondence, ablation_explicit_depth.yaml:11-12): the tiled,
    planned, atomic-free dense kernels (pixel-space sums, fm_procrustes_scatter_dense) against the fp64
    oracle's align_rigid on the same correspondences (projection.py:226-249, procrustes.py:7-51) — poses
    and the gradients w.r.t. depth, weight logits and a per-frame K — and against the generic
    gather / atomic kernels (explicit arange indices).  Large flows push samples out of a tile's LDS
    window (global fallback), across several tiles of the static tap lists, and against the image
    border (clamped taps)."""
    from flowmap_amd import _ops

    g = torch.Generator().manual_seed(h * w)
    depth = 1.0 + 0.3 * torch.rand((1, f, h, w), generator=g)
    k = orc.focal_to_k(torch.tensor(0.85), (h, w)).repeat(1, f, 1, 1) + 0.01 * torch.randn((1, f, 3, 3), generator=g)
    flow = flow_sigma * torch.randn((1, f - 1, h, w, 2), generator=g)
    logits = 0.01 * torch.randn((1, f - 1, h, w), generator=g)
    cot_b, cot_f = torch.randn((1, f - 1, 4, 4), generator=g), torch.randn((1, f - 1, 4, 4), generator=g)
    cot_b[..., 3, :] = 0
    cot_f[..., 3, :] = 0

    # fp64 oracle
    d64, k64, l64 = (x.double().requires_grad_(True) for x in (depth, k, logits))
    xy, _ = orc.pixel_grid((h, w), dtype=torch.float64)
    surfaces = orc.lift(xy, d64, k64[:, :, None, None])
    later = surfaces[:, 1:].reshape(1, f - 1, h * w, 3)
    where = (xy + flow.double()).reshape(1, f - 1, h * w, 2)
    earlier = orc.bilinear_border(surfaces[:, :-1], where)
    rel64 = orc.rigid_fit(later, earlier, (100.0 * l64).sigmoid().reshape(1, f - 1, h * w))
    ((rel64 * cot_b.double()).sum() + (torch.linalg.inv(rel64) * cot_f.double()).sum()).backward()

    res = {}
    for name, idx, planned in (("tiled", None, False), ("tiled_planned", None, True), ("generic", torch.arange(h * w, device=dev), False)):
        d, kk, lg = (x.clone().to(dev).requires_grad_(True) for x in (depth, k, logits))
        fl_dev = flow.clone().to(dev)
        _ops.options.dense_plan = planned
        try:
            t_bwd, t_fwd = _ops.ProcrustesFit.apply(d, kk, None, lg, fl_dev, idx, 100.0, 1)
        finally:
            _ops.options.dense_plan = None
        ((t_bwd * cot_b.to(dev)).sum() + (t_fwd * cot_f.to(dev)).sum()).backward()
        res[name] = (t_bwd.detach(), d.grad, lg.grad, kk.grad)
        assert ("_fm_dense_plan" in fl_dev.__dict__) == planned  # the fused pass (the default) builds no lists
        if planned:
            res["plan"] = fl_dev._fm_dense_plan[1]
    # left to itself the facade picks by the flow: i.i.d. flows of a few pixels leave the fused pass's window (planned kernels), a
    # flow that is constant inside the tiles does not (no lists)
    for fl_auto, want_plan in (((0.5 * torch.randn(flow.shape, generator=g)).to(dev), True), (torch.full_like(flow, 0.01).to(dev), False)):
        d, kk, lg = (x.clone().to(dev).requires_grad_(True) for x in (depth, k, logits))
        t_bwd, t_fwd = _ops.ProcrustesFit.apply(d, kk, None, lg, fl_auto, None, 100.0, 1)
        assert ("_fm_dense_plan" in fl_auto.__dict__) == want_plan, (flow_sigma, want_plan)
    truth = (rel64.detach(), d64.grad, l64.grad, k64.grad)
    # the static tap lists: one entry per (later pixel, earlier-frame tile its taps touch), every tile's list ascending
    first, entries = res["plan"]
    first, entries = first.cpu(), entries.cpu().numpy().view(np.uint32)
    assert int(first[-1]) >= (f - 1) * h * w and int(first[-1]) <= 4 * (f - 1) * h * w
    for lo, hi in zip(first[:-1].tolist(), first[1:].tolist()):
        seg = entries[lo:hi]
        assert (seg[1:] > seg[:-1]).all(), "tile list not in ascending pixel order"
        assert ((seg >> 16) < h).all() and ((seg & 0xFFFF) < w).all()
    for mode in ("tiled", "tiled_planned"):
        for a, b, c, what in zip(res[mode], res["generic"], truth, ("t_bwd", "g_depth", "g_logits", "g_k")):
            assert torch.isfinite(a).all(), (mode, what)
            assert_close(a, c, TOL, what=f"{mode}: {what} vs fp64 oracle")
            assert_close(a, b, TOL, abs_=1e-7, what=f"{mode}: {what} vs generic kernels")
        assert maxerr(res[mode][1], truth[1]) <= 10 * TOL, f"{mode}: g_depth: max-abs"
        assert maxerr(res[mode][2], truth[2]) <= 10 * TOL, f"{mode}: g_logits: max-abs"
    assert_close(res["tiled"][1], res["tiled_planned"][1], 1e-6, what="g_depth: fused pass vs planned kernels")
    assert_close(res["tiled"][2], res["tiled_planned"][2], 1e-6, what="g_logits: fused pass vs planned kernels")


def _small_problem(dev, f=5, h=24, w=32, points=60, tracking=True, seed=21, intrinsics=None):
    """(model, batch, flows, loss(out)) of a consistent scene: flow + tracking losses, regressed intrinsics."""
    import flowmap_amd
    from flowmap_amd import Batch
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from helpers import to_flows, to_tracks

    sc = orc.synth_scene(f, h, w, seed=seed)
    flowmap_amd.set_lazy_surfaces(True)
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), intrinsics or IntrinsicsRegressedCfg("regressed", 0.9),
                           ExtrinsicsProcrustesCfg("procrustes", points, False)), num_frames=f, image_shape=(h, w))
    model.backbone.depth.data = sc["depth_init"].clone()
    model.backbone.weights.data = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(seed))
    model = model.to(dev)
    flows = to_flows(sc["flows"], dev)
    tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=seed, interval=2, radius=2, grid=5), dev) if tracking else None
    batch = Batch(torch.zeros((1, f, 3, h, w), device=dev))
    flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", mapping_cfg("huber")))

    def loss_of(out):
        total = flow_fn(batch, flows, None, out, 0)
        return total + track_fn(batch, flows, tracks, out, 0) if tracking else total

    return model, batch, flows, loss_of


def case_tap_exchange(dev):
    """The tap exchange between the fused flow loss and the fused tracking loss (fm_flow_loss_fused_taps / fm_track_loss_fused_fwd_taps /
    fm_tap_grad_apply): from the second step on the tracking loss is evaluated ahead of the flow pass, which absorbs its depth gradient at
    the static taps and leaves the tap depths in a compact image the next evaluation samples from.  Values and every gradient equal the
    plain order's (the exchange switched off) however the two losses reach backward(): summed (no correction launched), scaled
    differently, one of them alone, in two backward calls; and after the depth parameter moved the stale image is not used."""
    import flowmap_amd
    from flowmap_amd import _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from helpers import to_tracks

    f, h, w = 6, 24, 32
    sc = orc.synth_scene(f, h, w, seed=33)
    otracks = orc.synth_tracks(f, h, w, scene=sc, seed=33, interval=2, radius=2, grid=6)

    def roots(lf, lt, how):
        return {"sum": [lf + lt], "scaled": [2.0 * lf + 0.5 * lt], "flow_only": [lf], "track_only": [lt], "two_calls": [lf, lt]}[how]

    def run(exchange: bool, how: str, steps: int = 3, move_at=None, track_kind="huber"):
        _ops.options.tap_exchange = exchange
        model, batch, flows, _ = _small_problem(dev, f=f, h=h, w=w, tracking=False, seed=33)
        tracks = to_tracks(otracks, dev)
        flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", mapping_cfg(track_kind)))
        results, sinks = [], []
        for step in range(steps):
            if move_at is not None and step == move_at:
                with torch.no_grad():
                    model.backbone.depth.mul_(1.01)  # the parameter moves (an optimiser the library does not know): the image is stale
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, step)
            lf = flow_fn(batch, flows, tracks, out, step)
            lt = track_fn(batch, flows, tracks, out, step)
            calls = roots(lf, lt, how)
            for i, root in enumerate(calls):
                root.backward(retain_graph=i + 1 < len(calls))
            sinks.append(_ops.depth_sink(out.depths))
            results.append([x.detach().clone() for x in (lf, lt, out.extrinsics, model.backbone.depth.grad, model.backbone.weights.grad,
                                                         model.intrinsics.focal_length.grad)])
        return results, sinks

    names = ("loss_flow", "loss_tracking", "extrinsics", "g_depth", "g_wlogit", "g_focal")
    min_bytes = _ops.options.tap_exchange_min_bytes
    _ops.options.tap_exchange_min_bytes = 0  # (by default the exchange engages for depth tensors beyond the last-level cache only)
    try:
        for how in ("sum", "scaled", "flow_only", "track_only", "two_calls"):
            before = dict(_ops.counters)
            plain, _ = run(False, how)
            assert _ops.counters["flow_tap_passes"] == before["flow_tap_passes"]
            got, sinks = run(True, how)
            assert _ops.counters["flow_tap_absorbs"] - before["flow_tap_absorbs"] == 2, (how, _ops.counters)  # steps 1 and 2
            assert _ops.counters["track_tap_samples"] - before["track_tap_samples"] == 1, (how, _ops.counters)  # step 2 (the image exists after step 1)
            free, launched = (sum(s.taps_settled()[i] for s in sinks) for i in (0, 1))
            assert (free, launched) == {"sum": (2, 0), "scaled": (0, 2), "flow_only": (0, 2), "track_only": (0, 2), "two_calls": (0, 2)}[how], (how, free, launched)
            for step, (a, b) in enumerate(zip(got, plain)):
                for x, y, what in zip(a, b, names):
                    # same arithmetic per element; the absorbed gradient joins the flow pass's sum in another order (fp32 rounding of one add)
                    err = float((x - y).abs().max())
                    assert err <= 2e-6 * max(float(y.abs().max()), 1e-30), (how, step, what, err, float(y.abs().max()))
            # the tracking part is really there (the comparison is not vacuous): dL/ddepth differs from the flow-only gradient at the taps
            assert float((got[2][3] - run(False, "flow_only", steps=1)[0][0][3]).abs().max()) > 0 or how == "flow_only"
        # a parameter that moved between the steps: the compact image is not sampled from (and the results still agree)
        before = dict(_ops.counters)
        plain, _ = run(False, "sum", steps=4, move_at=2)
        got, _ = run(True, "sum", steps=4, move_at=2)
        assert _ops.counters["track_tap_samples"] - before["track_tap_samples"] == 1, _ops.counters  # step 3 only (step 2 saw a moved parameter)
        for step, (a, b) in enumerate(zip(got, plain)):
            for x, y, what in zip(a, b, names):
                err = float((x - y).abs().max())
                assert err <= 2e-6 * max(float(y.abs().max()), 1e-30), ("moved", step, what, err)
        # an edit behind the version counter (`param.data...`): the flow pass finds the image the tracking loss sampled stale — loudly
        _ops.options.tap_exchange = True
        model, batch, flows, _ = _small_problem(dev, f=f, h=h, w=w, tracking=False, seed=33)
        tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=34, interval=2, radius=2, grid=6), dev)
        flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
        track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", mapping_cfg("huber")))
        raised = False
        for step in range(3):
            if step == 2:
                model.backbone.depth.data.mul_(1.01)
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, step)
            try:
                total = flow_fn(batch, flows, tracks, out, step) + track_fn(batch, flows, tracks, out, step)
            except RuntimeError as exc:
                raised = "version counter" in str(exc)
                break
            total.backward()
        assert raised
        # another robust kernel on the tracking side (the exchange does not depend on it)
        plain, _ = run(False, "sum", track_kind="l1")
        got, _ = run(True, "sum", track_kind="l1")
        for x, y, what in zip(got[-1], plain[-1], names):
            assert float((x - y).abs().max()) <= 2e-6 * max(float(y.abs().max()), 1e-30), ("l1", what)
    finally:
        _ops.options.tap_exchange = True
        _ops.options.tap_exchange_min_bytes = min_bytes
        flowmap_amd.set_lazy_surfaces(False)


def case_step_torch_ops(dev):
    """What a step launches BESIDE the library's own kernels: the sum of the two losses and the two sums autograd forms where two consumers
    meet (dL/d extrinsics, dL/dK) — and nothing else: not even autograd's ones_like for loss.backward() (the losses are RootLoss tensors,
    which seed their backward with the registered ones tensor; with `_ops.options.unit_seed = False` the fill is back).  In particular no zeros tensors
    materialised for the non-differentiable outputs of the custom functions (K^-1 of FocalIntrinsics, scale / totals of the tracking
    loss): each was a fill launch per step until round 3."""
    import flowmap_amd
    from torch.utils._python_dispatch import TorchDispatchMode

    quiet = ("view", "unsqueeze", "squeeze", "detach", "empty", "select.int", "slice", "expand", "reshape", "alias", "permute", "as_strided",
             "t.default", "transpose", "_local_scalar_dense", "lift_fresh", "unbind", "split", "flowmap_amd", "profiler", "_to_copy", "clone")
    for tracking, allowed in ((False, {}), (True, {"aten.add.Tensor": 3}), (None, {"aten.ones_like.default": 1})):
        from flowmap_amd import _ops
        from flowmap_amd._lib import torch_ops

        try:
            _ops.options.unit_seed = tracking is not None  # (None: the plain-tensor path, flow only)
            model, batch, flows, loss_of = _small_problem(dev, tracking=bool(tracking))

            def step():
                model.zero_grad(set_to_none=True)
                loss = loss_of(model(batch, flows, 0))
                assert (type(loss) is _ops.RootLoss) == (tracking is not None)
                loss.backward()

            for _ in range(3):  # plans, packed inputs, arenas exist from the third step on
                step()
            seen = {}
            known_unit = torch_ops().unit_seed_uses()

            class Trace(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    name = str(func)
                    if not any(q in name for q in quiet):
                        seen[name] = seen.get(name, 0) + 1
                    return func(*args, **(kwargs or {}))

            with Trace():
                step()
            assert seen == allowed, (tracking, seen)
            assert torch_ops().unit_seed_uses() - known_unit == int(tracking is not None)  # the flow loss's node knew its seed on the host
        finally:
            _ops.options.unit_seed = True
            flowmap_amd.set_lazy_surfaces(False)


def _grads(model):
    return [p.grad.detach().clone() for p in (model.backbone.depth, model.backbone.weights, model.intrinsics.focal_length)]


def case_lazy_extrinsics(dev):
    """LazyExtrinsics (round 6; flowmap_amd/model/projection.py): while gradients are recorded and nothing has asked for the camera-to-world chain,
    align_surfaces returns it unevaluated — a flow-only step never chains the poses (the fit's launch runs without its last-block chain, the fused
    flow loss reads the relative poses).  Whatever then reads it gets exactly the tensor the fit's own launch would have produced, with the
    same gradients; the first read notes on the flow tensor that the chain is wanted and later steps produce it in the fit's launch again;
    torch.no_grad() and options.lazy_extrinsics = False keep the tensor."""
    import flowmap_amd
    from flowmap_amd import _ops, config
    from flowmap_amd.model.projection import LazyExtrinsics

    try:
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        flows.backward.__dict__.pop("_fm_extrinsics_wanted", None)
        with config.override(lazy_extrinsics=False):  # rounds 1-5: the chain in the fit's launch
            for _ in range(3):  # (the third step: the static plans exist from the second one on — the path every later step takes)
                model.zero_grad(set_to_none=True)
                eager = model(batch, flows, 0)
                assert torch.is_tensor(eager.extrinsics)
                loss_of(eager).backward()
            want, want_ext = _grads(model), eager.extrinsics.detach().clone()
            model.zero_grad(set_to_none=True)
            eager = model(batch, flows, 0)
            (loss_of(eager) + eager.extrinsics[0, :, :3, 3].square().sum()).backward()  # something that differentiates through the chain
            want_chain = _grads(model)
        for _ in range(2):  # flow-only steps: the chain is never evaluated, values and gradients are those of the eager step bit for bit
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, 0)
            assert isinstance(out.extrinsics, LazyExtrinsics) and tuple(out.extrinsics.shape) == tuple(want_ext.shape) and out.extrinsics.device == want_ext.device
            loss_of(out).backward()
            assert out.extrinsics._dense is None and "_fm_extrinsics_wanted" not in flows.backward.__dict__
            for a, b in zip(_grads(model), want):
                assert torch.equal(a, b)
        # a reader: the chain is evaluated (one launch), equal to the fit's own, differentiable through the fit's poses
        model.zero_grad(set_to_none=True)
        out = model(batch, flows, 0)
        (loss_of(out) + out.extrinsics[0, :, :3, 3].square().sum()).backward()
        assert torch.equal(out.extrinsics.materialize().detach(), want_ext) and torch.equal(torch.linalg.inv(out.extrinsics).detach(), torch.linalg.inv(want_ext))
        # the value behaves like the tensor where users touch it: torch.as_tensor, numpy, .to() / .cpu(), torch.save / load, deepcopy, iteration
        import copy
        import io

        import numpy as np

        assert torch.equal(torch.as_tensor(out.extrinsics), want_ext) and np.array_equal(np.asarray(out.extrinsics), want_ext.cpu().numpy())
        assert torch.equal(out.extrinsics.cpu(), want_ext.cpu()) and torch.equal(out.extrinsics.to(want_ext.device).detach(), want_ext) and len(out.extrinsics) == 1
        buffer = io.BytesIO()
        torch.save(out.extrinsics, buffer)
        buffer.seek(0)
        loaded = torch.load(buffer, weights_only=False)
        assert type(loaded) is torch.Tensor and torch.equal(loaded.to(want_ext.device), want_ext)
        copied = copy.deepcopy(out.extrinsics)
        assert type(copied) is torch.Tensor and torch.equal(copied, want_ext) and torch.equal(next(iter(out.extrinsics)).detach(), want_ext[0])
        for a, b in zip(_grads(model), want_chain):
            assert torch.equal(a, b)
        assert flows.backward.__dict__.get("_fm_extrinsics_wanted") is True
        assert torch.is_tensor(model(batch, flows, 0).extrinsics)  # ... and from now on the fit's launch chains the poses again
        flows.backward.__dict__.pop("_fm_extrinsics_wanted")
        with torch.no_grad():  # validation / Model.export (the reference's ModelExports checks its fields): a tensor
            assert torch.is_tensor(model(batch, flows, 0).extrinsics)
        # the tracking loss reads the chain: lazy on the first step only
        model, batch, flows, loss_of = _small_problem(dev, tracking=True)
        flows.backward.__dict__.pop("_fm_extrinsics_wanted", None)
        seen = []
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, 0)
            seen.append(isinstance(out.extrinsics, LazyExtrinsics))
            loss_of(out).backward()
        assert seen == [True, False, False]
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def case_root_loss(dev):
    """RootLoss (flowmap_amd/_ops.py): the fused losses seed their own backward() with the registered ones tensor.  Same gradients as the
    plain-tensor path however the loss reaches backward — directly, summed, scaled by a weight (an ordinary upstream gradient again),
    with an explicit gradient, through torch.autograd.grad, twice through a retained graph; a seed somebody wrote into is not trusted any
    more; and the value behaves like a plain tensor where users touch it (format strings, detach, torch.save / torch.load, deepcopy)."""
    import copy
    import io

    import flowmap_amd
    from flowmap_amd import _ops
    from flowmap_amd._lib import torch_ops

    try:
        model, batch, flows, loss_of = _small_problem(dev, tracking=True)

        def grads_of(run):
            model.zero_grad(set_to_none=True)
            run(loss_of(model(batch, flows, 0)))
            return _grads(model)

        for _ in range(3):
            grads_of(lambda loss: loss.backward())
        _ops.options.unit_seed = False
        plain = grads_of(lambda loss: loss.backward())
        plain3 = grads_of(lambda loss: (3.0 * loss).backward())
        _ops.options.unit_seed = True

        def same(got, want, what):
            for a, b in zip(got, want):
                assert torch.equal(a, b), what

        uses = torch_ops().unit_seed_uses()
        same(grads_of(lambda loss: loss.backward()), plain, "seeded backward")
        assert torch_ops().unit_seed_uses() == uses + 1
        same(grads_of(lambda loss: (3.0 * loss).backward()), plain3, "a weighted loss: an ordinary upstream gradient")
        # a trainer's normalisation by its accumulation factor (Lightning: closure_loss / accumulate_grad_batches): the number 1 is the loss itself
        before = torch_ops().unit_seed_uses()
        same(grads_of(lambda loss: (loss / 1).backward()), plain, "loss / 1")
        assert torch_ops().unit_seed_uses() == before + 1
        probe = loss_of(model(batch, flows, 0))
        assert (probe / 1) is probe and (probe / 1.0) is probe and (probe / 2) is not probe and (probe / torch.ones((), device=probe.device)) is not probe
        del probe
        same(grads_of(lambda loss: loss.backward(gradient=torch.full_like(loss, 3.0))), plain3, "explicit gradient")
        assert torch_ops().unit_seed_uses() == uses + 2  # (the plain backward and loss / 1 above; neither the weighted loss nor the explicit gradient)

        def by_grad(loss):
            params = (model.backbone.depth, model.backbone.weights, model.intrinsics.focal_length)
            for p, g in zip(params, torch.autograd.grad(loss, params)):
                p.grad = g

        same(grads_of(by_grad), plain, "torch.autograd.grad")

        def twice(loss):
            loss.backward(retain_graph=True)
            loss.backward()

        for a, b in zip(grads_of(twice), plain):
            assert_close(a, 2 * b, 1e-5, "two backward calls through a retained graph")

        # somebody scaled the seed in place: the registered tensor's version counter moved, so it is an ordinary gradient from then on
        # (read on the device), and the next backward() makes a fresh one
        seed = _ops.unit_seed(torch.device(dev))
        seed.mul_(3.0)
        same(grads_of(lambda loss: loss.backward(gradient=seed)), plain3, "an edited seed is read, not trusted")
        assert _ops.unit_seed(torch.device(dev)) is not seed and float(_ops.unit_seed(torch.device(dev))) == 1.0
        same(grads_of(lambda loss: loss.backward()), plain, "a fresh seed after the edit")

        loss = loss_of(model(batch, flows, 0))
        assert type(loss) is _ops.RootLoss and type(loss.detach()) is torch.Tensor and type(loss * 2) is _ops.RootLoss
        assert type(torch.stack([loss, loss])) is torch.Tensor
        assert f"{loss:.4f}" == f"{loss.item():.4f}" and f"{loss.detach():.3e}" == f"{loss.item():.3e}"
        buf = io.BytesIO()
        torch.save({"loss": loss.detach(), "live": loss}, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=True)
        assert type(back["live"]) is torch.Tensor and torch.equal(back["loss"].cpu(), loss.detach().cpu())
        assert type(copy.deepcopy(loss.detach())) is torch.Tensor
        with torch.no_grad():
            assert type(loss_of(model(batch, flows, 0))) is torch.Tensor  # nothing to seed without a graph
    finally:
        _ops.options.unit_seed = True
        flowmap_amd.set_lazy_surfaces(False)


def case_grad_arena(dev):
    """The persistent dL/dweights storage (GradArena, csrc/fm_torch.cpp): the same gradients as fresh zeros every
    step; one storage reused from the first planned step on; a gradient that is kept (accumulation without
    zero_grad) or edited in place (clipping) is never written under — the step falls back to a fresh buffer."""
    import flowmap_amd
    from flowmap_amd import _ops

    try:
        history = {}
        for arena in (False, True):
            _ops.options.grad_arena = arena
            model, batch, flows, loss_of = _small_problem(dev, tracking=False)
            steps = []
            for _ in range(5):  # step 1: atomics into zeros; step 2 builds the plan; from then on the arena
                model.zero_grad(set_to_none=True)
                loss_of(model(batch, flows, 0)).backward()
                steps.append(_grads(model))
            history[arena] = (steps, model)
        for step, (a, b) in enumerate(zip(history[False][0], history[True][0])):
            for x, y, what in zip(a, b, ("g_depth", "g_weights", "g_focal")):
                assert_close(y, x, 2e-5, abs_=1e-7, what=f"{what} step {step}")
            assert torch.equal(a[1] != 0, b[1] != 0), step  # the same sparsity pattern: nothing stale left behind
        model = history[True][1]
        arena = model.backbone.weights.__dict__["_fm_arena"]
        assert arena.reused() >= 2 and arena.refilled() == 1
        g_w = history[True][0][-1][1]
        assert 0 < int((g_w != 0).sum()) <= g_w.shape[0] * 60  # P entries per pair at most, zeros elsewhere

        # gradient accumulation: .grad from the previous step is still alive -> a fresh buffer, and autograd's own sum
        model.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 0)).backward()
        once = _grads(model)
        loss_of(model(batch, flows, 0)).backward()  # no zero_grad in between
        twice = _grads(model)
        for x, y, what in zip(once, twice, ("g_depth", "g_weights", "g_focal")):
            assert_close(y, 2 * x, 2e-5, abs_=1e-7, what=f"accumulated {what}")
        # an in-place edit of the gradient (clipping) moves its version counter: the next step starts from zeros again
        model.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 0)).backward()
        refills = arena.refilled()
        model.backbone.weights.grad.add_(1.0)  # junk everywhere, as a careless in-place op would leave it
        model.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 0)).backward()
        assert arena.refilled() == refills + 1
        assert_close(model.backbone.weights.grad, once[1], 2e-5, abs_=1e-7, what="g_weights after an in-place edit")
    finally:
        _ops.options.grad_arena = True
        flowmap_amd.set_lazy_surfaces(False)


def case_second_backward(dev):
    """What the reference's plain autograd graph allows, the fused operators allow too: a second backward through a
    retained graph (the fused losses re-launch from their saved inputs) and torch.autograd.grad w.r.t. a subset."""
    import flowmap_amd

    try:
        model, batch, flows, loss_of = _small_problem(dev)
        model.zero_grad(set_to_none=True)
        loss = loss_of(model(batch, flows, 0))
        loss.backward(retain_graph=True)
        first = _grads(model)
        model.zero_grad(set_to_none=True)
        loss.backward()  # the retained graph, again
        second = _grads(model)
        for x, y, what in zip(first, second, ("g_depth", "g_weights", "g_focal")):
            assert_close(y, x, 2e-5, abs_=1e-7, what=f"second backward {what}")
        # torch.autograd.grad for one parameter only
        loss = loss_of(model(batch, flows, 0))
        (g_focal,) = torch.autograd.grad(loss, [model.intrinsics.focal_length], retain_graph=True)
        assert_close(g_focal, first[2], 2e-5, abs_=1e-7, what="autograd.grad focal")
        (g_depth,) = torch.autograd.grad(loss, [model.backbone.depth])
        assert_close(g_depth, first[0], 2e-5, abs_=1e-7, what="autograd.grad depth")
        # an optimiser step between forward and backward is an error, as it is for any autograd graph
        loss = loss_of(model(batch, flows, 0))
        with torch.no_grad():
            model.backbone.depth.add_(1e-3)
        try:
            loss.backward()
        except RuntimeError as exc:
            assert "modified by an inplace operation" in str(exc)
        else:
            raise AssertionError("a stale graph was accepted")
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def case_threads_and_hooks(dev):
    """Autograd runs backward on its own worker thread, DDP adds gradient hooks: two independent models stepped
    concurrently from two Python threads (shared process state would cross their gradients), with
    post-accumulate-grad hooks on every parameter (as DistributedDataParallel registers them) that must fire
    exactly once per step and see the final gradient."""
    import threading

    import flowmap_amd

    try:
        problems = [_small_problem(dev, seed=31), _small_problem(dev, seed=32)]
        expected = []
        for model, batch, flows, loss_of in problems:  # serial reference
            model.zero_grad(set_to_none=True)
            loss_of(model(batch, flows, 0)).backward()
            expected.append(_grads(model))
        seen = [dict() for _ in problems]
        for which, (model, *_rest) in enumerate(problems):
            for name, p in model.named_parameters():
                p.register_post_accumulate_grad_hook(lambda param, which=which, name=name: seen[which].setdefault(name, []).append(param.grad.detach().clone()))
        errors = []

        def work(which):
            try:
                model, batch, flows, loss_of = problems[which]
                for _ in range(6):
                    model.zero_grad(set_to_none=True)
                    loss_of(model(batch, flows, 0)).backward()
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(problems))]
        for t_ in threads:
            t_.start()
        for t_ in threads:
            t_.join()
        assert not errors, errors
        for which, (model, *_rest) in enumerate(problems):
            got = _grads(model)
            for x, y, what in zip(expected[which], got, ("g_depth", "g_weights", "g_focal")):
                assert_close(y, x, 2e-5, abs_=1e-7, what=f"thread {which} {what}")
            assert all(len(v) == 6 for v in seen[which].values()) and len(seen[which]) == 3
            last = {name: v[-1] for name, v in seen[which].items()}
            assert_close(last["backbone.depth"], expected[which][0], 2e-5, abs_=1e-7, what="hook saw the final depth gradient")
            assert_close(last["backbone.weights"], expected[which][1], 2e-5, abs_=1e-7, what="hook saw the final weight gradient")
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def case_in_pass_adam(dev, steps=40, track_after=6, lr=1e-3, softmin=False, exchange=False):
    """FusedAdam.fuse_depth_update: the depth update applied inside the fused flow pass (fm_flow_loss_fused_adam) + the
    element-list update of the touched pixels (fm_adam_step_elements) walk the same trajectory as torch.optim.Adam on the
    same losses (model_wrapper_overfit.py:104-105) — flow loss from step 0, tracking loss switched on later
    (config/loss/tracking.yaml: enable_after), which must not surprise the fused update."""
    import flowmap_amd
    from flowmap_amd import Batch, FusedAdam, _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from helpers import to_tracks

    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg, RegressionCfg

    f, h, w = 5, 24, 32
    trajectories, engaged = {}, 0
    # ``exchange``: with the tap exchange forced on at this size — once the tracking loss runs, the flow pass absorbs its gradient, updates its
    # taps in the pass like any other pixel, and the next evaluation samples the image the update left (around the Procrustes pixels)
    min_bytes, counted = _ops.options.tap_exchange_min_bytes, dict(_ops.counters)
    _ops.options.tap_exchange_min_bytes = 0 if exchange else 1 << 60
    # softmin: the reference's default intrinsics (config/model/intrinsics/softmin.yaml) — the sweep reads random pixels of
    # frames 0 / 1, new every step, then hands over to a regressed focal length; both phases and the hand-over are crossed
    intrinsics = IntrinsicsSoftminCfg("softmin", 200, 0.5, 2.0, 8, RegressionCfg(steps // 2, 5)) if softmin else None
    try:
        for mode in ("torch", "fused", "in_pass", "fused_dense"):
            torch.manual_seed(7)  # the sweep draws its pixels from torch's CPU generator
            _ops.options.grad_arena = mode != "fused_dense"  # fused_dense: fresh zeros for dL/dweights, dense update of the logits
            model, batch, flows, _ = _small_problem(dev, f=f, h=h, w=w, tracking=False, intrinsics=intrinsics)
            sc = orc.synth_scene(f, h, w, seed=21)
            tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=21, interval=2, radius=2, grid=5), dev)
            flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
            track_fn = LossTracking(LossTrackingCfg(track_after, 100.0, "tracking", mapping_cfg("huber")))
            optimizer = torch.optim.Adam(model.parameters(), lr=lr) if mode == "torch" else FusedAdam(model.parameters(), lr=lr)
            if mode == "in_pass":
                optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
            history = []
            for step in range(steps):
                optimizer.zero_grad(set_to_none=True)
                before = model.backbone.depth.detach().clone()
                out = model(batch, flows, step)
                total = flow_fn(batch, flows, tracks, out, step)
                moved = mode == "in_pass" and not torch.equal(before, model.backbone.depth.detach())
                engaged += int(moved)
                total = total + track_fn(batch, flows, tracks, out, step)
                total.backward()
                optimizer.step()
                focal = next(p for name, p in model.named_parameters() if name.endswith("focal_length"))
                history.append([p.detach().clone() for p in (model.backbone.depth, model.backbone.weights, focal)]
                               + [total.detach().clone()])
            trajectories[mode] = (history, optimizer, model)
    finally:
        flowmap_amd.set_lazy_surfaces(False)
        _ops.options.grad_arena = True
        _ops.options.tap_exchange_min_bytes = min_bytes
    if exchange:
        tracked_steps = steps - track_after
        # per mode: the first tracked step runs in the plain order, every later one ahead of the flow pass; with the in-pass update the image
        # is sampled from the step after the first absorbing one on, with the separate updates never (the parameter moves after the pass)
        assert _ops.counters["flow_tap_absorbs"] - counted["flow_tap_absorbs"] == 4 * (tracked_steps - 1), _ops.counters
        assert _ops.counters["track_tap_samples"] - counted["track_tap_samples"] >= tracked_steps - 3, _ops.counters
    assert engaged >= steps - 3, engaged  # the plan exists from the third step on
    assert trajectories["fused_dense"][1].counters["sparse_updates"] == 0
    for mode in ("fused", "in_pass"):  # the weight logits: element-list update from the first planned step on
        # (under the softmin sweep too: the first weight image, where the sweep adds gradient at random pixels, is listed whole)
        assert trajectories[mode][1].counters["sparse_updates"] >= steps - 3, trajectories[mode][1].counters
    assert trajectories["in_pass"][1].counters["in_pass_updates"] == engaged
    # (1) the in-pass update IS the separate FusedAdam update: same arithmetic per element, only the pass it runs in differs
    # and the element-list update of the weight logits IS the dense one (fused vs fused_dense).
    # (asserted on the serial host double; on the GPU the unplanned scatters of the first steps and of the softmin sweep are
    # float atomics whose order differs from run to run, and the comparison below carries the weight)
    for other in ("in_pass", "fused_dense"):
        for step, (a, b) in enumerate(zip(trajectories[other][0], trajectories["fused"][0])):
            for x, y, what in zip(a, b, ("depth", "weights", "focal", "loss")):
                err = float((x - y).abs().max())
                assert str(dev) != "cpu" or err <= 2e-7 * max(1.0, float(y.abs().max())), (other, "vs fused", step, what, err)
    # (2) FusedAdam follows torch.optim.Adam: 2e-6 of the largest parameter at every step (lr here is 33x the reference's
    # 3e-5 so that the parameters visibly move; one-ulp differences between the two implementations' roundings are amplified
    # that much more than in a real run)
    ref = trajectories["torch"][0]
    worst = {}
    for mode in ("fused", "in_pass"):
        for step, (a, b) in enumerate(zip(trajectories[mode][0], ref)):
            for x, y, what in zip(a, b, ("depth", "weights", "focal", "loss")):
                err = float((x - y).abs().max())
                bound = 2e-6 * max(1.0, float(y.abs().max())) if what != "loss" else 2e-5 * max(1.0, float(y.abs().max()))
                if softmin and what == "weights":
                    # Adam moves an element by ~lr whatever the size of its gradient.  The sweep's gradient of a weight logit
                    # (a few hundred random pixels of the FIRST weight image per step) passes through dL/dK, a sum that cancels
                    # to rounding level: one ulp of difference in depth upstream changes it by tens of per cent, so those
                    # elements follow no reproducible trajectory in any implementation.  The other images are held to the bar.
                    err = float((x[1:] - y[1:]).abs().max())
                if softmin:
                    # the same noise reaches K, hence every pose and every gradient, on top of their well-conditioned parts;
                    # on the GPU the sweep's scatter adds with float atomics in a different order every run, which this
                    # amplification turns into scatter between ANY two runs (seen up to 1e-4 at 100x the reference's learning rate).  A missed or doubled update of an
                    # element would be an error of lr = 1e-3 after one step — two orders above either bar.
                    bound *= 2 if str(dev) == "cpu" else 4
                worst[(mode, what)] = max(worst.get((mode, what), 0.0), err / bound)
                assert err <= bound, (mode, step, what, err, bound)
    print("in-pass Adam: worst error / bound", {k: round(v, 3) for k, v in worst.items()})
    # the same optimiser state as the separate update, for every element (touched or not)
    sa, sb = (trajectories[m][1].state[trajectories[m][2].backbone.depth] for m in ("fused", "in_pass"))
    assert float(sa["step"]) == float(sb["step"]) == steps
    for name in ("exp_avg", "exp_avg_sq"):
        scale = float(sa[name].abs().max())
        assert float((sa[name] - sb[name]).abs().max()) <= 1e-4 * scale, name
    # the update has moved depth well beyond the tolerance (the comparison above is not vacuous)
    assert float((ref[-1][0] - ref[0][0]).abs().max()) > 5e-3


def case_in_pass_adam_exchange_unequal_upstreams(dev):
    """the tap exchange + FusedAdam.fuse_depth_update with the two losses reaching backward() under DIFFERENT upstream
    gradients (`flow + 3·tracking`).  The absorbing flow pass has then applied the depth update at the taps with the tracking gradient at
    factor 1 — not correctable afterwards — so the fit's node raises the optimiser's scaled-loss flag on the device (fm_tap_grad_apply's
    mismatch flag) and step() reports it; the plain sum of the same two losses runs clean; and a SECOND tracking loss on the same depth
    (its own track set) keeps its pixels in the element list: the trajectory of torch.optim.Adam."""
    import flowmap_amd
    from flowmap_amd import FusedAdam, _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from helpers import to_tracks

    f, h, w = 5, 24, 32
    min_bytes = _ops.options.tap_exchange_min_bytes
    _ops.options.tap_exchange_min_bytes = 0
    try:
        sc = orc.synth_scene(f, h, w, seed=21)
        tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=21, interval=2, radius=2, grid=5), dev)
        other = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=5, interval=3, radius=1, grid=4), dev)

        def loop(factor, steps, mode, second=False):
            model, batch, flows, _ = _small_problem(dev, f=f, h=h, w=w, tracking=False)
            flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
            track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", mapping_cfg("huber")))
            second_fn = LossTracking(LossTrackingCfg(0, 50.0, "tracking", mapping_cfg("huber"))) if second else None
            optimizer = torch.optim.Adam(model.parameters(), lr=1e-3) if mode == "torch" else FusedAdam(model.parameters(), lr=1e-3)
            if mode == "in_pass":
                optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
                optimizer.verify_unit_upstream_every = 1
            absorbed = _ops.counters["flow_tap_absorbs"]
            for step in range(steps):
                optimizer.zero_grad(set_to_none=True)
                out = model(batch, flows, step)
                total = flow_fn(batch, flows, tracks, out, step)
                tracked = track_fn(batch, flows, tracks, out, step)
                total = total + (tracked if factor == 1.0 else factor * tracked)
                if second_fn is not None:
                    total = total + second_fn(batch, flows, other, out, step)
                total.backward()
                optimizer.step()
            return model.backbone.depth.detach().clone(), optimizer, _ops.counters["flow_tap_absorbs"] - absorbed

        # equal upstreams: clean, and the pass did absorb
        _, optimizer, absorbed = loop(1.0, 6, "in_pass")
        assert absorbed >= 3 and optimizer.counters["in_pass_updates"] >= 4
        # unequal upstreams: loud, not silently wrong
        with pytest.raises(RuntimeError, match="unscaled"):
            loop(3.0, 6, "in_pass")
        # two tracking losses with their own track sets on one depth: the exchange (one tap set, one absorbed gradient per parameter) switches
        # itself off for that parameter, both losses' pixels stay in the element list, the in-pass update still runs
        d_ref, _, _ = loop(1.0, 8, "torch", second=True)
        d_ours, optimizer, absorbed = loop(1.0, 8, "in_pass", second=True)
        assert absorbed == 0 and optimizer.counters["in_pass_updates"] >= 5
        assert float((d_ours - d_ref).abs().max()) <= 4e-6 * float(d_ref.abs().max()), float((d_ours - d_ref).abs().max())
    finally:
        flowmap_amd.set_lazy_surfaces(False)
        _ops.options.tap_exchange_min_bytes = min_bytes


def case_in_pass_adam_refusals(dev):
    """Where the in-pass update cannot run it does not: weight decay, a dense fit (every pixel is a correspondence), a
    consumer of depth that shows up after the update has been applied (loud), a second backward."""
    import flowmap_amd
    from flowmap_amd import FusedAdam, _ops

    try:
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        optimizer = FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.1)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        for step in range(4):
            optimizer.zero_grad(set_to_none=True)
            before = model.backbone.depth.detach().clone()
            out = model(batch, flows, step)
            total = loss_of(out)
            assert torch.equal(before, model.backbone.depth.detach())  # weight decay: the usual path
            total.backward()
            optimizer.step()

        # too large a touched set (here most of a 24 x 32 image; 4.4 % with the tracking loss at 720p): the separate update stays
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        optimizer = FusedAdam(model.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth)  # default max_touched_fraction = 0.02
        for step in range(4):
            optimizer.zero_grad(set_to_none=True)
            loss_of(model(batch, flows, step)).backward()
            optimizer.step()
        assert optimizer.counters["in_pass_updates"] == 0 and optimizer.counters["sparse_updates"] == 3

        # a frame shard (FrameShard.prepare_model marks the frames shared with a neighbour): those frames' gradient is complete only
        # after the exchange, so the pass leaves them whole to step(), which updates them densely; every interior frame is
        # updated in the pass.  Same trajectory as torch.optim.Adam.
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        twin, _, _, twin_loss = _small_problem(dev, tracking=False)
        optimizer, reference = FusedAdam(model.parameters(), lr=1e-3), torch.optim.Adam(twin.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        frames = model.backbone.depth.shape[0]
        model.backbone.depth.__dict__["_fm_halo_frames"] = (0, frames - 1)
        for step in range(6):
            for m, o, fn in ((model, optimizer, loss_of), (twin, reference, twin_loss)):
                o.zero_grad(set_to_none=True)
                fn(m(batch, flows, step)).backward()
                o.step()
            assert_close(model.backbone.depth.detach(), twin.backbone.depth.detach(), 2e-6, abs_=2e-6, what=f"depth (halo frames dense) step {step}")
        assert optimizer.counters["in_pass_updates"] == 5  # (step 0: the fit's backward is not planned yet)
        state = optimizer.state[model.backbone.depth]
        assert float(state["step"]) == 6
        for name in ("exp_avg", "exp_avg_sq"):
            assert_close(state[name], reference.state[twin.backbone.depth][name], 1e-4, abs_=1e-12, what=f"{name} with halo frames")

        # a width the 16-byte path does not take (f=5, 24x30): the update must NOT engage, and nothing may be left pending
        model, batch, flows, loss_of = _small_problem(dev, h=24, w=30, tracking=False)
        twin, _, _, twin_loss = _small_problem(dev, h=24, w=30, tracking=False)
        optimizer, reference = FusedAdam(model.parameters(), lr=1e-3), torch.optim.Adam(twin.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        for step in range(5):
            for m, o, fn in ((model, optimizer, loss_of), (twin, reference, twin_loss)):
                o.zero_grad(set_to_none=True)
                fn(m(batch, flows, step)).backward()
                o.step()
            assert not optimizer.in_pass_pending(model.backbone.depth)
            assert_close(model.backbone.depth.detach(), twin.backbone.depth.detach(), 2e-6, abs_=2e-6, what=f"depth at width 30, step {step}")
        assert optimizer.counters["in_pass_updates"] == 0 and float(optimizer.state[model.backbone.depth]["step"]) == 5

        # a grad-enabled forward that never reaches backward() + step() (a loss recomputed for logging): the parameter has moved,
        # so the NEXT forward refuses loudly instead of silently dropping the dense part of that step; step() without
        # backward refuses too; a forward under no_grad is fine
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        optimizer = FusedAdam(model.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        for step in range(3):
            optimizer.zero_grad(set_to_none=True)
            loss_of(model(batch, flows, step)).backward()
            optimizer.step()
        with torch.no_grad():
            loss_of(model(batch, flows, 3))
        assert not optimizer.in_pass_pending(model.backbone.depth)
        optimizer.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 3))  # applied in the pass, then abandoned
        assert optimizer.in_pass_pending(model.backbone.depth)
        with pytest.raises(RuntimeError, match="never ran"):
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 3))
        with pytest.raises(RuntimeError, match="has not run since"):
            loss_of(model(batch, flows, 3))

        # a scaled loss: the in-pass update has used the unscaled gradient — reported at the next verification step
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        optimizer = FusedAdam(model.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        optimizer.verify_unit_upstream_every = 2
        with pytest.raises(RuntimeError, match="unscaled"):
            for step in range(6):
                optimizer.zero_grad(set_to_none=True)
                (0.5 * loss_of(model(batch, flows, step))).backward()
                optimizer.step()
        assert optimizer.counters["in_pass_updates"] >= 1

        # the element-list update of the weight logits: not when the gradient was edited after backward (clipping) —
        # the dense update runs, and the list is taken up again once the moments are verified zero elsewhere
        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        twin, _, _, twin_loss = _small_problem(dev, tracking=False)
        optimizer, reference = FusedAdam(model.parameters(), lr=1e-3), torch.optim.Adam(twin.parameters(), lr=1e-3)
        for step in range(7):
            for m, o, fn in ((model, optimizer, loss_of), (twin, reference, twin_loss)):
                o.zero_grad(set_to_none=True)
                fn(m(batch, flows, step)).backward()
                if step == 4:
                    m.backbone.weights.grad.mul_(0.5)
                o.step()
            assert optimizer.counters["sparse_updates"] == (0, 1, 2, 3, 3, 4, 5)[step], (step, optimizer.counters)
            assert_close(model.backbone.weights.detach(), twin.backbone.weights.detach(), 2e-6, abs_=2e-6, what=f"weights step {step}")
        optimizer.load_state_dict(optimizer.state_dict())  # loaded moments are verified before the list is used again
        optimizer.zero_grad(set_to_none=True)
        loss_of(model(batch, flows, 7)).backward()
        optimizer.step()
        assert optimizer.counters["sparse_updates"] == 6

        model, batch, flows, loss_of = _small_problem(dev, tracking=False)
        optimizer = FusedAdam(model.parameters(), lr=1e-3)
        optimizer.fuse_depth_update(model.backbone.depth, max_touched_fraction=1.0)
        with pytest.raises(ValueError):
            optimizer.fuse_depth_update(torch.zeros(3, device=dev, requires_grad=True))
        for step in range(3):
            optimizer.zero_grad(set_to_none=True)
            loss_of(model(batch, flows, step)).backward()
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        out = model(batch, flows, 3)
        total = loss_of(out)  # applied in the pass
        assert optimizer.in_pass_pending(model.backbone.depth)
        with pytest.raises(RuntimeError, match="already updated"):
            _ops.note_touched(out.depths, "late consumer", torch.zeros((4,), dtype=torch.int64, device=dev))
        total.backward(retain_graph=True)
        with pytest.raises(RuntimeError, match="twice"):
            total.backward()
        optimizer.step()
        assert not optimizer.in_pass_pending(model.backbone.depth)
        optimizer.fuse_depth_update(model.backbone.depth, enabled=False)
        optimizer.zero_grad(set_to_none=True)
        before = model.backbone.depth.detach().clone()
        total = loss_of(model(batch, flows, 4))
        assert torch.equal(before, model.backbone.depth.detach())
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def case_views_are_copied_loudly(dev):
    """A non-contiguous input is accepted (the C ABI takes dense buffers: it is copied) with the result of the
    contiguous call, and a copy the size of a pass of the step warns instead of happening silently."""
    import warnings

    f, h, w = 3, 1024, 2731  # 2 pairs x 2.8 M pixels x 2 floats = 22 M elements per flow tensor: above the 16 M threshold
    sc = orc.synth_scene(f, 8, 8, seed=3)  # poses / intrinsics only
    k = sc["intrinsics_gt"].expand(1, f, 3, 3).contiguous().to(dev)
    ext = sc["extrinsics_gt"][None].contiguous().to(dev)
    g = torch.Generator().manual_seed(0)
    surfaces = (1.0 + torch.rand((1, f, h, w, 3), generator=g)).to(dev)
    wide = (1.0 + torch.rand((1, f, h, w, 4), generator=g)).to(dev)
    view = wide[..., :3]
    assert not view.is_contiguous()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        a = fm.compute_forward_flow(view, ext, k)
    assert any("non-contiguous view" in str(c.message) for c in caught), [str(c.message) for c in caught]
    b = fm.compute_forward_flow(view.contiguous(), ext, k)
    assert torch.equal(a, b)


def case_pretraining_mode(dev):
    """The reference's pretraining loop brings a NEW Flows object with batch size > 1 every step
    (model_wrapper_pretrain.py:46-71): nothing that is worth its cost only for inputs that come back may run — no re-layout of
    the flows (fm_flow_pack_inputs), no scatter plan — and every step must still match the oracle.  The same inputs offered a
    second time ARE packed and planned (the overfit loop), and the packed originals can then be released."""
    import flowmap_amd
    from flowmap_amd import Batch, Flows, ModelOutput, _ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg

    b, f, h, w, p = 2, 4, 16, 24, 60
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))
    idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64)
    idx_dev = idx.to(dev)

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        depth = 1.1 + 0.1 * torch.rand((b, f, h, w), generator=g)
        weights = torch.rand((b, f - 1, h, w), generator=g)
        fl = orc.OFlows(0.01 * torch.randn((b, f - 1, h, w, 2), generator=g), 0.01 * torch.randn((b, f - 1, h, w, 2), generator=g),
                        torch.rand((b, f - 1, h, w), generator=g), torch.rand((b, f - 1, h, w), generator=g))
        return depth, weights, fl

    k = orc.focal_to_k(torch.tensor([0.8, 0.9]), (h, w))[:, None].expand(b, f, 3, 3).contiguous()

    def ours(depth, weights, flows):
        d, wt, kk = (x.detach().clone().to(dev).requires_grad_(True) for x in (depth, weights, k))
        xy, _ = fm.sample_image_grid((h, w), dev)
        surfaces = fm.unproject(xy, d, kk[:, :, None, None])
        ext = fm.align_surfaces(surfaces, flows.backward, wt, idx_dev)
        loss = loss_fn(Batch(torch.zeros((b, f, 3, h, w), device=dev)), flows, None, ModelOutput(d, surfaces, kk, ext, wt), 0)
        loss.backward()
        return loss.detach(), d.grad, wt.grad

    def truth(depth, weights, fl):
        d64, w64, k64 = depth.double().requires_grad_(True), weights.double().requires_grad_(True), k.double()
        fl64 = orc.OFlows(*(x.double() for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask)))
        o = orc.model_forward(d64, w64, k64, fl64, idx)
        ref = 1000.0 * orc.flow_loss(o.surfaces, o.extrinsics, k64, fl64, (h, w))
        ref.backward()
        return ref.detach(), d64.grad, w64.grad

    fm.set_lazy_surfaces(True)
    try:
        packs, plans, planned = _ops.counters["flow_packs"], _ops.counters["procrustes_plans_built"], _ops.counters["procrustes_planned"]
        for step in range(4):  # a fresh batch every step
            depth, weights, fl = inputs(100 + step)
            flows = Flows(fl.forward.to(dev), fl.backward.to(dev), fl.forward_mask.to(dev), fl.backward_mask.to(dev))
            got, want = ours(depth, weights, flows), truth(depth, weights, fl)
            for a_, b_, what in zip(got, want, ("loss", "g_depth", "g_weights")):
                assert_close(a_, b_, 3 * TOL if what == "g_weights" else TOL, what=f"{what}, fresh batch {step}")
        assert _ops.counters["flow_packs"] == packs, "a Flows object seen once was re-laid-out"
        assert _ops.counters["procrustes_plans_built"] == plans and _ops.counters["procrustes_planned"] == planned, "a scatter plan was built for inputs seen once"

        # the same Flows object again and again (overfitting): packed and planned at the second step, then released
        depth, weights, fl = inputs(7)
        flows = Flows(fl.forward.to(dev), fl.backward.to(dev), fl.forward_mask.to(dev), fl.backward_mask.to(dev))
        want = truth(depth, weights, fl)
        history = []
        for step in range(5):
            history.append(ours(depth, weights, flows))
            assert _ops.counters["flow_packs"] - packs == (0 if step == 0 else 1), step
            if step == 2:
                freed = flowmap_amd.release_flow_originals(flows)
                assert freed == (flows.forward.numel() + 2 * flows.forward_mask.numel()) * 4
                assert flows.forward.untyped_storage().nbytes() <= 16 and flows.forward.shape == (b, f - 1, h, w, 2)
        assert _ops.counters["procrustes_plans_built"] == plans + 1
        for step, got in enumerate(history):
            for a_, b_, what in zip(got, want, ("loss", "g_depth", "g_weights")):
                assert_close(a_, b_, 3 * TOL if what == "g_weights" else TOL, what=f"{what}, constant flows step {step}")
        for a_, b_, what in zip(history[-1], history[1], ("loss", "g_depth", "g_weights")):
            assert_close(a_, b_, 1e-6, abs_=1e-9, what=f"{what} after the originals were released")
        with pytest.raises(RuntimeError, match="not been packed"):
            flowmap_amd.release_flow_originals(Flows(*(x.clone().to(dev) for x in (fl.forward, fl.backward, fl.forward_mask, fl.backward_mask))))
    finally:
        fm.set_lazy_surfaces(False)


def case_frame_windows(dev):
    """Frame windows of larger tensors — x[:, s:s+f] (loss_tracking.py:44-52), earlier(x) / later(x) (projection.py:139-140), a
    slice of a pretraining batch — are read IN PLACE through the C ABI's fm_layout strides (fm_*_views): same loss and
    gradients as contiguous clones of the windows, over several steps (so the packed copy, the scatter plan and the static
    tap records are built from views too), and not one copy made."""
    import warnings

    from flowmap_amd import Batch, Flows, ModelOutput, _ops
    from flowmap_amd._lib import torch_ops
    from flowmap_amd.loss import LossFlow, LossFlowCfg

    b, frames_full, h, w, p = 2, 8, 16, 24, 60
    s0, f = 2, 5  # the window: frames 2..6 of 8
    g = torch.Generator().manual_seed(17)
    depth_full = (1.1 + 0.1 * torch.rand((b, frames_full, h, w), generator=g)).to(dev)
    weights_full = torch.rand((b, frames_full - 1, h, w), generator=g).to(dev)
    flows_full = [t.to(dev) for t in (0.01 * torch.randn((b, frames_full - 1, h, w, 2), generator=g), 0.01 * torch.randn((b, frames_full - 1, h, w, 2), generator=g),
                                      torch.rand((b, frames_full - 1, h, w), generator=g), torch.rand((b, frames_full - 1, h, w), generator=g))]
    k = orc.focal_to_k(torch.tensor([0.8, 0.9]), (h, w))[:, None].expand(b, f, 3, 3).contiguous().to(dev)
    idx = torch.linspace(0, h * w - 1, p, dtype=torch.int64).to(dev)
    loss_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", mapping_cfg("huber")))

    def run(window: bool, steps=3):
        d_leaf = depth_full.clone().requires_grad_(True)
        w_leaf = weights_full.clone().requires_grad_(True)
        kk = k.clone().requires_grad_(True)
        if window:
            flows = Flows(*(t[:, s0 : s0 + f - 1] for t in flows_full))
        else:
            flows = Flows(*(t[:, s0 : s0 + f - 1].contiguous() for t in flows_full))
        out = []
        for _ in range(steps):
            for leaf in (d_leaf, w_leaf, kk):
                leaf.grad = None
            d = d_leaf[:, s0 : s0 + f] if window else d_leaf[:, s0 : s0 + f].contiguous()
            wt = w_leaf[:, s0 : s0 + f - 1] if window else w_leaf[:, s0 : s0 + f - 1].contiguous()
            assert d.is_contiguous() != window
            xy, _ = fm.sample_image_grid((h, w), dev)
            surfaces = fm.unproject(xy, d, kk[:, :, None, None])
            ext = fm.align_surfaces(surfaces, flows.backward, wt, idx)
            loss = loss_fn(Batch(torch.zeros((b, f, 3, h, w), device=dev)), flows, None, ModelOutput(d, surfaces, kk, ext, wt), 0)
            loss.backward()
            out.append((loss.detach().clone(), ext.detach().clone(), d_leaf.grad.clone(), w_leaf.grad.clone(), kk.grad.clone()))
        return out

    fm.set_lazy_surfaces(True)
    try:
        dense = run(False)
        copies, packs, plans = torch_ops().view_copies(), _ops.counters["flow_packs"], _ops.counters["procrustes_plans_built"]
        with warnings.catch_warnings():
            warnings.simplefilter("error")  # "... is a non-contiguous view ... and is copied on every call" must not appear
            views = run(True)
        assert torch_ops().view_copies() == copies, "a frame window was copied"
        assert _ops.counters["flow_packs"] == packs + 1 and _ops.counters["procrustes_plans_built"] == plans + 1  # packed / planned FROM the views
        for step, (a, c) in enumerate(zip(views, dense)):
            for x, y, what in zip(a, c, ("loss", "extrinsics", "g_depth (scattered back into the full tensor)", "g_weights", "g_k")):
                assert_close(x, y, 2e-5, abs_=1e-7, what=f"{what}, step {step}")
        # outside the window nothing may have been written
        g_depth = views[-1][2]
        assert float(g_depth[:, :s0].abs().max()) == 0 and float(g_depth[:, s0 + f :].abs().max()) == 0
        # a view that is NOT a frame window (pixels transposed) is still accepted — copied, and counted
        odd = depth_full[:, s0 : s0 + f].transpose(2, 3).contiguous().transpose(2, 3)
        xy, _ = fm.sample_image_grid((h, w), dev)
        fm.align_surfaces(fm.unproject(xy, odd, k[:, :, None, None]), flows_full[1][:, s0 : s0 + f - 1], weights_full[:, s0 : s0 + f - 1], idx)
        assert torch_ops().view_copies() == copies + 1
        # batch-EXPANDED stacks (stride 0 over the batch: one video's flows / masks shown to every batch entry) overlap between batch
        # entries, which the launchers refuse: the binding copies them instead of passing them on as "views" 
        assert _ops.frame_window_layout(flows_full[2][:1].expand(b, -1, -1, -1)) is None
        one = Flows(*(t[:1, s0 : s0 + f - 1].expand(b, *t.shape[1:][:0], f - 1, *t.shape[2:]) for t in flows_full))
        same = Flows(*(t.contiguous() for t in (one.forward, one.backward, one.forward_mask, one.backward_mask)))
        results = []
        for fl in (one, same):
            d = depth_full[:, s0 : s0 + f].clone().requires_grad_(True)
            surfaces = fm.unproject(xy, d, k[:, :, None, None])
            wt = weights_full[:, s0 : s0 + f - 1].contiguous()
            ext = fm.align_surfaces(surfaces, fl.backward, wt, idx)
            loss = loss_fn(Batch(torch.zeros((b, f, 3, h, w), device=dev)), fl, None, ModelOutput(d, surfaces, k, ext, wt), 0)
            loss.backward()
            results.append((loss.detach(), d.grad))
        assert_close(results[0][0], results[1][0], 1e-6, what="loss, batch-expanded flows")
        assert_close(results[0][1], results[1][1], 1e-6, abs_=1e-9, what="g_depth, batch-expanded flows")
    finally:
        fm.set_lazy_surfaces(False)


def case_halo_kernels(dev):
    """fm_halo_copy / _delta / _add / _scatter / _ghost_begin / _delta_sparse (csrc/fm_shard.hip: the local work of the frame shards' halo exchange) against the
    torch indexing operators they replace, with both, one and no boundary active, and a frame size the 16-byte path does not take."""
    from flowmap_amd._lib import call, ptr, stream_for

    for frames, h, w in ((5, 12, 16), (2, 9, 7)):
        n = h * w
        g = torch.Generator().manual_seed(frames)
        grad = torch.randn((frames, h, w), generator=g).to(dev)
        px = [torch.randperm(n, generator=g)[: n // 3].to(dev), torch.randperm(n, generator=g)[: n // 4].to(dev)]
        for sides in ((True, True), (True, False), (False, True), (False, False)):
            first, last = sides
            sent = [torch.full((h, w), float("nan"), device=dev) if on else None for on in sides]
            call("fm_halo_copy", ptr(grad), n, frames, ptr(sent[0]), ptr(sent[1]), stream_for(grad))
            if first:
                assert torch.equal(sent[0], grad[0])
            if last:
                assert torch.equal(sent[1], grad[-1])
            moved = grad.clone()
            moved[0].view(-1)[px[0]] += 1.5
            moved[-1].view(-1)[px[1]] -= 0.25
            out = [torch.full((px[i].numel(),), float("nan"), device=dev) if sides[i] else None for i in range(2)]
            call("fm_halo_delta", ptr(moved), n, frames, ptr(sent[0]), ptr(px[0]), px[0].numel(), ptr(out[0]), ptr(sent[1]), ptr(px[1]), px[1].numel(),
                 ptr(out[1]), stream_for(grad))
            if first:
                assert_close(out[0], moved[0].view(-1)[px[0]] - grad[0].view(-1)[px[0]], 1e-6, abs_=1e-7, what="delta, first frame")
            if last:
                assert_close(out[1], moved[-1].view(-1)[px[1]] - grad[-1].view(-1)[px[1]], 1e-6, abs_=1e-7, what="delta, last frame")
            # the ghost halo's forms of the same two steps: a compact baseline (the touched pixels only) gathered by the launch that also packs
            # the boundary pairs' poses, K and K^-1 (fm_halo_ghost_begin), and the delta against it (fm_halo_delta_sparse)
            pairs = 3
            t_fwd, t_bwd = torch.randn((pairs, 4, 4), generator=g).to(dev), torch.randn((pairs, 4, 4), generator=g).to(dev)
            k3, kinv3 = torch.randn((3, 3), generator=g).to(dev), torch.randn((3, 3), generator=g).to(dev)
            for with_pack in (True, False):
                base = [torch.full((px[i].numel(),), float("nan"), device=dev) if sides[i] else None for i in range(2)]
                pack = torch.full((82,), float("nan"), device=dev) if with_pack else None
                call("fm_halo_ghost_begin", ptr(grad), n, frames, ptr(px[0]), px[0].numel(), ptr(base[0]), ptr(px[1]), px[1].numel(), ptr(base[1]),
                     ptr(t_fwd), ptr(t_bwd), pairs, ptr(k3), ptr(kinv3), ptr(pack), stream_for(grad))
                if with_pack:
                    want_pack = torch.cat([t_fwd[0].reshape(-1), t_bwd[-1].reshape(-1), t_bwd[0].reshape(-1), t_fwd[-1].reshape(-1), k3.reshape(-1), kinv3.reshape(-1)])
                    assert torch.equal(pack, want_pack)
                for i, frame in ((0, 0), (1, -1)):
                    if sides[i]:
                        assert torch.equal(base[i], grad[frame].reshape(-1)[px[i]])
                out2 = [torch.full((px[i].numel(),), float("nan"), device=dev) if sides[i] else None for i in range(2)]
                call("fm_halo_delta_sparse", ptr(moved), n, frames, ptr(base[0]), ptr(px[0]), px[0].numel(), ptr(out2[0]), ptr(base[1]), ptr(px[1]), px[1].numel(),
                     ptr(out2[1]), stream_for(grad))
                for i in range(2):
                    if sides[i]:
                        assert torch.equal(out2[i], out[i]), "the delta against the compact baseline = the delta against the copied frame"
            dense = [torch.randn((h, w), generator=g).to(dev) if on else None for on in sides]
            vals = [torch.randn((px[i].numel(),), generator=g).to(dev) if sides[i] else None for i in range(2)]
            want = grad.clone()
            if first:
                want[0] += dense[0]
                want[0].view(-1).index_add_(0, px[0], vals[0])
            if last:
                want[-1] += dense[1]
                want[-1].view(-1).index_add_(0, px[1], vals[1])
            got = grad.clone()
            call("fm_halo_add", ptr(got), n, frames, ptr(dense[0]), ptr(dense[1]), stream_for(grad))
            call("fm_halo_scatter", ptr(got), n, frames, ptr(px[0]), ptr(vals[0]), px[0].numel(), ptr(px[1]), ptr(vals[1]), px[1].numel(), stream_for(grad))
            assert_close(got, want, 1e-6, abs_=1e-6, what=f"add + scatter, sides {sides}")
            assert torch.equal(got[1:-1], grad[1:-1])  # interior frames untouched
