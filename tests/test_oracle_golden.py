"""Pin oracle/flowmap_oracle.py to the golden vectors produced by the reference itself
(oracle/make_golden.py).  CPU only."""

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_close_or_reference_gap, load_golden, t
from oracle import flowmap_oracle as orc

TOL = 2e-5  # oracle and reference are both fp32 torch; they differ only by op order


def _flows(g, dtype=torch.float32):
    return orc.OFlows(t(g["fwd"], dtype), t(g["bwd"], dtype), t(g["fwd_mask"], dtype), t(g["bwd_mask"], dtype))


def _tracks(g, dtype=torch.float32):
    if "n_segments" not in g:
        return None
    return [
        orc.OTracks(t(g[f"trk{i}_xy"], dtype), t(g[f"trk{i}_vis"]), int(g[f"trk{i}_start"]))
        for i in range(int(g["n_segments"]))
    ]


def run_oracle_step(g, kind="huber", dtype=torch.float32):
    depth = t(g["depth"], dtype).requires_grad_(True)
    wlogit = t(g["wlogit"], dtype).requires_grad_(True)
    focal = torch.tensor(float(g["focal"]), dtype=dtype, requires_grad=True)
    hw = depth.shape[1:]
    npts = int(g["num_points"])
    total, parts, out = orc.explicit_depth_step(
        depth, wlogit, focal, _flows(g, dtype), hw, num_points=None if npts < 0 else npts, tracks=_tracks(g, dtype), kind=kind
    )
    total.backward()
    return total, parts, out, depth.grad, wlogit.grad, focal.grad


@pytest.mark.parametrize(
    "name,kind",
    [("step_iid_flow", "huber"), ("step_scene_flow_tracking", "huber"), ("step_iid_l1_odd", "l1"), ("step_iid_l2_odd", "l2")],
)
def test_step_matches_reference(name, kind):
    g = load_golden(name)
    total, parts, out, gd, gw, gf = run_oracle_step(g, kind)
    assert_close(total, g["total"], TOL, what="total")
    assert_close(parts["flow"], g["loss_flow"], TOL, what="flow")
    if "tracking" in parts:
        assert_close(parts["tracking"], g["loss_tracking"], TOL, what="tracking")
    assert_close(out.extrinsics, g["extrinsics"], TOL, what="extrinsics")
    assert_close(out.intrinsics, g["intrinsics"], TOL, what="intrinsics")
    assert_close(gd, g["g_depth"], 1e-4, what="g_depth")
    assert_close(gw, g["g_wlogit"], 1e-4, what="g_wlogit")
    assert_close(gf, g["g_focal"], 1e-3, abs_=1e-4 * abs(float(g["total"])), what="g_focal")


@pytest.mark.parametrize("name", ["step_iid_flow", "step_scene_flow_tracking"])
def test_step_fp64_matches_reference_fp64(name):
    g = load_golden(name)
    total, parts, out, gd, gw, gf = run_oracle_step(g, "huber", torch.float64)
    assert_close(total, g["f64_total"], 1e-10, what="total")
    assert_close(gd, g["f64_g_depth"], 1e-8, what="g_depth")
    assert_close(gw, g["f64_g_wlogit"], 1e-8, what="g_wlogit")
    assert_close(gf, g["f64_g_focal"], 1e-8, what="g_focal")


def test_grid_and_unproject():
    g = load_golden("fn_unproject")
    h, w = g["xy"].shape[:2]
    xy, ij = orc.pixel_grid((h, w))
    assert np.array_equal(xy.numpy(), g["xy"])
    assert np.array_equal(ij.numpy(), g["ij"])
    z = t(g["z"]).requires_grad_(True)
    k = t(g["k"]).requires_grad_(True)
    s = orc.lift(xy, z, k[:, :, None, None])
    (s * t(g["cot"])).sum().backward()
    assert_close(s, g["surfaces"], TOL)
    assert_close(z.grad, g["g_z"], TOL)
    assert_close(k.grad, g["g_k"], TOL)


def test_flow_positions():
    g = load_golden("fn_flow_positions")
    s = t(g["surfaces"]).requires_grad_(True)
    e = t(g["extrinsics"]).requires_grad_(True)
    k = t(g["intrinsics"]).requires_grad_(True)
    f = orc.forward_flow_positions(s, e, k)
    b = orc.backward_flow_positions(s, e, k)
    ((f * t(g["cot_f"])).sum() + (b * t(g["cot_b"])).sum()).backward()
    assert_close(f, g["xy_fwd"], TOL)
    assert_close(b, g["xy_bwd"], TOL)
    assert_close(s.grad, g["g_surfaces"], TOL)
    assert_close(e.grad, g["g_extrinsics"], 5e-5)
    assert_close(k.grad, g["g_intrinsics"], TOL)
    g1 = load_golden("fn_flow_positions_1d")
    b1 = orc.backward_flow_positions(t(g1["surfaces"]), t(g1["extrinsics"]), t(g1["intrinsics"]))
    assert_close(b1, g1["xy_bwd"], TOL)


def test_projection_edges():
    g = load_golden("fn_project_edge")
    out = orc.pinhole(t(g["points"]), t(g["k"]))
    ref = t(g["xy"])
    assert torch.isfinite(out).all()
    assert_close(out, ref, 1e-6)
    g = load_golden("fn_reproject")
    assert_close(orc.warp_points(t(g["xyz"]), t(g["rel"]), t(g["k"])), g["xy"], TOL)
    g = load_golden("fn_project")
    xy, front = orc.world_to_image(t(g["xyz"]), t(g["extrinsics"]), t(g["k"]))
    assert_close(xy, g["xy"], TOL)
    assert np.array_equal(front.numpy(), g["in_front"])


def test_pose_chain():
    g = load_golden("fn_get_extrinsics")
    rel = t(g["rel"]).requires_grad_(True)
    e = orc.chain_poses(rel)
    (e * t(g["cot"])).sum().backward()
    assert_close(e, g["extrinsics"], TOL)
    assert_close(rel.grad, g["g_rel"], TOL)


@pytest.mark.parametrize("case", ["generic", "noisy_planar", "few"])
def test_rigid_fit(case):
    g = load_golden("fn_align_rigid")
    p = t(g[f"{case}_p"]).requires_grad_(True)
    q = t(g[f"{case}_q"]).requires_grad_(True)
    w = t(g[f"{case}_w"]).requires_grad_(True)
    T = orc.rigid_fit(p, q, w)
    (T * t(g[f"{case}_cot"])).sum().backward()
    assert_close(T, g[f"{case}_T"], TOL)
    # the oracle's fp32 gradients against the reference's own fp64 ones, at 1e-4 or twice the gap of the reference's fp32 gradients
    for mine, name in ((p.grad, "p"), (q.grad, "q"), (w.grad, "w")):
        assert_close_or_reference_gap(mine, g[f"{case}_f64_g_{name}"], g[f"{case}_g_{name}"], TOL, what=f"g_{name}")
    # and the oracle in fp64 IS the reference in fp64
    p64, q64, w64 = (t(g[f"{case}_{n}"]).double().requires_grad_(True) for n in "pqw")
    (orc.rigid_fit(p64, q64, w64) * t(g[f"{case}_cot"]).double()).sum().backward()
    for mine, name in ((p64.grad, "p"), (q64.grad, "q"), (w64.grad, "w")):
        assert_close(mine, g[f"{case}_f64_g_{name}"], 1e-9, what=f"fp64 g_{name}")


def test_fit_poses():
    g = load_golden("fn_align_surfaces")
    z = t(g["z"]).requires_grad_(True)
    k = t(g["k"]).requires_grad_(True)
    w = t(g["weights"]).requires_grad_(True)
    h, wd = z.shape[2:]
    xy, _ = orc.pixel_grid((h, wd))
    e = orc.fit_poses(orc.lift(xy, z, k[:, :, None, None]), t(g["bwd_flow"]), w, t(g["indices"]))
    (e * t(g["cot"])).sum().backward()
    assert_close(e, g["extrinsics"], TOL)
    for mine, name in ((z.grad, "z"), (k.grad, "k"), (w.grad, "weights")):  # (against the reference in fp64, as above)
        assert_close_or_reference_gap(mine, g[f"f64_g_{name}"], g[f"g_{name}"], TOL, what=f"g_{name}")


def test_track_positions():
    g = load_golden("fn_track_flow")
    z = t(g["z"]).requires_grad_(True)
    k = t(g["k"]).requires_grad_(True)
    e = t(g["extrinsics"]).requires_grad_(True)
    h, w = z.shape[2:]
    xy, _ = orc.pixel_grid((h, w))
    tgt, vis = orc.track_positions(orc.lift(xy, z, k[:, :, None, None]), e, k, orc.OTracks(t(g["track_xy"]), t(g["track_vis"]), 0))
    (tgt * t(g["cot"])).sum().backward()
    assert_close(tgt, g["xy_target"], TOL)
    assert np.array_equal(vis.numpy(), g["visibility"])
    assert_close(z.grad, g["g_z"], TOL)
    assert_close(k.grad, g["g_k"], TOL)
    assert_close(e.grad, g["g_extrinsics"], 5e-5)


@pytest.mark.parametrize("kind", ["huber", "l1", "l2"])
def test_mappings(kind):
    g = load_golden("fn_mapping")
    a = t(g["a"]).requires_grad_(True)
    val = orc.robust(a, t(g["b"]), tuple(int(x) for x in g["image_shape"]), kind)
    val.sum().backward()
    assert_close(val, g[f"{kind}_val"], 1e-6)
    assert_close(a.grad, g[f"{kind}_g_a"], 1e-6)
    assert torch.isfinite(a.grad).all()


def test_softmin_intrinsics():
    g = load_golden("fn_softmin")
    d = t(g["depth"])[None].requires_grad_(True)
    w = t(g["weights"]).requires_grad_(True)
    h, wd = d.shape[2:]
    k = orc.softmin_intrinsics(d, w, t(g["bwd"]), t(g["candidates"]), t(g["indices"]), (h, wd))
    (k[0] * t(g["cot"])).sum().backward()
    assert_close(k[0], g["intrinsics"], TOL, what="intrinsics")
    assert_close_or_reference_gap(d.grad[0], g["f64_g_depth"][0], g["g_depth"][0], TOL, what="g_depth")
    assert_close_or_reference_gap(w.grad, g["f64_g_weights"], g["g_weights"], TOL, what="g_weights")
    # the oracle in fp64 IS the reference module in fp64 (the truth tests/test_install_reference.py measures the softmin sweep against)
    d64 = t(g["depth"])[None].double().requires_grad_(True)
    w64 = t(g["weights"]).double().requires_grad_(True)
    k64 = orc.softmin_intrinsics(d64, w64, t(g["bwd"]).double(), t(g["candidates"]).double(), t(g["indices"]), (h, wd))
    (k64[0] * t(g["cot"]).double()).sum().backward()
    assert_close(k64[0], g["f64_intrinsics"], 1e-9, what="fp64 intrinsics")
    assert_close(d64.grad[0], g["f64_g_depth"][0], 1e-7, what="fp64 g_depth")
    assert_close(w64.grad, g["f64_g_weights"], 1e-7, what="fp64 g_weights")


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_flow_preprocess(tag):
    g = load_golden("fn_flow_preprocess")
    videos, raw, shape = t(g[f"{tag}_videos"]), t(g[f"{tag}_raw"]), tuple(int(x) for x in g[f"{tag}_shape"])
    assert_close(orc.consistency_mask(videos, raw), g[f"{tag}_mask_full"], 1e-6, what="mask_full")
    flows = orc.bidirectional_flows(videos, lambda v: raw if torch.equal(v, videos) else orc.standin_predictor(v), shape)
    for name, val in (("forward", flows.forward), ("backward", flows.backward), ("forward_mask", flows.forward_mask),
                      ("backward_mask", flows.backward_mask)):
        assert_close(val, g[f"{tag}_{name}"], 1e-6, what=name)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_cropping(tag):
    from cases import CROPPING_CASES

    g = load_golden("fn_cropping")
    image_shape, mult, patch = CROPPING_CASES[tag]
    videos, k = t(g[f"{tag}_videos"]), t(g[f"{tag}_intrinsics"])
    for name, m in (("model", 1), ("flow", mult)):
        out, k_out, resized = orc.crop_and_resize(videos, k, image_shape, patch, m)
        assert_close(out, g[f"{tag}_{name}_videos"], 1e-6, what=f"{name}_videos")
        assert_close(k_out, g[f"{tag}_{name}_intrinsics"], 1e-6, what=f"{name}_intrinsics")
        if m == 1:
            assert tuple(resized) == tuple(int(x) for x in g[f"{tag}_pre_crop"])


def test_export_point_cloud_and_ate():
    g = load_golden("fn_export")
    pts, cols = orc.world_point_cloud(t(g["depths"]), t(g["intrinsics"]), t(g["extrinsics"]), t(g["colors"]))
    assert_close(pts, g["points"], 1e-6, what="points")
    assert torch.equal(cols, t(g["point_colors"]))
    assert abs(orc.ate(t(g["ate_gt"]), t(g["ate_pred"])) - float(g["ate"])) < 1e-6
