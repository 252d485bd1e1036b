"""GPU parity: the HIP path (through the C ABI, on a real MI355X) against the oracle and
the reference's golden vectors.  Run with `pytest -m gpu` through gpurun."""

import pytest
import torch

from conftest import assert_close, assert_close_or_reference_gap, load_golden, t
from helpers import compare_step, run_oracle, run_ours, step_masks
from oracle import flowmap_oracle as orc
from test_oracle_golden import _flows, _tracks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def compare(ours, truth, ref32=None, masks=None):
    compare_step(ours, truth, ref32, masks=masks)


def test_native_library_is_the_one_loaded():
    from flowmap_amd import _lib

    assert not _lib.using_test_double()
    assert _lib.library() is not None
    _lib.torch_ops()
    maps = open("/proc/self/maps").read()
    assert "libflowmap_hip.so" in maps and "libflowmap_torch.so" in maps


@pytest.mark.parametrize("lazy", [True, False])
@pytest.mark.parametrize(
    "name,kind",
    [("step_iid_flow", "huber"), ("step_scene_flow_tracking", "huber"), ("step_iid_l1_odd", "l1"), ("step_iid_l2_odd", "l2")],
)
def test_step_vs_reference_golden(name, kind, lazy):
    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    npts = int(g["num_points"])
    ours = run_ours(depth, wlogit, float(g["focal"]), _flows(g), depth.shape[1:], None if npts < 0 else npts, _tracks(g), kind,
                    device=DEV, lazy=lazy)
    golden = {k: t(g[k]) for k in ("total", "loss_flow", "loss_tracking", "extrinsics", "g_depth", "g_wlogit", "g_focal")}
    # the truth is the fp64 oracle on the golden inputs; the reference's fp32 golden values give its own gap
    truth = run_oracle(depth, wlogit, float(g["focal"]), _flows(g), depth.shape[1:], None if npts < 0 else npts, _tracks(g), kind,
                       dtype=torch.float64)
    for key in ("total", "loss_flow", "loss_tracking", "extrinsics"):  # values: straight against the reference's own numbers too
        assert_close(ours[key], golden[key], 1e-4, what=f"{key} vs golden")
    compare(ours, truth, golden, masks=step_masks(depth.shape[1:], None if npts < 0 else npts, _flows(g), _tracks(g)))


@pytest.mark.parametrize(
    "f,h,w,p",
    [
        (16, 256, 256, 1000),  # C0 (BASELINE.json configs[0])
        (4, 720, 1280, 1000),  # C1's frame size, few frames (oracle finishes in seconds)
        (3, 1080, 1920, 1000),  # C3/C4's frame size
        (5, 90, 122, None),  # width not a multiple of 4 -> scalar kernel path, dense Procrustes
        (3, 64, 96, 600),
        (2, 32, 48, 100),  # a single pair
        (2, 24, 40, None),
    ],
)
def test_step_vs_oracle(f, h, w, p):
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=f + h)
    ours = run_ours(depth, wlogit, 0.85, flows, (h, w), p, device=DEV)
    truth = run_oracle(depth, wlogit, 0.85, flows, (h, w), p, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), p, dtype=torch.float32)  # i.i.d. inputs: the reference's own gap is measured
    compare(ours, truth, ref32, masks=step_masks((h, w), p, flows))


def test_consistent_scene_has_small_loss_and_matches_oracle():
    f, h, w = 8, 96, 128
    sc = orc.synth_scene(f, h, w, seed=2, depth_noise=0.0)
    wl = torch.zeros((f - 1, h, w))
    ours = run_ours(sc["depth_gt"], wl, sc["focal"], sc["flows"], (h, w), 500, device=DEV)
    ref = run_oracle(sc["depth_gt"], wl, sc["focal"], sc["flows"], (h, w), 500, dtype=torch.float64)
    assert float(ours["total"]) < 1.0  # ground-truth depth + flows generated from it (i.i.d. inputs give ~8)
    # a near-zero loss (ground-truth depth): the fp32 reference path itself is measured against the fp64 truth on the same inputs
    ref32 = run_oracle(sc["depth_gt"], wl, sc["focal"], sc["flows"], (h, w), 500, dtype=torch.float32)
    assert_close_or_reference_gap(ours["total"], ref["total"], ref32["total"], 1e-4, what="total")


def test_loss_scale_and_carry_gpu():
    f, h, w = 4, 64, 64
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=3)
    a = run_ours(depth, wlogit, 0.85, flows, (h, w), 100, device=DEV)
    b = run_ours(depth, wlogit, 0.85, flows, (h, w), 100, device=DEV, loss_scale=0.5)
    assert_close(b["g_depth"], 0.5 * a["g_depth"], 1e-5)
    assert_close(b["g_wlogit"], 0.5 * a["g_wlogit"], 1e-5)
    from flowmap_amd.loss import LossFlow

    LossFlow.carry_depth_grad = False
    try:
        c = run_ours(depth, wlogit, 0.85, flows, (h, w), 100, device=DEV)
    finally:
        LossFlow.carry_depth_grad = True
    assert_close(c["g_depth"], a["g_depth"], 1e-5)


@pytest.mark.parametrize("f,h,w,packed", [(150, 720, 1280, False), (150, 720, 1280, True), (150, 1080, 1920, True)],
                         ids=["c1-reference-layout", "c1-packed", "c4-per-gpu-shard-packed"])
def test_full_size_properties(f, h, w, packed):
    """BASELINE.json configs[1] (150 x 720 x 1280) and one GPU's share of configs[4]
    (150 x 1080 x 1920, 3.1e8 pixels: 64-bit base offsets) at full size: too big for the CPU
    oracle in test time, so check size-independent properties: run-to-run agreement and
    additivity of the loss numerator over frame shards (pairs [0,75) + [75,149))."""
    from flowmap_amd import _ops
    from flowmap_amd.model.projection import sample_image_grid  # noqa: F401

    g = torch.Generator(device=DEV).manual_seed(0)
    depth = 1.10 + 0.05 * torch.rand((1, f, h, w), device=DEV, generator=g)
    k = torch.tensor([[0.85 * (h * w) ** 0.5 / w, 0, 0.5], [0, 0.85 * (h * w) ** 0.5 / h, 0.5], [0, 0, 1.0]], device=DEV).expand(1, f, 3, 3).contiguous()
    ff = 0.01 * torch.randn((1, f - 1, h, w, 2), device=DEV, generator=g)
    fb = 0.01 * torch.randn((1, f - 1, h, w, 2), device=DEV, generator=g)
    mf = torch.rand((1, f - 1, h, w), device=DEV, generator=g)
    mb = torch.rand((1, f - 1, h, w), device=DEV, generator=g)
    wts = torch.full((1, f - 1, h, w), 0.5, device=DEV)
    idx = torch.linspace(0, h * w - 1, 1000, dtype=torch.int64, device=DEV)

    def run(lo, hi):  # frames [lo, hi]
        d = depth[:, lo : hi + 1].contiguous().requires_grad_(True)
        rel, _ = _ops.ProcrustesFit.apply(d, k[:, lo : hi + 1].contiguous(), None, wts[:, lo:hi].contiguous(), fb[:, lo:hi].contiguous(), idx)
        ext = _ops.PoseChain.apply(rel)
        rf, rb = _ops.RelativePoses.apply(ext)
        norm = torch.tensor([1.0, 1.0], device=DEV)  # un-normalised numerator
        parts = [x[:, lo:hi].contiguous() for x in (ff, fb, mf, mb)]
        pk = _ops.packed_flow_inputs(*parts, eager=True) if packed else None
        assert (pk is not None) == packed
        loss = _ops.FlowLossFused.apply(d, k[:, lo : hi + 1].contiguous(), rf, rb, *parts, norm, 0, 0.01, True, 0, pk)
        loss.backward()
        return loss.detach().double().cpu(), d.grad

    whole, g_whole = run(0, f - 1)
    again, g_again = run(0, f - 1)
    assert torch.isfinite(whole) and torch.isfinite(g_whole).all()
    assert_close(again, whole, 1e-6, what="run-to-run loss")
    assert_close(g_again, g_whole, 1e-5, what="run-to-run grad")
    a, ga = run(0, 75)
    b, gb = run(75, f - 1)
    assert_close(a + b, whole, 1e-5, what="shard additivity")
    # interior frames of each shard see identical gradients; the halo frame's is the sum
    assert_close(ga[:, :75], g_whole[:, :75], 1e-4, what="shard A grads")
    assert_close(gb[:, 1:], g_whole[:, 76:], 1e-4, what="shard B grads")
    assert_close(ga[:, 75] + gb[:, 0], g_whole[:, 75], 1e-4, what="halo grad")


# ---- function-level call surface on the GPU (same cases as test_hostsim_functions.py) ----
import cases  # noqa: E402


def test_fn_grid_and_unproject():
    cases.case_grid_and_unproject(DEV)


def test_fn_flow_positions():
    cases.case_flow_positions(DEV)


def test_fn_projection_edges():
    cases.case_projection_edges(DEV)


def test_fn_pose_chain():
    cases.case_pose_chain(DEV)


@pytest.mark.parametrize("case", ["generic", "noisy_planar", "few"])
def test_fn_align_rigid(case):
    cases.case_align_rigid(DEV, case)


@pytest.mark.parametrize("lazy", [True, False])
def test_fn_align_surfaces(lazy):
    cases.case_align_surfaces(DEV, lazy)


def test_fn_track_flow():
    cases.case_track_flow(DEV)


@pytest.mark.parametrize("kind", ["huber", "l1", "l2"])
def test_fn_mappings(kind):
    cases.case_mappings(DEV, kind)


def test_tracking_scene_vs_oracle_fp64():
    """Flow + tracking on a consistent scene with several overlapping segments (fused
    tracking kernels, deferred depth scatter) against the fp64 oracle."""
    f, h, w = 12, 40, 56
    sc = orc.synth_scene(f, h, w, seed=4)
    tr = orc.synth_tracks(f, h, w, scene=sc, seed=4, interval=3, radius=4, grid=9)
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(0))
    ours = run_ours(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), 300, tr, device=DEV)
    ref = run_oracle(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), 300, tr, dtype=torch.float64)
    compare(ours, ref, masks=step_masks((h, w), 300, sc["flows"], tr))  # consistent scene: 1e-4 outright, dL/dfocal included


@pytest.mark.parametrize("steps", [1, 5, 149, 700])
def test_pose_chain_scan_matches_serial_oracle(steps):
    """The parallel-scan pose chain (chunks + Hillis-Steele in LDS, additive suffix scan
    in the backward) against the oracle's serial fp64 loop, incl. chunk sizes > 1."""
    from flowmap_amd.model import projection as fm

    g = torch.Generator().manual_seed(steps)
    ang = 0.05 * torch.randn((2, steps, 3), generator=g, dtype=torch.float64)
    rel = torch.eye(4, dtype=torch.float64).repeat(2, steps, 1, 1)
    cx, sx = torch.cos(ang[..., 0]), torch.sin(ang[..., 0])
    cy, sy = torch.cos(ang[..., 1]), torch.sin(ang[..., 1])
    rel[..., 1, 1], rel[..., 1, 2], rel[..., 2, 1], rel[..., 2, 2] = cx, -sx, sx, cx
    ry = torch.eye(4, dtype=torch.float64).repeat(2, steps, 1, 1)
    ry[..., 0, 0], ry[..., 0, 2], ry[..., 2, 0], ry[..., 2, 2] = cy, sy, -sy, cy
    rel = ry @ rel
    rel[..., :3, 3] = 0.02 * torch.randn((2, steps, 3), generator=g, dtype=torch.float64)
    cot = torch.randn((2, steps + 1, 4, 4), generator=g, dtype=torch.float64)
    r64 = rel.clone().requires_grad_(True)
    (orc.chain_poses(r64) * cot).sum().backward()
    r32 = rel.float().to(DEV).requires_grad_(True)
    e = fm.get_extrinsics(r32)
    (e * cot.float().to(DEV)).sum().backward()
    assert_close(e, orc.chain_poses(rel), 1e-6, what="extrinsics")
    assert_close(r32.grad, r64.grad, 1e-5, what="g_rel")


def test_tracking_long_windows_vs_oracle_fp64():
    """41-frame track windows (the reference's radius 20) straddling a 45-frame clip."""
    f, h, w = 45, 48, 64
    sc = orc.synth_scene(f, h, w, seed=6)
    tr = orc.synth_tracks(f, h, w, scene=sc, seed=6, interval=15, radius=20, grid=10)
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(1))
    ours = run_ours(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), 400, tr, device=DEV)
    ref = run_oracle(sc["depth_init"], wl, 0.8, sc["flows"], (h, w), 400, tr, dtype=torch.float64)
    compare(ours, ref, masks=step_masks((h, w), 400, sc["flows"], tr))


@pytest.mark.parametrize("lazy", [True, False])
def test_fn_flow_loss_batched(lazy):
    cases.case_flow_loss_batched(DEV, lazy)


def test_fn_loss_gating_and_empty_tracks():
    cases.case_loss_gating_and_empty_tracks(DEV)


@pytest.mark.parametrize("lazy_weights", [False, True])
def test_fn_softmin_intrinsics(lazy_weights):
    cases.case_softmin_intrinsics(DEV, lazy_weights)


def test_softmin_whole_step():
    cases.case_softmin_step(DEV)


@pytest.mark.parametrize("hw", [(18, 28), (7, 9), (64, 96), (30, 52)])
def test_packed_inputs(hw):
    cases.case_packed_inputs(DEV, hw)


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_fused_adam(weight_decay):
    cases.case_fused_adam(DEV, weight_decay)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_flow_preprocess(tag):
    cases.case_flow_preprocess(DEV, tag)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_cropping(tag):
    cases.case_cropping(DEV, tag)


def test_procrustes_planned_backward():
    cases.case_procrustes_planned_backward(DEV)


def test_fill_and_sparse_store():
    cases.case_fill_and_sparse_store(DEV)


def test_track_scatter_plan():
    cases.case_track_scatter_plan(DEV)


def test_softmin_blend():
    cases.case_softmin_blend(DEV)


def test_focal_intrinsics():
    cases.case_focal_intrinsics(DEV)


def test_export(tmp_path):
    cases.case_export(DEV, tmp_path)


@pytest.mark.parametrize("h,w,flow_sigma", [(90, 122, 0.01), (70, 150, 0.25), (33, 200, 0.08), (16, 64, 0.5), (144, 256, 0.01)])
def test_dense_tiled_procrustes(h, w, flow_sigma):
    cases.case_dense_procrustes(DEV, h, w, flow_sigma)


def test_random_subset():
    cases.case_random_subset(DEV)


def test_capturable_pieces():
    cases.case_capturable_pieces(DEV)


def _graph_problem(f=6, h=48, w=64, softmin=False, tracking=True):
    from flowmap_amd import Batch
    from flowmap_amd.loss import LossFlow, LossFlowCfg, LossTracking, LossTrackingCfg
    from flowmap_amd.loss.mapping import MappingHuberCfg
    from flowmap_amd.model.extrinsics_procrustes import ExtrinsicsProcrustesCfg
    from flowmap_amd.model.intrinsics_softmin import IntrinsicsSoftminCfg
    from flowmap_amd.model.model import BackboneExplicitDepthCfg, IntrinsicsRegressedCfg, Model, ModelCfg
    from helpers import to_flows, to_tracks

    import flowmap_amd

    sc = orc.synth_scene(f, h, w, seed=12)
    flowmap_amd.set_lazy_surfaces(True)
    intr = IntrinsicsSoftminCfg("softmin", 400, 0.5, 2.0, 8, None) if softmin else IntrinsicsRegressedCfg("regressed", 0.9)
    model = Model(ModelCfg(BackboneExplicitDepthCfg("explicit_depth", 1.0, 100.0), intr, ExtrinsicsProcrustesCfg("procrustes", 300, False)),
                  num_frames=f, image_shape=(h, w))
    model.backbone.depth.data = sc["depth_init"].clone()
    model = model.to(DEV)
    flows = to_flows(sc["flows"], DEV)
    tracks = to_tracks(orc.synth_tracks(f, h, w, scene=sc, seed=12, interval=2, radius=2, grid=6), DEV) if tracking else None
    batch = Batch(torch.zeros((1, f, 3, h, w), device=DEV))
    flow_fn = LossFlow(LossFlowCfg(0, 1000.0, "flow", MappingHuberCfg("huber", 0.01)))
    track_fn = LossTracking(LossTrackingCfg(0, 100.0, "tracking", MappingHuberCfg("huber", 0.01)))

    def loss_of(out):
        total = flow_fn(batch, flows, None, out, 0)
        return total + track_fn(batch, flows, tracks, out, 0) if tracking else total

    return model, batch, flows, loss_of


def test_graphed_step_replays_the_eager_optimisation():
    """flowmap_amd.GraphedStep: forward + flow and tracking losses + backward + FusedAdam captured
    in one hipGraph; five replays follow five eager steps parameter for parameter."""
    import flowmap_amd

    try:
        results = {}
        for mode in ("eager", "graph"):
            model, batch, flows, loss_of = _graph_problem()
            opt = flowmap_amd.FusedAdam(model.parameters(), lr=1e-3, capturable=(mode == "graph"))

            def step():
                opt.zero_grad(set_to_none=True)
                loss = loss_of(model(batch, flows, 0))
                loss.backward()
                opt.step()
                return loss

            losses = []
            if mode == "eager":
                for _ in range(3 + 5):  # the graph variant runs 3 eager warm-up steps; the capture itself executes nothing
                    losses.append(float(step().detach()))
            else:
                graphed = flowmap_amd.GraphedStep(step, warmup=3)
                try:
                    for _ in range(5):
                        losses.append(float(graphed()))
                finally:
                    graphed.close()
            results[mode] = (losses[-5:], model.backbone.depth.detach().clone(), model.intrinsics.focal_length.detach().clone())
        assert_close(torch.tensor(results["graph"][0]), torch.tensor(results["eager"][0]), 1e-5, what="loss history")
        assert results["eager"][0][-1] < results["eager"][0][0]  # it optimises
        assert_close(results["graph"][1], results["eager"][1], 1e-5, what="depth")
        assert_close(results["graph"][2], results["eager"][2], 1e-5, what="focal length")
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def test_graphed_step_with_softmin_samples_afresh():
    """Inside a hipGraph the softmin sweep still draws new pixels on every replay (device-side
    sampler state): the loss changes from replay to replay although no parameter does."""
    import flowmap_amd

    try:
        model, batch, flows, loss_of = _graph_problem(softmin=True, tracking=False)

        def step():
            model.zero_grad(set_to_none=True)
            out = model(batch, flows, 0)
            loss = loss_of(out)
            loss.backward()
            return out.intrinsics

        graphed = flowmap_amd.GraphedStep(step, warmup=2)
        try:
            ks = [graphed()[0, 0].clone() for _ in range(4)]
        finally:
            graphed.close()
        assert all(torch.isfinite(k).all() for k in ks)
        assert len({round(float(k[0, 0]), 7) for k in ks}) > 1  # different samples -> slightly different blends
        assert model.backbone.depth.grad is not None and torch.isfinite(model.backbone.depth.grad).all()
    finally:
        flowmap_amd.set_lazy_surfaces(False)


def test_grad_arena():
    cases.case_grad_arena(DEV)


def test_second_backward_and_autograd_grad():
    cases.case_second_backward(DEV)


def test_backward_on_worker_threads_with_grad_hooks():
    cases.case_threads_and_hooks(DEV)


@pytest.mark.parametrize(
    "f,h,w,packed,kind",
    [(6, 48, 64, True, "huber"), (6, 48, 64, False, "huber"), (5, 90, 122, False, "huber"), (4, 64, 96, True, "l1"),
     (4, 64, 96, False, "l2"), (3, 720, 1280, True, "huber"), (3, 720, 1280, False, "huber")],
)
def test_flow_fused_leaves(f, h, w, packed, kind):
    """Per-frame dL/dK, dL/dT_fwd/bwd, dL/ddepth of the fused flow kernel with a general K and free
    poses, on the GPU (DPP wave sums, fp64 LDS/atomic reduction, flow_finalize_frame), both input
    layouts, a width that is not a multiple of 4, all three mappings, and a 720p frame."""
    cases.case_flow_fused_leaves(DEV, f, h, w, packed, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["huber", "l1", "l2"])
def test_ghost_terms(kind):
    cases.case_ghost_terms(DEV, kind)


@pytest.mark.gpu
def test_depth_adam_update_inside_the_flow_pass_with_the_tap_exchange():
    cases.case_in_pass_adam(DEV, steps=60, lr=3e-4, exchange=True)


@pytest.mark.gpu
def test_tap_exchange():
    cases.case_tap_exchange(DEV)


@pytest.mark.gpu
def test_lazy_extrinsics():
    cases.case_lazy_extrinsics(DEV)


@pytest.mark.gpu
def test_depth_adam_update_inside_the_flow_pass_follows_torch_adam():
    cases.case_in_pass_adam(DEV, steps=200, lr=3e-4)  # 10x the reference's learning rate (config/overfit.yaml:30)


@pytest.mark.gpu
def test_in_pass_adam_with_the_tap_exchange_is_loud_about_unequal_upstreams():
    cases.case_in_pass_adam_exchange_unequal_upstreams(DEV)


@pytest.mark.gpu
def test_in_pass_adam_update_refuses_what_it_cannot_do():
    cases.case_in_pass_adam_refusals(DEV)


def test_noncontiguous_views_are_copied_loudly():
    cases.case_views_are_copied_loudly(DEV)


def test_depth_adam_update_inside_the_flow_pass_with_the_softmin_sweep():
    cases.case_in_pass_adam(DEV, steps=60, lr=3e-4, softmin=True)


def test_one_launch_fit_agrees_with_the_three_launch_form_under_stress():
    """fm_procrustes_fit_chain (one block per pair, last-block election for the pose chain) against fm_procrustes_fit +
    fm_pose_chain_fwd over 1 200 back-to-back launches on changing inputs; the persistent workspace stays clean."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("fit_stress", Path(__file__).resolve().parent.parent / "tools" / "fit_stress.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(300, dev=str(DEV), verbose=False) < 1e-5


def test_pretraining_mode_never_packs_or_plans():
    cases.case_pretraining_mode(DEV)


def test_frame_windows_are_read_in_place():
    cases.case_frame_windows(DEV)


def test_halo_exchange_kernels():
    cases.case_halo_kernels(DEV)


def test_a_step_launches_no_stray_torch_kernels():
    cases.case_step_torch_ops(DEV)


def test_losses_seed_their_own_backward():
    cases.case_root_loss(DEV)
