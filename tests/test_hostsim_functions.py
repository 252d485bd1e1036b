"""Function-level call surface (flowmap_amd.model.projection / procrustes / loss.mapping)
against the reference's golden vectors — CPU, through the host test double."""

import pytest

import cases
from flowmap_amd import _lib
from helpers import build_host_sim


@pytest.fixture(autouse=True, scope="module")
def host_double():
    _lib.set_library_for_testing(build_host_sim())
    yield
    _lib.set_library_for_testing(None)


def test_grid_and_unproject():
    cases.case_grid_and_unproject("cpu")


def test_flow_positions():
    cases.case_flow_positions("cpu")


def test_projection_edges():
    cases.case_projection_edges("cpu")


def test_pose_chain():
    cases.case_pose_chain("cpu")


@pytest.mark.parametrize("case", ["generic", "noisy_planar", "few"])
def test_align_rigid(case):
    cases.case_align_rigid("cpu", case)


@pytest.mark.parametrize("lazy", [True, False])
def test_align_surfaces(lazy):
    cases.case_align_surfaces("cpu", lazy)


def test_track_flow():
    cases.case_track_flow("cpu")


@pytest.mark.parametrize("kind", ["huber", "l1", "l2"])
def test_mappings(kind):
    cases.case_mappings("cpu", kind)


@pytest.mark.parametrize("lazy", [True, False])
def test_flow_loss_batched(lazy):
    cases.case_flow_loss_batched("cpu", lazy)


def test_loss_gating_and_empty_tracks():
    cases.case_loss_gating_and_empty_tracks("cpu")


@pytest.mark.parametrize("lazy_weights", [False, True])
def test_softmin_intrinsics(lazy_weights):
    cases.case_softmin_intrinsics("cpu", lazy_weights)


def test_softmin_whole_step():
    cases.case_softmin_step("cpu")


@pytest.mark.parametrize("hw", [(18, 28), (7, 9)])
def test_packed_inputs(hw):
    cases.case_packed_inputs("cpu", hw)


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_fused_adam(weight_decay):
    cases.case_fused_adam("cpu", weight_decay)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_flow_preprocess(tag):
    cases.case_flow_preprocess("cpu", tag)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_cropping(tag):
    cases.case_cropping("cpu", tag)


def test_procrustes_planned_backward():
    cases.case_procrustes_planned_backward("cpu")


def test_fill_and_sparse_store():
    cases.case_fill_and_sparse_store("cpu")


def test_track_scatter_plan():
    cases.case_track_scatter_plan("cpu")


def test_softmin_blend():
    cases.case_softmin_blend("cpu")


def test_focal_intrinsics():
    cases.case_focal_intrinsics("cpu")


def test_export(tmp_path):
    cases.case_export("cpu", tmp_path)


def test_random_subset():
    cases.case_random_subset("cpu")


def test_capturable_pieces():
    cases.case_capturable_pieces("cpu")


@pytest.mark.parametrize("f,h,w,packed,kind", [(5, 12, 16, True, "huber"), (5, 12, 16, False, "huber"), (4, 9, 13, False, "l1"),
                                                (3, 10, 12, True, "l2")])
def test_flow_fused_leaves(f, h, w, packed, kind):
    cases.case_flow_fused_leaves("cpu", f, h, w, packed, kind)


@pytest.mark.parametrize("h,w,flow_sigma", [(20, 70, 0.02), (18, 66, 0.3), (33, 130, 0.08)])
def test_dense_procrustes(h, w, flow_sigma):
    cases.case_dense_procrustes("cpu", h, w, flow_sigma, f=3)


def test_grad_arena():
    cases.case_grad_arena("cpu")


def test_second_backward_and_autograd_grad():
    cases.case_second_backward("cpu")


def test_backward_on_worker_threads_with_grad_hooks():
    cases.case_threads_and_hooks("cpu")


def test_depth_adam_update_inside_the_flow_pass_follows_torch_adam():
    cases.case_in_pass_adam("cpu")


@pytest.mark.parametrize("kind", ["huber", "l1", "l2"])
def test_ghost_terms(kind):
    cases.case_ghost_terms("cpu", kind)


def test_depth_adam_update_inside_the_flow_pass_with_the_tap_exchange():
    cases.case_in_pass_adam("cpu", steps=24, exchange=True)


def test_tap_exchange():
    cases.case_tap_exchange("cpu")


def test_in_pass_adam_with_the_tap_exchange_is_loud_about_unequal_upstreams():
    cases.case_in_pass_adam_exchange_unequal_upstreams("cpu")


def test_in_pass_adam_update_refuses_what_it_cannot_do():
    cases.case_in_pass_adam_refusals("cpu")


def test_noncontiguous_views_are_copied_loudly():
    cases.case_views_are_copied_loudly("cpu")


def test_depth_adam_update_inside_the_flow_pass_with_the_softmin_sweep():
    cases.case_in_pass_adam("cpu", steps=24, softmin=True)


def test_pretraining_mode_never_packs_or_plans():
    cases.case_pretraining_mode("cpu")


def test_frame_windows_are_read_in_place():
    cases.case_frame_windows("cpu")


def test_halo_exchange_kernels():
    cases.case_halo_kernels("cpu")


def test_a_step_launches_no_stray_torch_kernels():
    cases.case_step_torch_ops("cpu")


def test_losses_seed_their_own_backward():
    cases.case_root_loss("cpu")


def test_dense_backward_selector_by_flow_roughness():
    """Which dense Procrustes backward a flow tensor gets (flowmap_amd/_ops.py: _dense_flow_is_rough): the fused one-pass kernel keeps a
    40 x 80 window of the earlier frame per 32 x 64 tile, displaced by the flow at the tile's centre — on i.i.d. flows (BASELINE configs[4]:
    N(0, 0.01^2) of the image size, i.e. +-13 / +-19 pixels at 720p / 1080p) nearly every tap leaves it and goes to memory one atomic at a
    time: 7.9 ms against 2.2 for the planned pair of kernels at C1 (profiles/r03_dense_microbench.txt), so those flows MUST take the plan;
    a consistent scene's flow (smooth inside a tile) must not pay for the lists."""
    import torch

    from flowmap_amd import _ops
    from oracle import flowmap_oracle as orc

    g = torch.Generator().manual_seed(4)
    for h, w in ((720, 1280), (1080, 1920)):
        iid = 0.01 * torch.randn((1, 2, h, w, 2), generator=g)
        assert _ops._dense_flow_is_rough(iid, h, w), (h, w)
    h, w = 360, 640
    scene = orc.synth_scene(3, h, w, seed=2)["flows"].backward
    assert not _ops._dense_flow_is_rough(scene.contiguous(), h, w)
    # tiles with a NaN in the flow (no usable window centre) count as rough; a handful of rough tiles below the threshold does not flip the choice
    broken = scene.clone()
    broken[0, 0] = float("nan")  # (one of the two pairs: half of the tiles)
    assert _ops._dense_flow_is_rough(broken, h, w)
    few = scene.clone()
    few[0, 0, :32, :64] += 0.2 * torch.randn((32, 64, 2), generator=g)
    assert not _ops._dense_flow_is_rough(few, h, w)  # (one of 240 tiles: far below dense_plan_rough_tiles = 0.25)


def test_packed_tracks_follow_their_tensors():
    """pack_tracks keeps the packed form of a track list on its first coordinate tensor and, from the second call on, only compares what
    could have changed: the same list gives the same object; an in-place edit of a segment's positions or visibility, a replaced
    segment, another start frame or another device / ownership gives a new one."""
    import torch

    from flowmap_amd import _ops
    from helpers import to_tracks
    from oracle import flowmap_oracle as orc

    sc = orc.synth_scene(6, 16, 24, seed=3)
    tracks = to_tracks(orc.synth_tracks(6, 16, 24, scene=sc, seed=3, interval=2, radius=2, grid=4), "cpu")
    first = _ops.pack_tracks(tracks, torch.device("cpu"))
    assert _ops.pack_tracks(tracks, torch.device("cpu")) is first and _ops.pack_tracks(tracks, torch.device("cpu")) is first
    tracks[1].xy.mul_(1.0)  # (version counter moves)
    second = _ops.pack_tracks(tracks, torch.device("cpu"))
    assert second is not first and _ops.pack_tracks(tracks, torch.device("cpu")) is second
    tracks[-1].visibility.logical_and_(tracks[-1].visibility)
    third = _ops.pack_tracks(tracks, torch.device("cpu"))
    assert third is not second
    tracks[0] = type(tracks[0])(tracks[0].xy.clone(), tracks[0].visibility, tracks[0].start_frame)
    fourth = _ops.pack_tracks(tracks, torch.device("cpu"))
    assert fourth is not third and _ops.pack_tracks(tracks, torch.device("cpu")) is fourth
    owned = _ops.pack_tracks(tracks, torch.device("cpu"), own=(0, 3))
    assert owned is not fourth and _ops.pack_tracks(tracks, torch.device("cpu"), own=(0, 3)) is owned
    fifth = _ops.pack_tracks(tracks, torch.device("cpu"))  # (one form is kept per track list: back to the whole video builds it again)
    assert fifth is not owned and _ops.pack_tracks(tracks, torch.device("cpu")) is fifth
    assert _ops.pack_tracks(list(tracks), torch.device("cpu")) is fifth  # (another list of the same segments: the full key finds it)


def test_lazy_extrinsics():
    cases.case_lazy_extrinsics("cpu")
