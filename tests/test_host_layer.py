"""Host-layer behaviour that needs no kernels at all, or only the host double: lazy handles,
broadcast flattening, caches, error behaviour."""

import pytest
import torch

import flowmap_amd
from flowmap_amd import _lib
from flowmap_amd.loss.loss import or_one
from flowmap_amd.model import projection as fm
from helpers import build_host_sim


@pytest.fixture(autouse=True, scope="module")
def host_double():
    _lib.set_library_for_testing(build_host_sim())
    yield
    _lib.set_library_for_testing(None)


def test_or_one_matches_python_or():
    assert float(or_one(torch.tensor(0.0))) == 1.0
    assert float(or_one(torch.tensor(3.5))) == 3.5
    assert int(or_one(torch.tensor(0))) == 1 and int(or_one(torch.tensor(7))) == 7


def test_grid_is_cached_and_matches_reference_arithmetic():
    a, ia = fm.sample_image_grid((5, 7))
    b, _ = fm.sample_image_grid((5, 7))
    assert a is b  # identity is what lets unproject recognise the canonical grid
    assert a.shape == (5, 7, 2) and ia.shape == (5, 7, 2) and ia.dtype == torch.int64
    assert torch.equal(a[2, 3], torch.tensor([(3 + 0.5) / 7, (2 + 0.5) / 5], dtype=torch.float32))
    assert torch.equal(ia[2, 3], torch.tensor([2, 3]))


def test_lazy_surfaces_quack_and_materialise():
    z = torch.rand((1, 3, 4, 6)) + 1
    k = torch.eye(3).expand(1, 3, 3, 3).contiguous()
    xy, _ = fm.sample_image_grid((4, 6))
    dense = fm.unproject(xy, z, k[:, :, None, None])
    fm.set_lazy_surfaces(True)
    try:
        lazy = fm.unproject(xy, z, k[:, :, None, None])
        not_canonical = fm.unproject(xy.clone(), z, k[:, :, None, None])  # a different grid object: stays dense
    finally:
        fm.set_lazy_surfaces(False)
    assert isinstance(lazy, fm.LazySurfaces) and isinstance(not_canonical, torch.Tensor)
    assert lazy.shape == dense.shape and lazy.device == dense.device and lazy.ndim == 5 and lazy.dtype == torch.float32
    sl = lazy[:, 1:3]  # frame slice stays lazy (loss_tracking.py:48)
    assert isinstance(sl, fm.LazySurfaces) and sl.shape == (1, 2, 4, 6, 3)
    assert torch.allclose(lazy.materialize(), dense)
    assert torch.allclose(torch.sum(lazy, dim=-1), dense.sum(-1))  # torch function -> materialises
    assert torch.allclose(lazy.mean(), dense.mean())  # attribute access -> materialises
    assert torch.allclose(lazy[0, 1, 2], dense[0, 1, 2])  # general indexing -> materialises


def test_lazy_weights_materialise_to_sigmoid():
    logits = torch.randn((1, 2, 3, 4))
    lw = fm.LazyWeights(logits, 100.0)
    assert lw.shape == logits.shape
    assert torch.allclose(lw.materialize(), (100 * logits).sigmoid())
    assert torch.allclose(torch.ones_like(lw), torch.ones_like(logits))


def test_unproject_broadcast_patterns():
    g = torch.Generator().manual_seed(0)
    xy = torch.rand((5, 2), generator=g)
    z = torch.rand((2, 3, 5), generator=g) + 1
    k = torch.eye(3).repeat(2, 3, 1, 1) + 0.1 * torch.rand((2, 3, 3, 3), generator=g)
    ours = fm.unproject(xy, z, k[:, :, None])
    ref = (torch.linalg.inv(k)[:, :, None] @ torch.cat([xy, torch.ones(5, 1)], -1)[..., None])[..., 0] * z[..., None]
    assert torch.allclose(ours, ref, atol=1e-5)
    # per-group coordinates
    xyg = torch.rand((2, 3, 5, 2), generator=g)
    ours = fm.unproject(xyg, z, k[:, :, None])
    ref = (torch.linalg.inv(k)[:, :, None] @ torch.cat([xyg, torch.ones(2, 3, 5, 1)], -1)[..., None])[..., 0] * z[..., None]
    assert torch.allclose(ours, ref, atol=1e-5)


def test_dtype_and_shape_errors_are_loud():
    with pytest.raises(RuntimeError, match="float32"):
        fm.get_extrinsics(torch.eye(4, dtype=torch.float64).repeat(1, 2, 1, 1))
    z = torch.rand((1, 3, 4, 6))
    with pytest.raises(RuntimeError, match="shapes do not match"):
        fm.align_surfaces(torch.rand((1, 3, 4, 6, 3)), torch.zeros((1, 2, 4, 5, 2)), torch.ones((1, 2, 4, 6)), None)
    with pytest.raises(RuntimeError, match="int64"):
        fm.align_surfaces(torch.rand((1, 3, 4, 6, 3)), torch.zeros((1, 2, 4, 6, 2)), torch.ones((1, 2, 4, 6)), torch.arange(4, dtype=torch.int32))
    flow = torch.zeros((1, 2, 4, 6, 2), requires_grad=True)
    with pytest.raises(RuntimeError, match="optical flow"):
        fm.align_surfaces(torch.rand((1, 3, 4, 6, 3)), flow, torch.ones((1, 2, 4, 6)), None)
    assert z is not None


def test_version_bump_invalidates_mask_norm_cache():
    from flowmap_amd import _ops

    m1, m2 = torch.rand((1, 2, 3, 4)), torch.rand((1, 2, 3, 4))
    a = _ops.flow_valid_norm(m1, m2, 10.0)
    assert _ops.flow_valid_norm(m1, m2, 10.0) is a  # cached
    m1.mul_(0.5)  # in-place edit bumps the version counter
    b = _ops.flow_valid_norm(m1, m2, 10.0)
    assert b is not a and abs(float(b[1]) - float(m1.sum() + m2.sum())) < 1e-4


def test_errors_of_the_newer_entry_points_are_loud():
    from flowmap_amd import FusedAdam, Tracks, _ops, export

    videos = torch.rand((1, 3, 3, 8, 10))
    with pytest.raises(RuntimeError, match="video's resolution"):
        _ops.flow_postprocess(videos, torch.zeros((1, 2, 4, 5, 2)), (4, 5), reverse=False)
    with pytest.raises(RuntimeError, match="batch, frame, 3"):
        _ops.consistency_mask(torch.rand((1, 3, 8, 10)), torch.zeros((1, 2, 8, 10, 2)))
    with pytest.raises(RuntimeError, match="1 <= count <= n"):
        _ops.random_subset(10, 11, "cpu")
    with pytest.raises(RuntimeError, match="frame, height, width"):
        export.world_point_cloud(torch.rand((1, 3, 4, 6)), torch.eye(3).repeat(3, 1, 1), torch.eye(4).repeat(3, 1, 1))
    with pytest.raises(RuntimeError, match="do not match"):
        export.world_point_cloud(torch.rand((3, 4, 6)), torch.eye(3).repeat(2, 1, 1), torch.eye(4).repeat(3, 1, 1))
    with pytest.raises(ValueError, match="amsgrad"):
        FusedAdam([torch.zeros(3, requires_grad=True)], amsgrad=True)
    p = torch.zeros(4, dtype=torch.float64, requires_grad=True)
    p.grad = torch.ones_like(p)
    with pytest.raises(RuntimeError, match="float32"):
        FusedAdam([p]).step()
    with pytest.raises(RuntimeError, match="LeadingFrames expects"):
        _ops.LeadingFrames.apply(torch.rand((2, 3, 4)), 1)
    # tracking: a segment past the last frame, and a depth window outside the video
    depth, k, ext = torch.rand((1, 3, 6, 8)) + 1, torch.eye(3).repeat(1, 3, 1, 1), torch.eye(4).repeat(1, 3, 1, 1)
    tracks = [Tracks(torch.rand((1, 4, 5, 2)), torch.ones((1, 4, 5), dtype=torch.bool), 0)]
    with pytest.raises(RuntimeError, match="past the last frame"):
        _ops.TrackLossFused.apply(depth, k, ext, _ops.pack_tracks(tracks, "cpu"), 1.0, 0, 0.01, False)
    short = [Tracks(torch.rand((1, 2, 5, 2)), torch.ones((1, 2, 5), dtype=torch.bool), 0)]
    with pytest.raises(RuntimeError, match="window of it"):
        _ops.TrackLossFused.apply(depth, k, ext, _ops.pack_tracks(short, "cpu"), 1.0, 0, 0.01, False, 2)


def test_packed_tracks_ownership_filters_sources():
    from flowmap_amd import Tracks, _ops

    tracks = [Tracks(torch.rand((1, 7, 3, 2)), torch.ones((1, 7, 3), dtype=torch.bool), 2),
              Tracks(torch.rand((1, 4, 3, 2)), torch.ones((1, 4, 3), dtype=torch.bool), 8)]
    whole = _ops.PackedTracks(tracks, "cpu")
    assert whole.nblocks == 11 and not whole.partial and whole.last_frame == 12
    # frame-major order of the (segment, frame) blocks
    frames = [int(whole.seg[s, 0]) + fr for s, fr in whole.blocks.tolist()]
    assert frames == sorted(frames)
    own = _ops.PackedTracks(tracks, "cpu", own=(4, 9))  # frames 4..8: five of segment 0, one of segment 1
    assert own.partial and own.nblocks == 6
    assert {(s, fr) for s, fr in own.blocks.tolist()} == {(0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (1, 0)}
    # tiles of TRACK_TILE source frames are kept when ANY of their frames is owned
    assert own.tiles.tolist() == [[0, 0], [0, _ops.TRACK_TILE], [1, 0]] if _ops.TRACK_TILE < 7 else True
    none = _ops.PackedTracks(tracks, "cpu", own=(20, 30))
    assert none.nblocks == 0 and none.ntiles == 0 and tuple(none.blocks.shape) == (0, 2)


def test_derived_constants_live_on_their_tensor_and_follow_its_version():
    """No module-level caches: what is derived from a constant input (packed flows, valid sums, plans,
    K^-1) is kept on the input tensor object, rebuilt after an in-place edit, gone with the tensor."""
    import gc
    import weakref

    from flowmap_amd import _ops

    mf, mb = torch.rand((1, 3, 8, 12)), torch.rand((1, 3, 8, 12))
    norm = _ops.flow_valid_norm(mf, mb, 1000.0)
    assert _ops.flow_valid_norm(mf, mb, 1000.0) is norm  # same tensors, same versions
    assert _ops.flow_valid_norm(mf, mb, 10.0) is not norm  # another weight
    again = _ops.flow_valid_norm(mf, mb, 1000.0)
    mb.mul_(0.5)  # in-place edit of the OTHER mask: rebuilt
    assert not torch.equal(_ops.flow_valid_norm(mf, mb, 1000.0), again)
    assert not [name for name in vars(_ops) if name.endswith("_cache")]  # nothing module-level left
    held = weakref.ref(_ops.flow_valid_norm(mf, mb, 1000.0))
    del mf, norm, again
    gc.collect()
    assert held() is None  # the derived tensor died with its owner


def test_freeze_gc_moves_live_objects_to_the_permanent_generation():
    import gc

    import flowmap_amd

    try:
        frozen = flowmap_amd.freeze_gc()
        assert frozen > 1000 and gc.get_freeze_count() == frozen
    finally:
        gc.unfreeze()
    assert gc.get_freeze_count() == 0


def test_intrinsics_inverse_cache_follows_object_version_and_views():
    from flowmap_amd import _ops

    k = torch.eye(3).repeat(2, 3, 1, 1)
    k[..., 0, 0], k[..., 1, 1], k[..., :2, 2] = 0.9, 1.2, 0.5
    first = _ops.intrinsics_inverse(k)
    assert _ops.intrinsics_inverse(k).data_ptr() == first.data_ptr()  # same object, same version
    view = k[:, :, None, None].reshape(2, 3, 3, 3)
    assert view._base is k and _ops.intrinsics_inverse(view).data_ptr() == first.data_ptr()  # a view of it
    part = _ops.intrinsics_inverse(k[1:])  # a view of other memory: its own inverse
    assert part.data_ptr() != first.data_ptr() and torch.equal(part, first[1:])
    assert _ops.intrinsics_inverse(k).data_ptr() == first.data_ptr()  # ... beside the whole tensor's, not instead of it
    k[..., 0, 0] *= 2  # in-place update: version bump, recomputed
    second = _ops.intrinsics_inverse(k)
    assert not torch.equal(second, first) and torch.allclose(second @ k, torch.eye(3).expand(2, 3, 3, 3), atol=1e-6)
    clone = k.clone()  # equal values, different object: never served another tensor's entry
    assert _ops.intrinsics_inverse(clone).data_ptr() != second.data_ptr()
    # the inverse lives on the K tensor object: it goes when K goes
    import gc
    import weakref

    held = weakref.ref(_ops.intrinsics_inverse(clone))
    del clone
    gc.collect()
    assert held() is None


def test_errors_of_the_preparation_and_intrinsics_entry_points_are_loud():
    from flowmap_amd import Batch, _ops
    from flowmap_amd.misc import cropping

    videos = torch.rand((1, 2, 3, 12, 16))
    with pytest.raises(RuntimeError, match="crop must fit"):
        _ops.resize_crop(videos, (8, 8), (9, 8))
    with pytest.raises(RuntimeError, match="float32"):
        _ops.resize_crop(videos.double(), (8, 8), (8, 8))
    out = _ops.resize_crop(videos, (12, 16), (12, 16))  # identity resize, no crop: a copy
    assert torch.equal(out, videos) and out.data_ptr() != videos.data_ptr()
    assert cropping.get_image_shape((37, 53), cropping.CroppingCfg(400, 2, 4)) == (17, 24)
    assert cropping.compute_patch_cropped_shape((17, 24), 4) == (16, 24)
    assert cropping.center_crop_intrinsics(None, (4, 4), (2, 2)) is None
    with pytest.raises(RuntimeError, match="float32"):
        _ops.focal_intrinsics(torch.tensor(0.9, dtype=torch.float64), (1, 2), (8, 8))
    k = _ops.focal_intrinsics(torch.tensor(0.9), (1, 2), (8, 8))  # no gradient asked for: still fine, nothing parked
    assert k.shape == (1, 2, 3, 3) and k.grad_fn is None
    # the fused softmin tail: constants must not carry gradients, shapes are checked
    depth, w, bwd = torch.rand((1, 2, 6, 8)) + 1, torch.rand((1, 1, 6, 8)), torch.zeros((1, 1, 6, 8, 2))
    idx, cand, rel = torch.arange(5), torch.eye(3).repeat(4, 1, 1), torch.eye(4).repeat(4, 1, 1)
    with pytest.raises(RuntimeError, match="constants"):
        _ops.softmin_intrinsics(depth, w, bwd, idx, cand.clone().requires_grad_(True), rel, 0.0, 3)
    with pytest.raises(RuntimeError, match="poses"):
        _ops.softmin_intrinsics(depth, w, bwd, idx, cand, rel[:3], 0.0, 3)
    k_soft, soft = _ops.softmin_intrinsics(depth, w, bwd, idx, cand, rel, 0.0, 3)
    assert k_soft.shape == (1, 3, 3, 3) and soft.shape == (1, 4) and abs(float(soft.sum()) - 1) < 1e-6
    assert Batch(videos).intrinsics is None


def test_options_live_in_one_object():
    """flowmap_amd.config (round 6): the run-time switches are fields of ONE object — set through install(options=...) / configure(), scoped with
    override(); `_ops` reads that very object at call time; an unknown name raises instead of silently doing nothing."""
    import flowmap_amd
    from flowmap_amd import _ops, config

    assert _ops.options is config.options and config.options.tap_exchange is True and config.options.tap_exchange_min_bytes == 128 << 20
    with config.override(tap_exchange=False, tap_exchange_min_bytes=0) as inside:
        assert inside is config.options and not _ops.options.tap_exchange and _ops.options.tap_exchange_min_bytes == 0
        with pytest.raises(AttributeError):
            config.configure(tap_exchnge=True)
    assert config.options.tap_exchange and config.options.tap_exchange_min_bytes == 128 << 20  # restored
    with pytest.raises(AttributeError):
        with config.override(no_such_option=1):
            pass
    assert not hasattr(_ops, "use_tap_exchange") and not hasattr(_ops, "use_unit_seed")  # (the module-level variables of rounds 1-5 are gone)
    assert config.defaults() == config.Options()
    import inspect

    assert "options" in inspect.signature(flowmap_amd.install).parameters
