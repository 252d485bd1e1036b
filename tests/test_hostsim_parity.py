"""CPU parity of the flowmap_amd host layer + the kernels' per-element math (through the
tests/host_sim double of the C ABI) against the oracle and the reference's golden
vectors.  What this cannot see (launch geometry, atomics, wave reductions) is covered
by the `-m gpu` twins in test_gpu_parity.py."""

import numpy as np
import pytest
import torch

import flowmap_amd
from flowmap_amd import _lib
from conftest import assert_close, load_golden, t
from helpers import build_host_sim, compare_step, run_oracle, run_ours, step_masks
from oracle import flowmap_oracle as orc
from test_oracle_golden import _flows, _tracks


@pytest.fixture(autouse=True, scope="module")
def host_double():
    _lib.set_library_for_testing(build_host_sim())
    yield
    _lib.set_library_for_testing(None)


def compare(ours, truth, ref32=None, masks=None):
    compare_step(ours, truth, ref32, masks=masks)


@pytest.mark.parametrize("lazy", [True, False])
@pytest.mark.parametrize(
    "name,kind",
    [("step_iid_flow", "huber"), ("step_scene_flow_tracking", "huber"), ("step_iid_l1_odd", "l1"), ("step_iid_l2_odd", "l2")],
)
def test_step_vs_reference_golden(name, kind, lazy):
    g = load_golden(name)
    depth, wlogit = t(g["depth"]), t(g["wlogit"])
    npts = int(g["num_points"])
    ours = run_ours(depth, wlogit, float(g["focal"]), _flows(g), depth.shape[1:], None if npts < 0 else npts, _tracks(g), kind, lazy=lazy)
    golden = {k: t(g[k]) for k in ("total", "loss_flow", "loss_tracking", "extrinsics", "g_depth", "g_wlogit", "g_focal")}
    # the truth is the fp64 oracle on the golden inputs; the reference's fp32 golden values give its own gap
    truth = run_oracle(depth, wlogit, float(g["focal"]), _flows(g), depth.shape[1:], None if npts < 0 else npts, _tracks(g), kind,
                       dtype=torch.float64)
    for key in ("total", "loss_flow", "loss_tracking", "extrinsics"):  # values: straight against the reference's own numbers too
        assert_close(ours[key], golden[key], 1e-4, what=f"{key} vs golden")
    compare(ours, truth, golden, masks=step_masks(depth.shape[1:], None if npts < 0 else npts, _flows(g), _tracks(g)))


@pytest.mark.parametrize("f,h,w,p", [(4, 16, 24, 64), (3, 9, 13, None), (6, 32, 20, 200), (2, 8, 12, 30), (2, 8, 12, None)])
def test_step_vs_oracle_fp64(f, h, w, p):
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=f * 7 + h)
    ours = run_ours(depth, wlogit, 0.85, flows, (h, w), p)
    truth = run_oracle(depth, wlogit, 0.85, flows, (h, w), p, dtype=torch.float64)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), p, dtype=torch.float32)
    compare(ours, truth, ref32, masks=step_masks((h, w), p, flows))


def test_loss_scale_and_carry():
    """grad_output != 1 exercises fm_scale_if_needed; results must scale linearly and the
    carried-gradient path must equal the plain dense path."""
    f, h, w = 4, 12, 16
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=3)
    a = run_ours(depth, wlogit, 0.85, flows, (h, w), 40)
    b = run_ours(depth, wlogit, 0.85, flows, (h, w), 40, loss_scale=0.5)
    assert_close(b["g_depth"], 0.5 * a["g_depth"], 1e-6)
    assert_close(b["g_wlogit"], 0.5 * a["g_wlogit"], 1e-6)
    from flowmap_amd.loss import LossFlow

    LossFlow.carry_depth_grad = False
    try:
        c = run_ours(depth, wlogit, 0.85, flows, (h, w), 40)
    finally:
        LossFlow.carry_depth_grad = True
    assert_close(c["g_depth"], a["g_depth"], 1e-6)


def test_zero_masks_give_zero_loss():
    f, h, w = 3, 8, 12
    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=1)
    flows.forward_mask.zero_()
    flows.backward_mask.zero_()
    ours = run_ours(depth, wlogit, 0.85, flows, (h, w), 30)
    assert float(ours["total"]) == 0.0
    assert float(ours["g_depth"].abs().max()) == 0.0


def test_full_size_module_logic_at_toy_size():
    """tests/test_gpu_full_size.py (C1 / C2 at 150 x 720 x 1280 on the GPU) driven at 12 x 40 x 56 through
    the host double: one oracle forward, one backward per loss, masked + element-wise comparison."""
    import test_gpu_full_size as full

    built = full.build_reference(12, 40, 56, 300, "cpu", torch.float64, interval=3, radius=4, grid=9)
    full.compare_flow_only(built, (40, 56), 300, "cpu")
    full.compare_flow_and_tracking(built, (40, 56), 300, "cpu")
    full.compare_flow_and_tracking(built, (40, 56), 300, "cpu", steps=3, label="C2-tap-exchange")  # (the third step: tap exchange + tap image)


def test_full_size_cases_run_small_on_the_host_double():
    """The bodies of the C3 / C4-shard full-size GPU tests (tests/test_gpu_full_size.py) at a size that takes a second,
    on the host double: the comparison code itself is exercised by the CPU suite."""
    import warnings

    import test_gpu_full_size as full

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full.c3_case(12, 40, 56, 200, "cpu", torch.float64, "C3 body, small")
        record = full.c4_shard_case(6, 24, 32, 60, "cpu", torch.float64, "C4-shard body, small")
    assert record["oracle_dtype"] == "torch.float64" and "g_depth_fp32_reference_gap" in record
