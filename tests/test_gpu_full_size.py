"""Parity at the sizes the headline numbers are quoted on: BASELINE.json configs[1] (150 frames @
720x1280, flow loss, Procrustes P = 1000) and configs[2] (+ tracking: 30 segments x 1225 tracks, the
layout of flowmap/tracking/__init__.py:49-70), HIP path vs the oracle run ONCE on the GPU box's host
cores (about a minute: one forward, one backward per loss).  The scene is the consistent one of SURVEY.md §8d (generated
on the GPU by the oracle's own functions, seconds instead of minutes), so every gradient —
dL/dfocal included — is well conditioned and held to 1e-4; dL/ddepth is also compared element-wise
and on the pixels the Procrustes fit / the tracks write to.

Round 3: BASELINE.json configs[3] at its own size (65 frames @ 1080x1920, consistent scene, flow loss)
and ONE GPU's shard of configs[4] (150 frames @ 1080x1920, i.i.d. depth / flow / masks, flow loss) against the oracle.  On
i.i.d. inputs the fp32 reference itself is further than 1e-4 from the fp64 truth in the Procrustes-conditioned gradients, so
that test runs the oracle twice (fp64 = truth, fp32 = the reference's own arithmetic), records both gaps and holds ours to
max(1e-4, 2 x the reference's gap).

FLOWMAP_SKIP_FULL_SIZE=1 skips the module (iteration runs); the oracle runs in fp64 when the host has
the memory for it (>= 256 GB free), else in fp32 — the reference's own precision — and says which: every comparison emits
its record (oracle dtype, achieved errors) as a UserWarning, so that it shows in pytest's warnings summary — the driver's
log — and appends it to $FLOWMAP_PARITY_RECORD when that is set."""

import json
import os
import warnings

import pytest
import torch

from conftest import assert_close, assert_grad_close, maxerr, relerr
from helpers import mapping_cfg, run_ours, step_masks
from oracle import flowmap_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("FLOWMAP_SKIP_FULL_SIZE") == "1", reason="FLOWMAP_SKIP_FULL_SIZE=1")]
DEV = "cuda:0"
F, H, W, P = 150, 720, 1280, 1000
FOCAL = 0.8  # not the scene's 0.85: dL/dfocal is then a first-order quantity


def _host_memory_gb():
    try:
        import psutil

        return psutil.virtual_memory().available / 2**30
    except Exception:
        return 0.0


def build_reference(f, h, w, p, dev, dtype, **track_layout):
    """(scene, weight logits, tracks, oracle results): one oracle forward, one backward per loss."""
    sc = orc.synth_scene(f, h, w, seed=1, device=dev)
    tracks = orc.synth_tracks(f, h, w, scene=sc, seed=1, **track_layout)  # default: every 5th frame, +-20 frames, 35 x 35 queries
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(7))
    d = sc["depth_init"].to(dtype).requires_grad_(True)
    wp = wl.to(dtype).requires_grad_(True)
    fo = torch.tensor(FOCAL, dtype=dtype, requires_grad=True)
    fl = orc.OFlows(*(x.to(dtype) for x in (sc["flows"].forward, sc["flows"].backward, sc["flows"].forward_mask, sc["flows"].backward_mask)))
    tr = [orc.OTracks(t.xy.to(dtype), t.visibility, t.start_frame) for t in tracks]
    total, parts, out = orc.explicit_depth_step(d, wp, fo, fl, (h, w), num_points=p, tracks=tr)
    ref = {"dtype": dtype, "extrinsics": out.extrinsics.detach(), "loss_flow": parts["flow"].detach(), "loss_tracking": parts["tracking"].detach()}
    for name, last in (("flow", False), ("tracking", True)):
        g = torch.autograd.grad(parts[name], (d, wp, fo), retain_graph=not last)
        ref[name] = {"g_depth": g[0], "g_wlogit": g[1], "g_focal": g[2]}
    return sc, wl, tracks, ref


@pytest.fixture(scope="module")
def full_size():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dtype = torch.float64 if _host_memory_gb() >= 256 else torch.float32
    sc, wl, tracks, ref = build_reference(F, H, W, P, DEV, dtype)
    assert len(tracks) == 30 and tracks[0].xy.shape[2] == 1225
    return sc, wl, tracks, ref


def emit(record):
    print(record)
    warnings.warn("full-size parity record: " + json.dumps(record))  # (pytest's warnings summary: the driver's log keeps it)
    out = os.environ.get("FLOWMAP_PARITY_RECORD")  # tools/gpu_call.sh points this under gpurun_out/
    if out:
        with open(out, "a") as fh:
            fh.write(json.dumps(record) + "\n")


def check(ours, ref, grads, masks, what):
    tol = 1e-4
    record = {"case": what, "oracle_dtype": str(ref["dtype"]), "extrinsics": relerr(ours["extrinsics"], ref["extrinsics"])}
    for key in ("g_depth", "g_wlogit", "g_focal"):
        record[key] = relerr(ours[key], grads[key])
        record[key + "_max_abs_over_max_ref"] = maxerr(ours[key], grads[key])
    for name, mask in masks.items():
        record[f"g_depth[{name}]"] = relerr(ours["g_depth"][mask], grads["g_depth"][mask])
    emit(record)
    assert_close(ours["extrinsics"], ref["extrinsics"], tol, what="extrinsics")
    assert_grad_close(ours["g_depth"], grads["g_depth"], tol, masks=masks, what="g_depth")
    assert_grad_close(ours["g_wlogit"], grads["g_wlogit"], tol, what="g_wlogit")
    assert_close(ours["g_focal"], grads["g_focal"], tol, what="g_focal")


def compare_flow_only(built, hw, p, dev):
    sc, wl, _, ref = built
    ours = run_ours(sc["depth_init"], wl, FOCAL, sc["flows"], hw, p, device=dev)
    assert_close(ours["loss_flow"], ref["loss_flow"], 1e-4, what="loss_flow")
    check(ours, ref, ref["flow"], step_masks(hw, p, sc["flows"]), "C1")


def compare_flow_and_tracking(built, hw, p, dev):
    sc, wl, tracks, ref = built
    ours = run_ours(sc["depth_init"], wl, FOCAL, sc["flows"], hw, p, tracks, device=dev)
    assert_close(ours["loss_flow"], ref["loss_flow"], 1e-4, what="loss_flow")
    assert_close(ours["loss_tracking"], ref["loss_tracking"], 1e-4, what="loss_tracking")
    both = {k: ref["flow"][k] + ref["tracking"][k] for k in ("g_depth", "g_wlogit", "g_focal")}
    check(ours, ref, both, step_masks(hw, p, sc["flows"], tracks, frames=sc["depth_init"].shape[0]), "C2")


def test_c1_flow_loss_150x720x1280_vs_oracle(full_size):
    compare_flow_only(full_size, (H, W), P, DEV)


def test_c2_flow_and_tracking_150x720x1280_vs_oracle(full_size):
    compare_flow_and_tracking(full_size, (H, W), P, DEV)


# ---- BASELINE.json configs[3], configs[4] at their own frame size (row J1) ----


def _oracle_dtype():
    return torch.float64 if _host_memory_gb() >= 256 else torch.float32


def c3_case(f, h, w, points, dev, dtype, label):
    sc = orc.synth_scene(f, h, w, seed=3, device=dev)
    wl = 0.01 * torch.randn((f - 1, h, w), generator=torch.Generator().manual_seed(8))
    d = sc["depth_init"].to(dtype).requires_grad_(True)
    wp = wl.to(dtype).requires_grad_(True)
    fo = torch.tensor(FOCAL, dtype=dtype, requires_grad=True)
    fl = orc.OFlows(*(x.to(dtype) for x in (sc["flows"].forward, sc["flows"].backward, sc["flows"].forward_mask, sc["flows"].backward_mask)))
    total, parts, out = orc.explicit_depth_step(d, wp, fo, fl, (h, w), num_points=points)
    g = torch.autograd.grad(parts["flow"], (d, wp, fo))
    ref = {"dtype": dtype, "extrinsics": out.extrinsics.detach(), "loss_flow": parts["flow"].detach()}
    grads = {"g_depth": g[0], "g_wlogit": g[1], "g_focal": g[2]}
    del total, parts, out, d, wp, fl
    ours = run_ours(sc["depth_init"], wl, FOCAL, sc["flows"], (h, w), points, device=dev)
    assert_close(ours["loss_flow"], ref["loss_flow"], 1e-4, what="loss_flow")
    check(ours, ref, grads, step_masks((h, w), points, sc["flows"]), label)


def test_c3_flow_loss_65x1080x1920_vs_oracle():
    """configs[3]: 65 frames @ 1080x1920 (the video BASELINE.json shards over 4 GPUs), consistent scene, flow loss, P = 1000."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    c3_case(65, 1080, 1920, P, DEV, _oracle_dtype(), "C3 (65 x 1080x1920, scene, flow)")


def c4_shard_case(f, h, w, points, dev, dtype, label):
    from helpers import FOCAL_ULPS, run_oracle

    depth, wlogit, flows = orc.synth_iid(f, h, w, seed=4)
    ours = run_ours(depth, wlogit, 0.85, flows, (h, w), points, device=dev)
    truth = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=dtype)
    ref32 = run_oracle(depth, wlogit, 0.85, flows, (h, w), points, dtype=torch.float32) if dtype == torch.float64 else None
    masks = step_masks((h, w), points, flows)
    record = {"case": label, "oracle_dtype": str(dtype)}
    for key in ("loss_flow", "extrinsics", "g_depth", "g_wlogit"):
        record[key] = relerr(ours[key], truth[key])
        if ref32 is not None:
            record[key + "_fp32_reference_gap"] = relerr(ref32[key], truth[key])
    for name, mask in masks.items():
        record[f"g_depth[{name}]"] = relerr(ours["g_depth"][mask], truth["g_depth"][mask])
        if ref32 is not None:
            record[f"g_depth[{name}]_fp32_reference_gap"] = relerr(ref32["g_depth"][mask], truth["g_depth"][mask])
    err_focal = abs(float(ours["g_focal"]) - float(truth["g_focal"]))
    record.update({"g_focal_abs_err": err_focal, "g_focal": float(truth["g_focal"]), "g_focal_sum_of_abs_terms": truth["g_focal_terms"],
                   "g_focal_fp32_reference_abs_gap": None if ref32 is None else abs(float(ref32["g_focal"]) - float(truth["g_focal"]))})
    emit(record)
    slack = 2.0
    for key in ("loss_flow", "extrinsics"):
        assert record[key] <= 1e-4, (key, record)
    for key in ["g_depth", "g_wlogit"] + [f"g_depth[{name}]" for name in masks]:
        gap = record.get(key + "_fp32_reference_gap", 0.0)
        assert record[key] <= max(1e-4, slack * gap), (key, record[key], gap)
    assert err_focal <= max(1e-4 * abs(float(truth["g_focal"])), FOCAL_ULPS * 2.0**-24 * truth["g_focal_terms"]), record
    return record


def test_c4_shard_flow_loss_150x1080x1920_iid_vs_oracle():
    """One GPU's shard of configs[4]: 150 frames @ 1080x1920 of i.i.d. inputs (depth U(1.10,1.15), flows
    N(0,0.01²), masks U(0,1), weight logits N(0,0.01²)), flow loss, P = 1000.  Truth = the fp64 oracle; the fp32 oracle =
    what the reference's own arithmetic delivers on these inputs.  Both gaps are recorded."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    c4_shard_case(150, 1080, 1920, P, DEV, _oracle_dtype(), "C4 shard (150 x 1080x1920, i.i.d., flow)")
