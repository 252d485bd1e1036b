"""ctypes binding of libflowmap_hip.so (the C ABI declared in include/flowmap_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``flowmap_amd.build``.
There is NO fallback inside this package: if the shared object is missing or a call reports
a non-zero status, a RuntimeError is raised, and a host tensor reaching a kernel is refused
(``check_device``).  (After ``flowmap_amd.install()`` the entry points the reference reaches
hand HOST-tensor calls back to the reference's own functions before they get here —
flowmap_amd/_reference.py; that is the host application's code, not a second implementation.)
``set_library_for_testing`` exists so the CPU test suite can inject tests/host_sim's serial
build of the same math; the product never selects it on its own.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path
from typing import Optional

import torch  # noqa: F401  (imported first so our .so binds to torch's libamdhip64)

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libflowmap_hip.so"

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float
D = ctypes.c_double

# name -> argtypes, in the order of include/flowmap_hip.h
SIGNATURES = {
    "fm_flow_loss_fused": [P] * 11 + [I, I, I, I, I, F, F, F, P, P, I, P],
    "fm_flow_loss_fused_adam": [P] * 11 + [I, I, I, I, I, F, F, F, P, P, I, P, P, P, L, D, D, D, D, P],
    "fm_flow_pack_inputs": [P, P, P, P, I, I, I, I, P, P],
    "fm_flow_loss_fused_taps": [P] * 11 + [I, I, I, I, I, F, F, F, P, P, I, P, P, P, P, L, D, D, D, D, P],
    "fm_flow_loss_finalize": [P] * 6 + [I, I, F, F] + [P] * 4 + [P],
    "fm_flow_valid_norm": [P, P, L, F, P, P, P],
    "fm_scale_if_needed": [P, L, P, L, P, P, P],
    "fm_abi_version": [],
    "fm_halo_copy": [P, L, I, P, P, P],
    "fm_flow_ghost_terms": [P] * 14 + [I, I, I, F, F, F, P],
    "fm_halo_delta": [P, L, I, P, P, L, P, P, P, L, P, P],
    "fm_halo_ghost_begin": [P, L, I, P, L, P, P, L, P, P, P, I, P, P, P, P],
    "fm_halo_delta_sparse": [P, L, I, P, P, L, P, P, P, L, P, P],
    "fm_halo_add": [P, L, I, P, P, P],
    "fm_halo_scatter": [P, L, I, P, P, L, P, P, L, P],
    "fm_flow_loss_fused_views": [P] * 11 + [I, I, I, I, I, F, F, F, P, P, I, P, P],
    "fm_flow_valid_norm_views": [P, P, I, I, L, F, P, P, P, P],
    "fm_flow_pack_inputs_views": [P, P, P, P, I, I, I, I, P, P, P],
    "fm_procrustes_fit_views": [P] * 5 + [F, P, L, I, I, I, I, P, P, P, P, P, P],
    "fm_procrustes_fit_chain_views": [P] * 5 + [F, P, L, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "fm_procrustes_scatter_views": [P] * 5 + [F, P, L, I, I, I, I] + [P] * 7 + [P, P],
    "fm_procrustes_scatter_plan_views": [P, P, L, I, I, I, I, P, P, P, P],
    "fm_softmin_blend_fwd": [P, P, I, I, I, P, P, P, P],
    "fm_softmin_blend_bwd": [P, P, P, I, I, I, P, P],
    "fm_softmin_score_fwd": [P, P, F, P, P, L, P, P, P, I, I, I, I, P, P],
    "fm_softmin_score_bwd": [P, P, F, P, P, L, P, P, P, I, I, I, I, P, P, P, P, P, P],
    "fm_random_subset": [ctypes.c_ulonglong, L, L, P, P],
    "fm_random_subset_stateful": [P, L, L, P, P],
    "fm_world_points": [P, P, P, P, I, I, I, P, P, P],
    "fm_consistency_mask": [P, P, I, I, I, I, P, P],
    "fm_flow_postprocess": [P, P, I, I, I, I, I, I, I, P, P, P],
    "fm_resize_crop": [P, L, I, I, I, I, I, I, I, I, P, P],
    "fm_fill_zero": [P, L, I, P],
    "fm_adam_step": [P, P, P, P, L, L, D, D, D, D, D, P],
    "fm_adam_step_elements": [P, P, P, P, P, L, L, D, D, D, D, D, P],
    "fm_adam_step_capturable": [P, P, P, P, L, P, D, D, D, D, D, P],
    "fm_procrustes_stats": [P] * 5 + [F, P, L, I, I, I, I, I, P, P],
    "fm_procrustes_fit": [P] * 5 + [F, P, L, I, I, I, I, I, P, P, P, P, P],
    "fm_procrustes_fit_chain": [P] * 5 + [F, P, L, I, I, I, I, P, P, P, P, P, P, P, P],
    "fm_pose_solve": [P, I, P, P, P, P],
    "fm_pose_solve_bwd": [P, P, P, P, I, P, P, L, P],
    "fm_procrustes_scatter": [P] * 5 + [F, P, L, I, I, I, I, I] + [P] * 8 + [P],
    "fm_procrustes_dense_tiles": [I, I, P],
    "fm_procrustes_dense_plan": [P, I, I, I, I, P, P, P, P],
    "fm_procrustes_scatter_dense": [P, P, P, P, F, I, I, I, I, P, P, P, P, P, P, P, P],
    "fm_pose_solve_bwd_kinv": [P, P, P, P, P, I, I, P, P, P],
    "fm_sparse_store": [P, P, L, I, L, P, P],
    "fm_procrustes_scatter_plan": [P, P, L, I, I, I, I, P, P, P],
    "fm_pose_chain_fwd": [P, I, I, P, P],
    "fm_pose_chain_bwd": [P, P, P, I, I, P, P],
    "fm_relative_pose_fwd": [P, I, I, P, P, P],
    "fm_relative_pose_bwd": [P, P, P, I, I, P, P],
    "fm_allpairs_pose_fwd": [P, I, I, P, P],
    "fm_allpairs_pose_bwd": [P, P, I, I, P, P],
    "fm_focal_intrinsics_fwd": [P, L, L, I, I, P, P, P],
    "fm_focal_intrinsics_bwd": [P, L, L, I, I, P, P],
    "fm_intrinsics_inverse": [P, I, P, P],
    "fm_intrinsics_inverse_bwd": [P, P, I, P, I, P],
    "fm_unproject_fwd": [P, L, P, P, I, L, P, P],
    "fm_unproject_bwd": [P, L, P, P, P, I, L, P, P, P],
    "fm_reproject_fwd": [P, P, P, I, L, P, P],
    "fm_reproject_bwd": [P, P, P, P, I, L, P, P, P, P, P],
    "fm_bilinear_sample_fwd": [P, P, I, I, I, I, L, P, P],
    "fm_bilinear_sample_bwd": [P, P, I, I, I, I, L, P, P],
    "fm_mapping_fwd": [P, P, L, I, F, F, F, P, P],
    "fm_mapping_bwd": [P, P, P, L, I, F, F, F, P, P, P],
    "fm_align_rigid_stats": [P, P, P, I, L, P, P],
    "fm_align_rigid_bwd": [P, P, P, I, L, P, P, P, P, P, P],
    "fm_extrinsics_inverse": [P, I, P, P],
    "fm_track_points": [P, I] + [P] * 4 + [I] + [P] * 4 + [I, I, I, I, P, P, P, P],
    "fm_track_loss_fwd": [P] * 6 + [I, I, I, P, P, I, I, I, I, F, F, F, F] + [P] * 7 + [P],
    "fm_track_loss_fused_fwd": [P, I, I, I, P, P, P, P, I, P, P, P, P, I, I, I, I, I, I, F, F, F, F] + [P] * 10 + [P],
    "fm_track_loss_fused_fwd_taps": [P, P, P, P, P, I, P, P, P, P, I, I, I, I, I, I, F, F, F, F] + [P] * 10 + [P] * 6 + [L, P, L, P, P],
    "fm_tap_grad_apply": [P, P, L, P, P, P, P, P, P],
    "fm_track_loss_bwd": [P] * 7 + [I, P, P, P],
    "fm_track_scatter": [P] * 6 + [I, I, P, P, P, I, I, I, P, P],
    "fm_track_scatter_plan": [P, P, P, P, I, I, I, I, P, P, P],
    "fm_depth_gather": [P, P, P, P, P, L, P, P, P, I, I, L, P, P],
    "fm_depth_gather_kgrad": [P, P, P, P, P, L, P, I, I, P, P, I, P, I, P],
    "fm_procrustes_bwd_planned": [P, P, F, L, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, P],
}

_lib: Optional[ctypes.CDLL] = None
_lib_is_test_double = False


def _bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue  # checked by tests/test_abi.py; calling a missing symbol raises below
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    return lib


def library() -> ctypes.CDLL:
    """Return the loaded native library, loading libflowmap_hip.so on first use."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            try:  # a fresh checkout: compile the kernels in-tree (hipcc cross-compiles gfx950 anywhere)
                from .build import build_library

                build_library(verbose=False)
            except Exception as exc:
                raise RuntimeError(
                    f"{LIB_PATH} is missing and could not be built ({exc}); run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                    "flowmap_amd has no CPU or eager fallback."
                ) from exc
        _lib = _bind(ctypes.CDLL(str(LIB_PATH)))
    return _lib


TORCH_LIB_PATH = _HERE / "libflowmap_torch.so"
_torch_ops = None


def torch_ops():
    """``torch.ops.flowmap_amd``: the C++ operators of libflowmap_torch.so (csrc/fm_torch.cpp), bound to the
    same build of the C ABI as ``library()``.  Loaded on first use; built in-tree if missing."""
    global _torch_ops
    if _torch_ops is None:
        if not TORCH_LIB_PATH.exists():
            try:
                from .build import build_torch_binding

                build_torch_binding(verbose=False)
            except Exception as exc:
                raise RuntimeError(f"{TORCH_LIB_PATH} is missing and could not be built ({exc}); run "
                                   "`python -c 'import __graft_entry__ as g; g.build()'`.  flowmap_amd has no CPU or eager fallback.") from exc
        torch.ops.load_library(str(TORCH_LIB_PATH))
        library()  # the C ABI itself (builds it if need be)
        torch.ops.flowmap_amd.set_library(str(_lib._name), _lib_is_test_double)
        _torch_ops = torch.ops.flowmap_amd
    return _torch_ops


def set_library_for_testing(path: Optional[os.PathLike]) -> None:
    """Inject a different build of the same C ABI (tests only), or reset with None."""
    global _lib, _lib_is_test_double
    if path is None:
        _lib, _lib_is_test_double = None, False
    else:
        _lib, _lib_is_test_double = _bind(ctypes.CDLL(str(path))), True
    if _torch_ops is not None:  # keep the C++ operators on the same library
        library()
        _torch_ops.set_library(str(_lib._name), _lib_is_test_double)


def using_test_double() -> bool:
    return _lib_is_test_double


def ptr(t) -> Optional[int]:
    """Device (or host, for the test double) address of a tensor; None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


def stream_for(t: torch.Tensor) -> Optional[int]:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def check_device(*tensors) -> torch.device:
    """All tensors must live on one device; CUDA(HIP) unless the test double is active."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"flowmap_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is None:
        raise RuntimeError("flowmap_amd: no tensor arguments")
    if dev.type != "cuda" and not _lib_is_test_double:
        raise RuntimeError(
            f"flowmap_amd: tensors are on {dev}; the HIP path needs a GPU (device 'cuda' on ROCm). "
            "There is no CPU fallback."
        )
    if dev.type == "cuda" and _lib_is_test_double:
        raise RuntimeError("flowmap_amd: the host test double cannot take GPU tensors")
    return dev


_STATUS = {1: "invalid argument", 2: "HIP launch/runtime failure"}


def call(name: str, *args) -> None:
    lib = library()
    fn = getattr(lib, name, None)
    if fn is None:
        raise RuntimeError(f"flowmap_amd: native symbol {name} is not exported by the loaded library")
    status = fn(*args)
    if status != 0:
        raise RuntimeError(f"flowmap_amd: {name} failed: {_STATUS.get(status, status)}")
