"""Host (CPU) tensors after ``flowmap_amd.install()`` ("dtype / device": "CPU tensors -> fall back to the
reference-equivalent torch path (needed for C0)").

The kernels of this package run on the GPU only; there is no CPU implementation in it.  But when the REFERENCE is importable —
that is what ``install()`` patches — its own functions and classes are right there: ``install()`` records the originals it
replaces (``twins``), and every entry point of this package that the reference reaches through its registries or import sites
hands a call whose tensors live on the host to its original (``flowmap.overfit`` with ``device = "cpu"``, BASELINE.json
configs[0]: "plumbing, no GPU").  Nothing here touches ``oracle/``.  Without ``install()`` (stand-alone use of this package) no
twin is known and host tensors raise as before (``_lib.check_device``); with the tests' host double injected
(``_lib.set_library_for_testing``) host tensors ARE the double's input and nothing is handed back.
"""

from __future__ import annotations

import dataclasses
from typing import Optional

import torch

from . import _lib

# name -> the reference's original: functions of flowmap.model.projection / procrustes, and classes by their reference name
twins: dict = {}
counters = {"host_calls": 0}


def _first_tensor(obj, depth: int = 0) -> Optional[torch.Tensor]:
    if torch.is_tensor(obj):
        return obj
    lazy = getattr(type(obj), "_fm_lazy", None)  # LazyWeights / LazyExtrinsics: any other attribute read would EVALUATE what they stand for
    if lazy:
        return obj.__dict__[lazy]
    depths = getattr(obj, "depths", None)  # LazySurfaces, BackboneOutput, ModelOutput
    if torch.is_tensor(depths):
        return depths
    videos = getattr(obj, "videos", None)  # Batch
    if torch.is_tensor(videos):
        return videos
    if depth >= 2:
        return None
    if isinstance(obj, (list, tuple)):
        for x in obj:
            t = _first_tensor(x, depth + 1)
            if t is not None:
                return t
    elif dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        for f in dataclasses.fields(obj):
            t = _first_tensor(getattr(obj, f.name, None), depth + 1)
            if t is not None:
                return t
    return None


_Tensor = torch.Tensor


def on_host(*objs) -> bool:
    """Do the tensors of this call live on the host while the C ABI is the HIP library?  (Called several times per optimisation step by
    every rebound name: the common cases — a tensor, a Batch, a LazySurfaces as the first argument — are decided without recursion.)"""
    if _lib._lib_is_test_double:
        return False
    for obj in objs:
        if isinstance(obj, _Tensor):
            return obj.device.type == "cpu"
        lazy = getattr(type(obj), "_fm_lazy", None)  # LazyWeights / LazyExtrinsics (see _first_tensor)
        if lazy:
            return obj.__dict__[lazy].device.type == "cpu"
        t = getattr(obj, "depths", None)  # LazySurfaces, BackboneOutput, ModelOutput
        if not isinstance(t, _Tensor):
            t = getattr(obj, "videos", None)  # Batch
        if not isinstance(t, _Tensor):
            t = _first_tensor(obj)
        if t is not None:
            return t.device.type == "cpu"
    return False


def host_twin(name: str, *objs):
    """The reference's original for ``name`` when this call must go to it, else None."""
    if not twins:  # nothing installed: stand-alone use of this package
        return None
    ref = twins.get(name)
    if ref is None or not on_host(*objs):
        return None
    counters["host_calls"] += 1
    return ref


def dispatching(name: str, ours, ref):
    """``ours`` for device tensors, the reference's own ``ref`` for host tensors (what install() binds at the reference's import sites)."""
    import functools

    twins[name] = ref

    @functools.wraps(ours)
    def call(*args, **kwargs):
        if (on_host(*args, *kwargs.values()) if kwargs else on_host(*args)):
            counters["host_calls"] += 1
            return ref(*args, **kwargs)
        return ours(*args, **kwargs)

    call._fm_ours, call._fm_reference = ours, ref
    return call
