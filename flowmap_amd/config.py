"""Run-time options of the package in ONE explicit object (until round 5: eleven module-level variables of ``flowmap_amd._ops`` that tests
and A/B runs assigned to).

    flowmap_amd.install(options={"tap_exchange": False})           # with the rebinding
    flowmap_amd.config.configure(tap_exchange_min_bytes=0)          # stand-alone
    with flowmap_amd.config.override(unit_seed=False): ...          # for a scope (tests, A/B runs): restored on exit

``options`` is read at call time by the Python layer only (``_ops``, the loss classes, ``training``); the kernels and the C ABI have no global
state.  Unknown names raise: a misspelt switch must not silently do nothing.
"""

from __future__ import annotations

import contextlib
import dataclasses
import os
from typing import Optional


@dataclasses.dataclass
class Options:
    # Persistent dL/dweights storage (GradArena) for sparse fits with a constant index set; False = fresh zeros every step
    grad_arena: bool = True
    # Moments, finish + solve and the pose chain as ONE launch (fm_procrustes_fit_chain) instead of a memset and three kernels.
    fit_chain: bool = True
    # Dense Procrustes (`num_points: null`) backward.  False = one fused pass over the later pixels, tap gradients summed in an LDS image of the
    # earlier-frame window and flushed with atomics (1.2 ms at 150 x 720x1280 when the flow varies by a few pixels inside a 32x64 tile — camera
    # motion —, but every tap that leaves the window is a scattered atomic: 2.2 ms on rough flows, 8 ms on i.i.d. ones); True = the static tap
    # lists (built once per flow tensor, 4 B per pixel and pair) and the planned pair of kernels without atomics (2.0-2.2 ms on ANY flow,
    # dL/ddepth bit-reproducible); None (default) = decided once per flow tensor by how much the flow varies inside the tiles.
    dense_plan: Optional[bool] = None
    # (auto) the planned kernels are chosen when more than this fraction of the tiles has flows leaving the fused pass's window
    dense_plan_rough_tiles: float = 0.25
    # The packed copy costs as much HBM as the flows and masks themselves (3.3 GB at C1); False streams the caller's tensors directly
    # (tests exercise both kernels).
    packed_inputs: bool = True
    # tests: pack the first time a set of flows is seen (a one-step test then runs the packed kernel instance)
    pack_on_first_sight: bool = False
    # sample the tracking loss's tap depths from the image the flow pass leaves (while the parameter's version counter has not moved)
    tap_image: bool = True
    # The tap exchange pays where the depth images are far larger than the last-level cache (256 MB of Infinity Cache on an MI355X): at
    # 150 x 720p (553 MB) the tracking loss's taps are cold lines and the exchange takes 0.08 ms off a 1.25 ms step; at the reference's default
    # 180 x 240 (26 MB, cache-resident) there is nothing cold to avoid and its bookkeeping costs 0.05 ms.  Depth tensors below this size run
    # as in round 3.
    tap_exchange_min_bytes: int = 128 << 20
    # the tap exchange between the fused flow loss and the fused tracking loss; False: both run as in round 3
    tap_exchange: bool = True
    # The fused losses come back as RootLoss tensors (flowmap_amd/_ops.py); False: plain tensors, and a step pays autograd's ones_like fill and
    # the flow loss's is-the-seed-one launch again (two of the eight launches of a flow-only step)
    unit_seed: bool = True
    # ExtrinsicsProcrustes hands the camera-to-world chain on UNEVALUATED (LazyExtrinsics) while gradients are being recorded and nothing has asked
    # for it yet: the fused flow loss reads the fit's relative poses, and the chain is the last block's ~7 us at 149 poses with the rest of the GPU
    # idle.  Needs lazy surfaces (install()'s default); False = the fit's launch always chains the poses (rounds 1-5)
    lazy_extrinsics: bool = True
    # RootLoss.backward() runs autograd's nodes on the calling thread (no hand-over to the device's worker thread);
    # FLOWMAP_AMD_BACKWARD_THREAD=engine (or False here): autograd's default
    backward_on_calling_thread: bool = dataclasses.field(
        default_factory=lambda: os.environ.get("FLOWMAP_AMD_BACKWARD_THREAD", "caller").lower() != "engine")

    def __setattr__(self, name, value):
        if name not in _FIELDS:
            raise AttributeError(f"flowmap_amd.config: no option called {name!r} (options: {', '.join(sorted(_FIELDS))})")
        object.__setattr__(self, name, value)


_FIELDS = frozenset(f.name for f in dataclasses.fields(Options))
options = Options()  # THE instance: the Python layer reads its attributes at call time


def configure(**values) -> Options:
    """Set options for the rest of the process (what ``install(options=...)`` calls)."""
    for name, value in values.items():
        setattr(options, name, value)
    return options


@contextlib.contextmanager
def override(**values):
    """Set options for a ``with`` block; the previous values come back on exit, whatever happens inside."""
    previous = {name: getattr(options, name) for name in values}  # (an unknown name raises here, before anything is changed)
    try:
        configure(**values)
        yield options
    finally:
        for name, value in previous.items():
            setattr(options, name, value)


def defaults() -> Options:
    return Options()
