"""Host-side helper for launch-bound optimisation loops."""

from __future__ import annotations

import gc


def freeze_gc() -> int:
    """Collect once, then move everything alive now to the permanent generation: a full pass of the cyclic collector
    over torch's import-time objects costs ≈50 ms, i.e. ≈40 steps.  Returns the number of objects frozen."""
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()
