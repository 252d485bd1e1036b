"""Stand-in: the weighted rigid fit (host arithmetic: the oracle's)."""
from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)


def align_rigid(points, targets, weights):
    return orc.rigid_fit(points, targets, weights)
