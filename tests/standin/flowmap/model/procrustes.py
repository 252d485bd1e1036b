"""Stand-in: the weighted rigid fit (host arithmetic: the oracle's)."""
from oracle import flowmap_oracle as orc


def align_rigid(points, targets, weights):
    return orc.rigid_fit(points, targets, weights)
