"""Drop-in for ``flowmap.model.procrustes`` (flowmap/model/procrustes.py)."""

from __future__ import annotations

from torch import Tensor

from .. import _ops
from .._lib import check_device


def align_rigid(p: Tensor, q: Tensor, weights: Tensor) -> Tensor:
    """flowmap/model/procrustes.py:7-51: rigid T (…,4,4) minimising Σ w‖T·p − q‖²
    (Sorkine-Hornung & Rabinovich).  p, q (*batch, P, 3), weights (*batch, P).

    Statistics are reduced on the GPU (weighted centroids with weights/(Σw+1e-8), raw-
    weight covariance), the 3×3 SVD with the reflection fix runs in registers, and the
    backward uses the polar-decomposition differential instead of svd_backward.
    """
    check_device(p, q, weights)
    batch = p.shape[:-2]
    n = p.shape[-2]
    g = 1
    for s in batch:
        g *= s
    q = q.expand(*batch, n, 3)
    weights = weights.expand(*batch, n)
    return _ops.AlignRigid.apply(p.reshape(g, n, 3), q.reshape(g, n, 3), weights.reshape(g, n)).reshape(*batch, 4, 4)
