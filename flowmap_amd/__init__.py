"""flowmap_amd — MI355X-native (gfx950) implementation of FlowMap's per-iteration
reprojection / flow-consistency inner loop behind the reference's Python call surface.

    flowmap_amd.model.projection   <-> flowmap/model/projection.py
    flowmap_amd.model.procrustes   <-> flowmap/model/procrustes.py
    flowmap_amd.loss               <-> flowmap/loss/**
    flowmap_amd.install()          rebinds an importable reference ``flowmap`` package
                                   to these implementations

All arithmetic of this package runs in hand-written HIP kernels (flowmap_amd/csrc) loaded
from libflowmap_hip.so; the package holds no CPU or eager implementation of it, and a missing
library or a host tensor reaching a kernel raises.  One hand-over exists and is not this
package's arithmetic: after ``install()`` a call whose tensors live on the HOST is passed to
the reference's own function or class that ``install()`` replaced (flowmap_amd/_reference.py;
SURVEY.md §8b "CPU tensors -> the reference-equivalent torch path", BASELINE.json configs[0]).
GPU tensors never reach it, and without ``install()`` there is nothing to hand over to.
"""

from . import config, flow, loss, model  # noqa: F401
from .install import install, uninstall  # noqa: F401
from .model.projection import set_lazy_surfaces  # noqa: F401
from .graph import GraphedShardedStep, GraphedStep  # noqa: F401
from .host import freeze_gc  # noqa: F401
from ._ops import release_flow_originals  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .types import BackboneOutput, Batch, Flows, ModelOutput, Tracks  # noqa: F401

__all__ = [
    "config", "loss", "model", "install", "uninstall", "set_lazy_surfaces", "FusedAdam", "GraphedStep", "GraphedShardedStep", "freeze_gc", "release_flow_originals",
    "Batch", "BackboneOutput", "Flows", "ModelOutput", "Tracks",
]
