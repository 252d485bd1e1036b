"""Stand-in: the loss registry and its factory."""
from .loss import Loss
from .loss_flow import LossFlow, LossFlowCfg
from .loss_tracking import LossTracking, LossTrackingCfg

LOSSES = {"flow": LossFlow, "tracking": LossTracking}


def get_losses(cfgs):
    return [LOSSES[cfg.name](cfg) for cfg in cfgs]
