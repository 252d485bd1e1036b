"""Stand-in: the flow container and the predictor base class whose two post-processing methods install() rebinds."""
from dataclasses import dataclass

from torch import Tensor, nn

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)


@dataclass
class Flows:
    forward: Tensor  # (batch, pair, height, width, 2)
    backward: Tensor
    forward_mask: Tensor  # (batch, pair, height, width)
    backward_mask: Tensor


class FlowPredictor(nn.Module):
    def forward(self, videos: Tensor) -> Tensor:  # concrete predictors return (batch, frame - 1, height, width, 2)
        raise NotImplementedError

    @staticmethod
    def compute_consistency_mask(videos: Tensor, flow: Tensor) -> Tensor:
        return orc.consistency_mask(videos, flow)

    def compute_bidirectional_flow(self, batch, flow_shape) -> Flows:
        o = orc.bidirectional_flows(batch.videos, self.forward, flow_shape)
        return Flows(o.forward, o.backward, o.forward_mask, o.backward_mask)
