"""Flow post-processing side of the hot path (flowmap/flow/): the producer of ``Flows``."""

from .flow_predictor import FlowPredictor, Flows, split_videos  # noqa: F401
