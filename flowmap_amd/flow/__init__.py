from .flow_predictor import FlowPredictor, Flows, split_videos  # noqa: F401
