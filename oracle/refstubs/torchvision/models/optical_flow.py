"""Names the reference imports at module import time (flow_predictor_raft.py:9).
Never called: the golden script feeds synthetic flows."""


class Raft_Large_Weights:
    DEFAULT = None


def raft_large(*args, **kwargs):
    raise RuntimeError("RAFT is not available in this container")
