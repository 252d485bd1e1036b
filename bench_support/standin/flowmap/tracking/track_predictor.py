"""Stand-in: one track segment (positions in (0, 1), visibility, first frame)."""
from dataclasses import dataclass

from torch import Tensor


@dataclass
class Tracks:
    xy: Tensor  # (batch, frame, point, 2)
    visibility: Tensor  # (batch, frame, point) bool
    start_frame: int
