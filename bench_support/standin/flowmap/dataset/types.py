"""Stand-in: the batch container the hot path reads ``videos`` from."""
from dataclasses import dataclass
from typing import Optional

from torch import Tensor


@dataclass
class Batch:
    videos: Tensor  # (batch, frame, 3, height, width)
    indices: Optional[Tensor] = None
    scenes: Optional[list] = None
    datasets: Optional[list] = None
    extrinsics: Optional[Tensor] = None  # (batch, frame, 4, 4) ground truth where a dataset has it
    intrinsics: Optional[Tensor] = None  # (batch, frame, 3, 3)
