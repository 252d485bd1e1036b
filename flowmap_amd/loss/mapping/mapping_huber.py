"""Drop-in for flowmap/loss/mapping/mapping_huber.py."""

from dataclasses import dataclass
from typing import Literal

from .mapping import Mapping


@dataclass
class MappingHuberCfg:
    name: Literal["huber"]
    delta: float


class MappingHuber(Mapping[MappingHuberCfg]):
    """F.huber_loss(‖r‖, 0, delta)/delta (mapping_huber.py:19-34)."""

    kind = "huber"
