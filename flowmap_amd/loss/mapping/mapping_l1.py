"""Drop-in for flowmap/loss/mapping/mapping_l1.py."""

from dataclasses import dataclass
from typing import Literal

from .mapping import Mapping


@dataclass
class MappingL1Cfg:
    name: Literal["l1"]


class MappingL1(Mapping[MappingL1Cfg]):
    """‖r‖₂ (mapping_l1.py:16-20)."""

    kind = "l1"
