"""Drop-in for flowmap/loss/mapping/mapping_l2.py."""

from dataclasses import dataclass
from typing import Literal

from .mapping import Mapping


@dataclass
class MappingL2Cfg:
    name: Literal["l2"]


class MappingL2(Mapping[MappingL2Cfg]):
    """½‖r‖² (mapping_l2.py:16-21)."""

    kind = "l2"
