"""Stand-in: the robust-mapping registry."""
from dataclasses import dataclass

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)


@dataclass
class MappingHuberCfg:
    name: str
    delta: float


@dataclass
class MappingL1Cfg:
    name: str


@dataclass
class MappingL2Cfg:
    name: str


class Mapping:
    def __init__(self, cfg):
        self.cfg = cfg

    def forward(self, a, b, image_shape):
        return orc.robust(a, b, image_shape, self.cfg.name, getattr(self.cfg, "delta", 0.01))


class MappingHuber(Mapping):
    pass


class MappingL1(Mapping):
    pass


class MappingL2(Mapping):
    pass


MAPPINGS = {"huber": MappingHuber, "l1": MappingL1, "l2": MappingL2}


def get_mapping(cfg):
    return MAPPINGS[cfg.name](cfg)
