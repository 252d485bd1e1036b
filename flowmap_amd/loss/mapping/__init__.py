"""Robust residual mappings — counterpart of the reference package flowmap/loss/mapping
(registry at flowmap/loss/mapping/__init__.py:6-16)."""

from typing import Union

from .mapping import (
    Mapping,
    MappingHuber,
    MappingHuberCfg,
    MappingL1,
    MappingL1Cfg,
    MappingL2,
    MappingL2Cfg,
    fix_aspect_ratio,
)

# cfg.name -> class; every class carries the name of the kernel variant it selects
MAPPINGS = {cls.kind: cls for cls in (MappingHuber, MappingL1, MappingL2)}
MappingCfg = Union[MappingHuberCfg, MappingL1Cfg, MappingL2Cfg]


def get_mapping(cfg) -> Mapping:
    """Instantiate the mapping a config names (raises KeyError on an unknown name)."""
    return MAPPINGS[cfg.name](cfg)


__all__ = [
    "MAPPINGS", "Mapping", "MappingCfg", "MappingHuber", "MappingHuberCfg", "MappingL1", "MappingL1Cfg", "MappingL2", "MappingL2Cfg",
    "fix_aspect_ratio", "get_mapping",
]
