"""Stand-in: the extrinsics registry install() replaces the "procrustes" entry of."""
from .extrinsics_procrustes import ExtrinsicsProcrustes, ExtrinsicsProcrustesCfg

EXTRINSICS = {"procrustes": ExtrinsicsProcrustes}


def get_extrinsics(cfg, num_frames):
    return EXTRINSICS[cfg.name](cfg, num_frames)
