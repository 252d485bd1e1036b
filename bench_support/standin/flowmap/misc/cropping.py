"""Stand-in: the three resize / crop entry points install() rebinds (host arithmetic: the oracle's)."""
import dataclasses

from flowmap import orc  # (the oracle behind a lazy, host-only proxy: flowmap/__init__.py)


def resize_batch(batch, shape):
    return dataclasses.replace(batch, videos=orc.resize_bilinear(batch.videos, shape))


def crop_and_resize_batch_for_model(batch, cfg):
    videos, _ = orc.crop_and_resize(batch.videos, None, cfg.image_shape, cfg.patch_size)
    return dataclasses.replace(batch, videos=videos), videos.shape[-2:]


def crop_and_resize_batch_for_flow(batch, cfg):
    videos, _ = orc.crop_and_resize(batch.videos, None, cfg.image_shape, cfg.patch_size, cfg.flow_scale_multiplier)
    return dataclasses.replace(batch, videos=videos)
