"""Import-only stand-in for torchvision.transforms (absent here): the reference's datasets and image_io name its classes in annotations and
class bodies; nothing of it runs in the tests."""


class _Unavailable:
    def __init__(self, *args, **kwargs) -> None:
        pass

    def __call__(self, *args, **kwargs):
        raise RuntimeError("torchvision.transforms: import-only stand-in (oracle/refstubs)")


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return type(name, (_Unavailable,), {})
