"""Annotation-only stand-in for `jaxtyping` (absent from this image).

Used ONLY by oracle/make_golden.py to import the reference package in the build
container.  It carries no arithmetic: every annotation type is a subscriptable no-op
and the import hook is a null context manager.
"""
import contextlib


class _Anno:
    def __class_getitem__(cls, item):
        return cls


class Float(_Anno):
    pass


class Bool(_Anno):
    pass


class Int64(_Anno):
    pass


class Int32(_Anno):
    pass


class Int(_Anno):
    pass


class UInt8(_Anno):
    pass


class Shaped(_Anno):
    pass


def jaxtyped(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


@contextlib.contextmanager
def install_import_hook(*args, **kwargs):
    yield
