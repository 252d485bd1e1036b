from . import procrustes, projection  # noqa: F401
