"""Import-only stand-in (see ../../__init__.py): the annotation model_wrapper_overfit.py:6 imports."""
OptimizerLRScheduler = object
