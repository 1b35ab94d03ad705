"""Minimal stand-in for the un-vendored `ever` package (BUILD CONTAINER ONLY).

Only what the RSSFormer model import chain touches: ERModule,
registry.MODEL, logger.get_logger.  Written for this project; used solely by
oracle/make_golden.py to import /root/reference and emit golden vectors.
"""
from .interface import ERModule  # noqa: F401
from .core import registry  # noqa: F401
from .interface import MultiTransform  # noqa: F401  (er.MultiTransform, module/tta.py:13)
