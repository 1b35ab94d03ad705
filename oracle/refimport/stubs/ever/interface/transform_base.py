from . import Transform  # noqa: F401  (ever.interface.transform_base.Transform, imported by the reference's module/tta.py:49)
