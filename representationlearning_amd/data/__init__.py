"""Device-side input pipeline."""
from .loveda import DeviceAugment, LOVEDA_MEAN, LOVEDA_STD  # noqa: F401
