from . import registry, logger  # noqa: F401
