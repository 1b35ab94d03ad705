from . import registry  # noqa: F401
from .config import AttrDict, ConfigModule  # noqa: F401
