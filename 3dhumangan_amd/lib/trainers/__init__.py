from . import losses  # noqa: F401
from .d_step import discriminator_step  # noqa: F401
