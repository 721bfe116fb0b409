from . import losses  # noqa: F401
from .d_step import discriminator_step  # noqa: F401
from .g_step import generator_param_groups, generator_step, make_generator_optimizer  # noqa: F401
from .iteration import adversarial_iteration  # noqa: F401
