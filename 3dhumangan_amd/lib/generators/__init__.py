from . import volume_rendering  # noqa: F401
from .map3d_generator import Map3DGenerator, SynthesisNetwork  # noqa: F401
