from . import volume_rendering  # noqa: F401
