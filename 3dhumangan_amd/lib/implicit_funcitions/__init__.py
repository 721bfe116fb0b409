from .modulated import COORDCONCATSIREN  # noqa: F401
