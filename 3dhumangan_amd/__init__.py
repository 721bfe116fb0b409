"""3dhumangan_amd -- MI355X-native (gfx950) generator forward pass of 3DHumanGAN.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every hot op is a
hand-written HIP kernel in csrc/, reached through the C ABI of include/h3d.h via ctypes.

The directory name starts with a digit, so import it as::

    import importlib; h3d = importlib.import_module("3dhumangan_amd")

or simply ``import h3d`` (alias module at the repository root).  The sub-packages mirror the
reference layout: ``configs``, ``lib.generators``, ``lib.implicit_funcitions``, ``lib.components``.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def library_path():
    return _lib.LIB_PATH
