"""CPU-side checks of the C-ABI library: it builds, loads, and exports what include/h3d.h declares."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build = importlib.import_module("3dhumangan_amd._build")
    path = build.build_lib()
    return ctypes.CDLL(path)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "h3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h3d_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert "h3d_ray_integrate" in names and "h3d_version" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/h3d.h but not exported by libh3d.so: {missing}"


def test_version_and_error_string(lib):
    lib.h3d_version.restype = ctypes.c_int
    assert lib.h3d_version() == 100
    lib.h3d_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.h3d_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    """Bad arguments are rejected before any HIP call."""
    lib.h3d_ray_integrate.restype = ctypes.c_int
    lib.h3d_ray_integrate.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    rc = lib.h3d_ray_integrate(None, None, None, None, None, None, 4, 8, 3, 0, 0, 0, None)
    assert rc == -1
    assert b"null pointer" in lib.h3d_last_error()


def test_product_has_no_cpu_fallback():
    import torch
    vr = importlib.import_module("3dhumangan_amd.lib.generators.volume_rendering")
    h3dlib = importlib.import_module("3dhumangan_amd._lib")
    with pytest.raises(h3dlib.H3DError):
        vr.ray_integration(torch.zeros(1, 2, 4, 5), torch.zeros(1, 2, 4, 1), clamp_mode="relu", noise_std=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "3dhumangan_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "h3d_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f
