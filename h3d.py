"""Alias: ``import h3d`` == the package in ./3dhumangan_amd (whose name is not a Python identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("3dhumangan_amd")
