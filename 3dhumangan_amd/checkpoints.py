"""Reading the checkpoint files the reference's trainer writes (lib/trainers/base_trainer.py:183-202):

    <step>_generator.pth / _discriminator.pth   torch.save(module)   -- a PICKLED nn.Module of the reference's classes
    <step>_ema.pth                              torch.save(ema)      -- a pickled ExponentialMovingAverage object
    <step>_optimizer_{G,D}.pth, _scaler.pth     plain state dicts
    *_state_dict.pth                            plain state dicts (what the released checkpoint ships)

The pickled objects name classes under the reference's import paths (`lib.generators.map3d_generator.Map3DGenerator`,
`lib.components.ema.ExponentialMovingAverage`, ...), which do not exist here.  They are restored WITHOUT the reference on
the path: an unpickler maps every class under `lib.` / `configs` to a stand-in built on the fly (an nn.Module subclass when
the pickled state has module dictionaries), which is all `state_dict()` needs.  Every other global a pickle names must be
on an EXACT (module, name) allow-list (torch's tensor / storage / parameter rebuild helpers, the spectral-norm hook
classes, `collections.OrderedDict`, numpy's array reconstruction) or resolve to a torch dtype / storage class or to an
nn.Module class under `torch.nn.modules`; dotted names (`torch` + `os.system`), `builtins.getattr`, `functools.partial`,
`copyreg._reconstructor` and everything else raise `UnpicklingError`.  This narrows what a file can name to constructors of
data containers; it is a hardening of a pickle loader, not a sandbox -- prefer the `*_state_dict.pth` files (which go
through torch's `weights_only=True` loader and never reach this unpickler) for files of unknown origin.
"""
import io
import pickle

import torch
import torch.nn as nn

from .lib.components.ema import ExponentialMovingAverage

# Globals a trainer checkpoint legitimately names, besides the reference's own classes (which become stand-ins).  The list is
# EXACT (module, name) pairs plus two type-checked families; no module or prefix is allowed wholesale: pickle's find_class
# resolves dotted names through attributes (`torch` + `os.system` is `torch.os.system`), so "anything in torch" is "anything".
_ALLOWED_EXACT = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "slice"), ("builtins", "complex"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("_codecs", "encode"),                                                        # bytes objects in protocol 2
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
    ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    # tensor / parameter reconstruction
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor_v3"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch._utils", "_rebuild_qtensor"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch.nn.parameter", "Buffer"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.storage", "UntypedStorage"),
    ("torch.storage", "TypedStorage"), ("torch.serialization", "_get_layout"),
    # what a pickled module of the reference drags along: hooks of spectral_norm and the hook wrapper of nn.Module
    ("torch.nn.modules.module", "_WrappedHook"),
    ("torch.nn.utils.spectral_norm", "SpectralNorm"), ("torch.nn.utils.spectral_norm", "SpectralNormStateDictHook"),
    ("torch.nn.utils.spectral_norm", "SpectralNormLoadStateDictPreHook"),
}


def _allowed(module, name, resolve):
    """`resolve()` performs the lookup; it is only called for candidates of the two type-checked families, and the object
    it returns must BE a dtype / storage class (module `torch`) or an nn.Module class (module `torch.nn.modules.*`)."""
    if module == "__builtin__":                      # protocol-2 spelling, which pickle itself maps to builtins
        module = "builtins"
    if "." in name or not name.isidentifier():       # dotted names walk attributes: never
        return False
    if (module, name) in _ALLOWED_EXACT:
        return True
    if module == "torch":                            # torch.float32, torch.FloatStorage, ...
        obj = resolve()
        return isinstance(obj, torch.dtype) or (isinstance(obj, type) and name.endswith("Storage") and
                                                 issubclass(obj, (torch.storage._LegacyStorage, torch.storage.TypedStorage,
                                                                  torch.storage.UntypedStorage)))
    if module.startswith("torch.nn.modules."):       # Linear, Conv2d, SyncBatchNorm, Sequential, ...
        obj = resolve()
        return isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__.startswith("torch.nn.modules.")
    return False


class _Bag:
    """Stand-in for a non-module reference class: attribute bag."""


class _StubModule(nn.Module):
    """Stand-in for any reference nn.Module class: carries _parameters / _buffers / _modules, never runs forward."""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise RuntimeError("stand-in for a reference module: only its state_dict() is meaningful")


_module_stubs = {}


def _stub_for(module, name):
    if module.startswith("lib.components.ema") and name == "ExponentialMovingAverage":
        return ExponentialMovingAverage
    key = (module, name)
    if key not in _module_stubs:
        # nn.Module subclass: pickle restores __dict__ through nn.Module.__setstate__, plain objects through __dict__.update;
        # a Module stand-in works for both kinds because an attribute bag is all a non-module needs
        _module_stubs[key] = type(name, (_StubModule,), {"__module__": "3dhumangan_amd.checkpoints", "_ref_path": f"{module}.{name}"})
    return _module_stubs[key]


class _RefUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "lib" or module.startswith("lib.") or module == "configs" or module.startswith("configs."):
            return _stub_for(module, name)
        if not _allowed(module, name, lambda: pickle.Unpickler.find_class(self, module, name)):
            raise pickle.UnpicklingError(f"checkpoint names the global {module}.{name}, which is not on the allow-list of "
                                         "3dhumangan_amd.checkpoints (only tensors, torch.nn classes and plain containers load)")
        return super().find_class(module, name)


class _RefPickleModule:
    """The `pickle_module` torch.load expects: same surface as pickle, with our Unpickler."""
    __name__ = "pickle"
    Unpickler = _RefUnpickler
    load = staticmethod(lambda f, **kw: _RefUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _RefUnpickler(io.BytesIO(b), **kw).load())
    dumps, dump, HIGHEST_PROTOCOL, PickleError, UnpicklingError = (pickle.dumps, pickle.dump, pickle.HIGHEST_PROTOCOL,
                                                                   pickle.PickleError, pickle.UnpicklingError)


def load_reference_pickle(path, map_location="cpu"):
    """torch.load of a file written by the reference's trainer, reference classes replaced by stand-ins.  Plain state-dict
    files load through torch's own restricted loader; only files that hold pickled objects reach the allow-listing unpickler."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, TypeError):
        pass
    return torch.load(path, map_location=map_location, pickle_module=_RefPickleModule, weights_only=False)


def state_dict_of(obj):
    """state dict from whatever a checkpoint file holds: a dict of tensors, or a (stand-in) module."""
    if isinstance(obj, dict):
        return obj
    if isinstance(obj, nn.Module):
        return obj.state_dict()
    raise TypeError(f"cannot take a state dict from {type(obj).__name__}")


def load_generator(generator, path, ema_path=None, map_location="cpu", strict=True):
    """Load generator weights from any of the reference's formats into `generator` (this build's Map3DGenerator):
    `path` a state-dict file or a pickled module; `ema_path` (optional) a pickled ExponentialMovingAverage whose shadow
    parameters are then copied over the trainable parameters -- what the reference's apps evaluate with."""
    generator.load_state_dict(state_dict_of(load_reference_pickle(path, map_location)), strict=strict)
    if ema_path is not None:
        ema = load_reference_pickle(ema_path, map_location)
        if not isinstance(ema, ExponentialMovingAverage):
            raise TypeError(f"{ema_path} does not hold an ExponentialMovingAverage")
        live = [p for p in generator.parameters() if p.requires_grad]
        if len(live) != len(ema.shadow_params) or any(p.shape != s.shape for p, s in zip(live, ema.shadow_params)):
            raise ValueError("EMA shadow list does not match the generator's parameters (count / shapes / order)")
        ema.copy_to(generator.parameters())
    return generator
