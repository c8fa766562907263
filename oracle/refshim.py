"""Import shims that let the reference's *math* modules load under
transformers 5.x without diffusers (SURVEY.md "Oracle import recipe").

Only tests/golden/make_golden.py uses this, and only inside the build
container where /root/reference exists.  No arithmetic lives here.
"""
import dataclasses
import enum
import functools
import inspect
import sys
import types

import torch

REF_ROOT = "/root/reference"


def install():
    if "vibevoice.modular" in sys.modules:
        return
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_ROOT)
    from transformers.models.auto import auto_factory
    _orig = auto_factory._LazyAutoMapping.register
    auto_factory._LazyAutoMapping.register = (
        lambda self, k, v, exist_ok=False: _orig(self, k, v, exist_ok=True))

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    _mod("diffusers")
    cu = _mod("diffusers.configuration_utils")
    u = _mod("diffusers.utils")
    tu = _mod("diffusers.utils.torch_utils")
    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    class ConfigMixin:
        config = property(lambda self: self._internal_dict)

        def register_to_config(self, **kw):
            self.__dict__.setdefault("_internal_dict", _Cfg()).update(kw)

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **kw):
            ba = inspect.signature(init).bind(self, *a, **kw)
            ba.apply_defaults()
            ConfigMixin.register_to_config(
                self, **{k: v for k, v in ba.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return inner

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    u.deprecate = lambda *a, **k: None
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)
    su.KarrasDiffusionSchedulers = enum.Enum(
        "KarrasDiffusionSchedulers", "DPMSolverMultistepScheduler")
    su.SchedulerMixin = type("SchedulerMixin", (), {})
    su.SchedulerOutput = dataclasses.make_dataclass(
        "SchedulerOutput", [("prev_sample", torch.Tensor)])
    pkg = types.ModuleType("vibevoice.modular")
    pkg.__path__ = [REF_ROOT + "/vibevoice/modular"]
    sys.modules["vibevoice.modular"] = pkg
