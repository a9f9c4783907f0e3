"""Parameter containers for the boundary modules.

The reference's modules matter to callers in two ways only: their ``state_dict`` key names (the
on-disk format) and being ``nn.Module``s (``.parameters()``, ``.to()``,
``.eval()``).  ``ParamTree`` builds exactly that from the key table in ``statespec`` - a tree of
sub-modules whose leaves are the tensors - without any forward arithmetic: the arithmetic is in
the HIP library.  ``NativeBacked`` adds the (re)packing of those tensors into the library's
device blob whenever they change.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
from torch import nn

from ... import native, statespec
from ...init import reference_init


class ParamTree(nn.Module):
    """`spec` = (key, shape, kind) rows of `statespec`; values start from the reference's own initialisers (`init.reference_init`, drawn
    from torch's global RNG).  Tests, goldens and bench.py load `synth.synth_state_dict()` on top, explicitly."""

    def __init__(self, spec: Optional[Iterable] = None, key_prefix: str = ""):
        super().__init__()
        if spec is None:
            return
        for key, shape, kind in spec:
            value = reference_init(tuple(shape), kind)
            node = self
            parts = key.split(".")
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, ParamTree())
                node = node._modules[name]
            if kind in statespec.BUFFER_KINDS:
                node.register_buffer(parts[-1], value)
            else:
                node.register_parameter(parts[-1], nn.Parameter(value))

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("ParamTree is a parameter container; the arithmetic runs in libl2s_hip")


class NativeBacked(nn.Module):
    """Mixin: keeps a NativeModel in sync with this module's tensors (lazy, re-packed on change)."""

    _key_prefix = ""          # checkpoint prefix of this module's keys inside the library ("encoder." ...)

    def _init_native(self):
        self.__dict__["_native"] = None
        self.__dict__["_native_sig"] = None
        self.__dict__["_native_parent"] = None

    def _tensors(self):
        """The module's tensors by checkpoint key.  The dict is cached (walking ~600 state_dict entries costs ~1 ms per call - more than a
        tenth of a B=32 pass) and dropped whenever the module tree can have changed its tensor OBJECTS: `.to()/.cuda()/.float()` (`_apply`)
        and `load_state_dict`.  In-place updates keep the objects and show up in the version part of the signature."""
        cache = self.__dict__.get("_tensor_cache")
        if cache is None:
            cache = self._collect_tensors()
            self.__dict__["_tensor_cache"] = cache
        return cache

    def _collect_tensors(self):
        return {self._key_prefix + k: v for k, v in self.state_dict(keep_vars=True).items()}

    def _drop_tensor_caches(self):
        """This module's cached key -> tensor dict, its descendants' and - `encoder.cuda()` / `decoder.load_state_dict(...)` on a CHILD
        replaces tensor objects the parent `Lip2Speech` has cached too - the owning parent's."""
        for mod in self.modules():
            if isinstance(mod, NativeBacked):
                mod.__dict__["_tensor_cache"] = None
        parent = self.__dict__.get("_native_parent")
        if parent is not None:
            parent.__dict__["_tensor_cache"] = None

    def _apply(self, fn, *a, **k):
        self._drop_tensor_caches()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._drop_tensor_caches()
        return out

    def _signature(self, tensors):
        return tuple((t.data_ptr(), t._version) for t in tensors.values())

    def _same_storage(self, tensors) -> bool:
        """True when only tensor versions changed since the last pack (same device addresses): the bound pointers are still right."""
        old = self.__dict__.get("_native_sig")
        return old is not None and len(old) == len(tensors) and all(o[0] == t.data_ptr() for o, t in zip(old, tensors.values()))

    def native_model(self) -> native.NativeModel:
        parent = self.__dict__.get("_native_parent")
        if parent is not None:
            return parent.native_model()
        tensors = self._tensors()
        sig = self._signature(tensors)
        if self.__dict__["_native"] is not None and sig != self.__dict__["_native_sig"] and self.__dict__.get("_refresh_on_device") \
                and self._same_storage(tensors):
            self.__dict__["_native"].train_refresh_weights()           # in-place parameter updates (a torch optimizer stepped)
            self.__dict__["_native_sig"] = sig
        if self.__dict__["_native"] is None or sig != self.__dict__["_native_sig"]:
            if not next(iter(tensors.values())).is_cuda:
                raise RuntimeError("the Lip2Speech hot path runs on the GPU: call .to('cuda') first (no CPU fallback)")
            nm = self.__dict__["_native"] or native.NativeModel()       # re-pack into the same handle: training bindings stay valid
            nm.load(tensors, list(tensors.keys()))
            self.__dict__["_native"] = nm
            self.__dict__["_native_sig"] = sig
        return self.__dict__["_native"]

    def mark_weights_changed(self):
        """Call after updating parameters behind autograd's back - the fused optimizer writes the flat buffer directly, and writes through
        `p.data` (`p.data.clamp_()`, `p.data.copy_()`) bump no version counter, so `native_model()` cannot see them: the packed
        blob is rebuilt - on the device when the training state is set up (l2s_train_refresh_weights), else from the host on next use."""
        target = self.__dict__.get("_native_parent") or self
        if target.__dict__.get("_refresh_on_device") and target.__dict__.get("_native") is not None:
            target.__dict__["_native"].train_refresh_weights()
            target.__dict__["_native_sig"] = target._signature(target._tensors())
        else:
            target.__dict__["_native_sig"] = None
