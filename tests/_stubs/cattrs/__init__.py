"""TEST INFRASTRUCTURE ONLY - a stand-in for the third-party ``cattrs`` package, which is not installed in this image.

``/root/reference/baybe`` imports ``cattrs`` at module level for (de)serialisation (``serialization/core.py:19-39``) and for a
handful of attribute converters (``cattrs.structure(x, tuple[float, ...])``: ``parameters/numerical.py:34``,
``constraints/discrete.py:104``, ``objectives/desirability.py:92``, ``utils/basic.py:347``).  The tests that drive the
reference's OWN ``Campaign`` / ``SearchSpace`` / ``BayesianRecommender`` / ``simulate_*`` code over the HIP plug-in need neither
JSON round trips nor cattrs' validation machinery - only those converters and the decorator-style registration calls executed at
import time.  This package provides exactly that: ``structure`` for builtin / tuple / list / union / optional targets, a
``Converter`` whose ``register_*`` methods record and return their argument, and the exception / helper names the reference
imports.  ``Converter.structure`` / ``unstructure`` of attrs classes are NOT provided (``to_json`` / ``from_json`` raise).

Nothing under ``baybe_amd/`` imports this (``tests/test_lib_cpu.py`` checks); ``tests/conftest.py`` puts the directory on
``sys.path`` only when a real ``cattrs`` is not importable.
"""

from __future__ import annotations

import types
import typing
from typing import Any

__all__ = ["structure", "unstructure", "Converter", "override", "BaseValidationError", "ClassValidationError",
           "IterableValidationError", "StructureHandlerNotFoundError", "ForbiddenExtraKeysError"]


class BaseValidationError(Exception):
    def __init__(self, message="", exceptions=(), cl=None):
        super().__init__(message)
        self.message, self.exceptions, self.cl = message, tuple(exceptions), cl


class IterableValidationError(BaseValidationError):
    pass


class ClassValidationError(BaseValidationError):
    pass


class StructureHandlerNotFoundError(Exception):
    def __init__(self, message="", type_=None):
        super().__init__(message)
        self.type_ = type_


class ForbiddenExtraKeysError(Exception):
    pass


def _structure(obj: Any, tp: Any, hooks: dict | None = None) -> Any:
    hooks = hooks or {}
    if tp in hooks:
        return hooks[tp](obj, tp)
    if tp is Any:
        return obj
    origin = typing.get_origin(tp)
    args = typing.get_args(tp)
    if origin in (typing.Union, types.UnionType):
        if obj is None and type(None) in args:
            return None
        for a in args:  # pass-through of values that already have one of the member types, else the first that structures
            if isinstance(a, type) and isinstance(obj, a) and not (a in (int, float) and isinstance(obj, bool)):
                return obj
        last = None
        for a in args:
            if a is type(None):
                continue
            try:
                return _structure(obj, a, hooks)
            except Exception as ex:  # noqa: BLE001
                last = ex
        raise last if last is not None else StructureHandlerNotFoundError(f"cannot structure {obj!r} as {tp}", tp)
    if origin in (tuple, list, set, frozenset) or tp in (tuple, list, set, frozenset):
        kind = origin or tp
        if isinstance(obj, (str, bytes)) or not hasattr(obj, "__iter__"):
            raise IterableValidationError(f"While structuring {tp}", [TypeError(f"{obj!r} is not iterable")], tp)
        items = list(obj)
        if kind is tuple and args and not (len(args) == 2 and args[1] is Ellipsis):
            if len(args) != len(items):
                raise IterableValidationError(f"While structuring {tp}", [ValueError("wrong tuple length")], tp)
            elem_types = list(args)
        else:
            elem_types = [args[0] if args else Any] * len(items)
        out, errors = [], []
        for it, et in zip(items, elem_types):
            try:
                out.append(_structure(it, et, hooks))
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)
        if errors:
            raise IterableValidationError(f"While structuring {tp}", errors, tp)
        return kind(out)
    if origin is dict or tp is dict:
        kt, vt = args if args else (Any, Any)
        return {_structure(k, kt, hooks): _structure(v, vt, hooks) for k, v in dict(obj).items()}
    if tp in (int, float, str, bool, bytes, complex):
        if tp is float and isinstance(obj, str):
            return float(obj)
        return tp(obj)
    if typing.is_typeddict(tp):
        hints = typing.get_type_hints(tp)
        errors = []
        extra = set(obj) - set(hints)
        if extra:
            errors.append(ForbiddenExtraKeysError(f"extra keys {sorted(extra)}"))
        out = {}
        for k, v in obj.items():
            if k in hints:
                try:
                    out[k] = _structure(v, hints[k], hooks)
                except Exception as ex:  # noqa: BLE001
                    errors.append(ex)
        if errors:
            raise ClassValidationError(f"While structuring {tp.__name__}", errors, tp)
        return out
    if isinstance(tp, type) and isinstance(obj, tp):
        return obj
    if isinstance(tp, type) and isinstance(obj, dict) and "type" in obj:
        # the reference's type-tagged dictionaries ({"type": "qLogEI", ...}: serialization/core.py:109-146 look the
        # subclass up by class name or ``abbreviation`` and build it from the remaining keys)
        def walk(c):
            for sub in c.__subclasses__():
                yield sub
                yield from walk(sub)

        rest = {k: v for k, v in obj.items() if k != "type"}
        for sub in [tp, *walk(tp)]:
            if obj["type"] in (sub.__name__, getattr(sub, "abbreviation", None)):
                return sub(**rest)
    raise StructureHandlerNotFoundError(f"the cattrs stand-in of the test suite cannot structure {obj!r} as {tp}", tp)


def structure(obj: Any, cl: Any) -> Any:
    return _structure(obj, cl)


def unstructure(obj: Any, unstructure_as: Any = None) -> Any:
    raise NotImplementedError("serialisation is outside the cattrs stand-in of the test suite")


class _Override:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def override(**kw):
    return _Override(**kw)


class Converter:
    """Registration calls are recorded (and decorators hand their function back); structuring works for the plain types above
    with the registered per-type hooks applied (what ``surrogates/validation.py:61-99`` needs)."""

    def __init__(self, *args, **kwargs):
        self._structure_hooks: dict = {}
        self._registered: list = []

    def copy(self, *args, **kwargs):
        new = Converter()
        new._structure_hooks = dict(self._structure_hooks)
        return new

    # decorator or two-argument form
    def register_structure_hook(self, cl, func=None):
        if func is None and callable(cl) and not isinstance(cl, type):
            try:  # decorator form: the target is the ``type[T]`` annotation of the second parameter
                hints = [h for k, h in typing.get_type_hints(cl).items() if k != "return"]
                target = typing.get_args(hints[1])[0]
                self._structure_hooks[target] = cl
            except Exception:  # noqa: BLE001  (un-annotated / forward-referenced hooks: serialisation only)
                self._registered.append(("structure", cl, None))
            return cl
        self._structure_hooks[cl] = func
        return func

    def register_unstructure_hook(self, cl, func=None):
        self._registered.append(("unstructure", cl, func))
        return cl if func is None else func

    def register_structure_hook_func(self, check, func):
        self._registered.append(("structure_func", check, func))

    def register_unstructure_hook_func(self, check, func):
        self._registered.append(("unstructure_func", check, func))

    def register_structure_hook_factory(self, predicate, factory=None):
        self._registered.append(("structure_factory", predicate, factory))
        return (lambda f: f) if factory is None else factory

    def register_unstructure_hook_factory(self, predicate, factory=None):
        self._registered.append(("unstructure_factory", predicate, factory))
        return (lambda f: f) if factory is None else factory

    def get_structure_hook(self, cl, **kw):
        return lambda obj, tp=cl: self.structure(obj, tp)

    def get_unstructure_hook(self, cl, **kw):
        return lambda obj: self.unstructure(obj)

    def structure(self, obj, cl):
        return _structure(obj, cl, self._structure_hooks)

    def structure_attrs_fromdict(self, obj, cl):
        raise NotImplementedError("serialisation is outside the cattrs stand-in of the test suite")

    def unstructure(self, obj, unstructure_as=None):
        raise NotImplementedError("serialisation is outside the cattrs stand-in of the test suite")


from . import dispatch, gen, strategies  # noqa: E402,F401
