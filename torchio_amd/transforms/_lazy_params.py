"""``LazyParams`` — a params dict whose bulky entries become nested lists only when read.

The reference records every transform's parameters as plain JSON-serialisable data
(``transforms/transform.py:29-43``); for the elastic path that means 7x7x7x3 control
points per batch element as nested Python lists.  Building those lists (``Tensor.tolist``)
and parsing them back a few microseconds later in ``apply_transform`` was the single
largest host cost of a bench step (~35 %) and made the pipeline host-bound.

``LazyParams`` is a ``dict`` subclass: ``make_params`` parks the tensors with
:meth:`set_lazy`; ``apply_transform`` takes them back with :meth:`raw`; anything that
*reads* the entry (``params[key]``, ``.items()``, ``==``, ``json.dumps``, ``deepcopy``,
pickling, ``repr``) sees — and from then on stores — exactly the nested lists the eager
version would have produced.  History replay and ``inverse`` therefore behave as before.
"""
from __future__ import annotations

import copy as _copy
from typing import Any

import numpy as np
from torch import Tensor


def _to_lists(value):
    if value is None:
        return None
    if isinstance(value, Tensor):
        return value.detach().cpu().tolist()
    if isinstance(value, np.ndarray):
        return value.tolist()
    return [_to_lists(v) for v in value]  # per-instance list of tensors / None


class LazyParams(dict):
    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._lazy: dict[str, Any] = {}

    # -- producer / consumer side ------------------------------------------------------
    def set_lazy(self, key: str, value) -> None:
        """Park *value* (a tensor, ``None`` or a list of those) under *key*."""
        dict.__setitem__(self, key, None)  # placeholder: keeps the key and its insertion order
        self._lazy[key] = value

    def raw(self, key: str):
        """``(True, parked value)`` while *key* is still unread, else ``(False, None)``."""
        if key in self._lazy:
            return True, self._lazy[key]
        return False, None

    def _materialise(self, key: str | None = None) -> None:
        if not self._lazy:
            return
        keys = list(self._lazy) if key is None else ([key] if key in self._lazy else [])
        for k in keys:
            dict.__setitem__(self, k, _to_lists(self._lazy.pop(k)))

    # -- every read path goes through materialisation -------------------------------------
    def __getitem__(self, key):
        self._materialise(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._materialise(key)
        return dict.get(self, key, default)

    def __setitem__(self, key, value) -> None:
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)

    def pop(self, key, *default):
        self._materialise(key)
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        self._materialise(key)
        return dict.setdefault(self, key, default)

    # dict(params), {**params} and other.update(params) take CPython's C fast path for dict subclasses unless
    # ``__iter__`` / ``keys`` are overridden: with them overridden the generic mapping protocol (keys() +
    # __getitem__) is used, which materialises — otherwise a copy would carry the ``None`` placeholders.
    def __iter__(self):
        self._materialise()
        return dict.__iter__(self)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)

    def copy(self):
        self._materialise()
        return dict(self)

    def __eq__(self, other) -> bool:
        self._materialise()
        if isinstance(other, LazyParams):
            other._materialise()
        return dict.__eq__(self, other)

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    __hash__ = None  # type: ignore[assignment]

    def __repr__(self) -> str:
        self._materialise()
        return dict.__repr__(self)

    def __deepcopy__(self, memo):
        self._materialise()
        return _copy.deepcopy(dict(self), memo)

    def __reduce__(self):
        self._materialise()
        return (dict, (dict(self),))
