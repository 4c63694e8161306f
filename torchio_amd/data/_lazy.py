"""Copy-on-replace for device-resident batches.

The reference deep-copies its input before transforming it (``copy=True``,
reference transform.py:220-221) so that the caller's tensors are never touched.
The HIP engine never writes in place — every op allocates its output — so cloning
the image tensors up front only to replace them one transform later moves
``2 x batch bytes`` through HBM for nothing.  Inside a :class:`LazyCopyScope` the
``__deepcopy__`` of an image container shares the tensor instead and registers
itself; when the scope is closed every container that still holds the shared tensor
(no transform replaced it: gated out, excluded, empty Compose …) gets its private
clone.  The caller-visible contract is unchanged: the result never aliases the input.
"""
from __future__ import annotations

import threading

_state = threading.local()


def _shares_storage(a, b) -> bool:
    try:
        return a is not None and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    except (AttributeError, RuntimeError):
        return False


def active_scope() -> "LazyCopyScope | None":
    return getattr(_state, "scope", None)


class LazyCopyScope:
    def __init__(self) -> None:
        self._borrowed: list[tuple[object, object]] = []
        self._previous: LazyCopyScope | None = None

    def __enter__(self) -> "LazyCopyScope":
        self._previous = active_scope()
        _state.scope = self
        return self

    def __exit__(self, *exc) -> None:
        _state.scope = self._previous

    def borrow(self, holder, tensor):
        self._borrowed.append((holder, tensor))
        return tensor

    def materialise(self) -> None:
        """Give every container that still shares its input tensor a private copy."""
        for holder, tensor in self._borrowed:
            if getattr(holder, "_pending", None) is not None:
                holder._flush()  # deferred stages replace the tensor: no clone needed afterwards
            current = holder._data
            if current is tensor:
                holder._data = tensor.clone()
            elif _shares_storage(current, tensor):
                # a transform handed back a VIEW of the borrowed tensor (Crop slices, a user transform that
                # writes ``img.data[...]`` in place and re-assigns it): still the caller's memory
                holder._data = current.clone()
        self._borrowed.clear()
