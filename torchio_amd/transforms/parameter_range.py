"""Scalar-or-range transform parameters (mirror of reference ``transforms/parameter_range.py``).

This module fixes the ORDER in which the global CPU RNG is consumed, which is
what makes ``torch.manual_seed(k)`` reproduce the reference's sampled parameters
(SURVEY.md §8c "Global-RNG draw order"): one ``uniform_`` draw per
non-degenerate axis, none for constants or ``lo == hi`` (parameter_range.py:91-137).
"""
from __future__ import annotations

from collections.abc import Sequence

import math

import numpy as np
import torch
from torch.distributions import Distribution


class Choice:
    """A discrete set of values to sample from (parameter_range.py:27-82)."""

    def __init__(self, values: Sequence[float | int], probabilities: Sequence[float] | None = None) -> None:
        if len(values) < 1:
            raise ValueError("Choice requires at least one value")
        self._values = torch.tensor([float(v) for v in values])
        if probabilities is None:
            self._probs = torch.ones(len(values)) / len(values)
        else:
            if len(probabilities) != len(values):
                raise ValueError(f"Expected {len(values)} probabilities, got {len(probabilities)}")
            self._probs = torch.tensor([float(p) for p in probabilities])

    def sample(self) -> float:
        index = int(torch.multinomial(self._probs, 1).item())
        return float(self._values[index])

    def sample_batched(self, n: int) -> torch.Tensor:
        return self._values[torch.multinomial(self._probs, n, replacement=True)]

    def __repr__(self) -> str:
        values = ", ".join(f"{v:.1f}" if v == int(v) else f"{v}" for v in self._values.tolist())
        if torch.allclose(self._probs, self._probs[0].expand_as(self._probs)):
            return f"Choice([{values}])"
        probs = ", ".join(f"{p:.2f}" for p in self._probs.tolist())
        return f"Choice([{values}], p=[{probs}])"


def _is_number(x) -> bool:
    return isinstance(x, (int, float))


def _draw(spec, n: int | None, generator):
    """One draw (``n is None`` → float) or ``n`` draws (→ ``(n,)`` tensor) from an axis spec."""
    if _is_number(spec):
        return float(spec) if n is None else torch.full((n,), float(spec))
    if isinstance(spec, Choice):
        return spec.sample() if n is None else spec.sample_batched(n)
    if isinstance(spec, Distribution):
        if n is None:
            return spec.sample().item()
        return spec.sample((n,)).reshape(n).to(torch.float32)
    low, high = spec
    if low == high:
        return float(low) if n is None else torch.full((n,), float(low))
    drawn = torch.empty(1 if n is None else n).uniform_(float(low), float(high), generator=generator)
    return drawn.item() if n is None else drawn


def _fma32(x: float, slope: float, offset: float) -> float:
    """``fmaf(x, slope, offset)`` for float32 values held in Python floats.

    ATen's CPU ``uniform_(lo, hi)`` is ``fma(u, hi - lo, lo)`` in float32 (pinned against torch
    2.10 on 20 000 random ranges).  The product of two float32 values is exact in float64; the
    float64 sum and its exact error (TwoSum) then decide the float32 rounding: the only case
    where rounding the float64 sum again could disagree with a true fma is a sum that sits
    exactly on a float32 midpoint while the error term is non-zero (the 24-bit uniforms make
    exact midpoints common, so this path is real).
    """
    product = x * slope
    total = product + offset
    rounded = float(np.float32(total))
    if rounded != total:
        virtual = total - product
        error = (product - (total - virtual)) + (offset - virtual)  # exact: product + offset == total + error
        if error != 0.0:
            neighbour = float(np.nextafter(np.float32(rounded), np.float32(math.inf if total > rounded else -math.inf)))
            if (rounded + neighbour) / 2 == total:  # float64 sum on a float32 midpoint: the error term breaks the tie
                return max(rounded, neighbour) if error > 0 else min(rounded, neighbour)
    return rounded


class ScalarDrawPlan:
    """Several ``_ParameterRange.sample()`` calls in a row as ONE small ``uniform_`` call.

    ``torch.empty(n).uniform_(0, 1)`` for ``n < 16`` consumes the CPU generator exactly like
    ``n`` successive one-element draws (ATen's serial path), and ``uniform_(lo, hi)`` is
    ``fma(u, hi - lo, lo)``; so the values - and the generator state afterwards - are those of
    the reference's one-draw-per-axis sequence (parameter_range.py:97-106) at a fraction of the
    Python / dispatcher cost.  Only number and ``(lo, hi)`` axis specs qualify.
    """

    def __init__(self, axes: list) -> None:
        self.entries = []  # (constant, slope, offset): constant is None for a random axis
        for axis in axes:
            if _is_number(axis):
                self.entries.append((float(axis), 0.0, 0.0))
            else:
                low, high = axis
                if low == high:
                    self.entries.append((float(low), 0.0, 0.0))
                else:
                    low32, high32 = np.float32(low), np.float32(high)
                    self.entries.append((None, float(np.float32(high32 - low32)), float(low32)))
        self.n_random = sum(1 for constant, _, _ in self.entries if constant is None)
        # column layout of `map_block_array`: where the constants and the drawn values go, and the fma constants as arrays
        self._random_columns = np.array([i for i, (constant, _, _) in enumerate(self.entries) if constant is None], dtype=np.intp)
        self._constant_columns = np.array([i for i, (constant, _, _) in enumerate(self.entries) if constant is not None], dtype=np.intp)
        self._constant_values = np.array([constant for constant, _, _ in self.entries if constant is not None], dtype=np.float64)
        self._slopes = np.array([slope for constant, slope, _ in self.entries if constant is None], dtype=np.float64)
        self._offsets = np.array([offset for constant, _, offset in self.entries if constant is None], dtype=np.float64)

    @staticmethod
    def build(ranges: list, counts: list[int]) -> "ScalarDrawPlan | None":
        """Plan for the first ``counts[i]`` axes of each range, or ``None`` if any spec needs the general path."""
        axes = []
        for parameter_range, count in zip(ranges, counts, strict=True):
            for axis in parameter_range._axes[:count]:
                if not (_is_number(axis) or isinstance(axis, tuple)):
                    return None
                axes.append(axis)
        plan = ScalarDrawPlan(axes)
        return plan if plan.n_random < 16 else None  # >= 16 values take ATen's vectorised path: different stream

    def sample(self) -> list[float]:
        uniforms = torch.empty(self.n_random).uniform_(0.0, 1.0).tolist() if self.n_random else []
        return self.map(uniforms)

    def map_block_array(self, uniforms: np.ndarray) -> np.ndarray:
        """``np.array([self.map(row) for row in uniforms])`` for a ``(rows, n_random)`` float32 block: ``(rows, entries)`` float64.

        ``fma(u, slope, offset)`` in float32 is the float64 ``u * slope + offset`` (the product is exact) rounded once
        more — except when that float64 sum sits exactly on a float32 midpoint (see ``_fma32``); those entries, found
        by their bit pattern, are redone with the scalar routine.
        """
        rows = uniforms.shape[0]
        out = np.empty((rows, len(self.entries)), dtype=np.float64)
        if self._constant_columns.size:
            out[:, self._constant_columns] = self._constant_values
        if self.n_random:
            slopes, offsets = self._slopes, self._offsets
            u = uniforms.astype(np.float64)
            total = u * slopes + offsets
            mapped = total.astype(np.float32).astype(np.float64)
            ties = (total.view(np.uint64) & np.uint64(0x1FFFFFFF)) == np.uint64(0x10000000)  # a float32 midpoint held in float64
            if ties.any():
                for r, c in zip(*np.nonzero(ties)):
                    mapped[r, c] = _fma32(float(u[r, c]), float(slopes[c]), float(offsets[c]))
            out[:, self._random_columns] = mapped
        return out

    def map_block(self, uniforms: np.ndarray) -> list[list[float]]:
        """``[self.map(row) for row in uniforms]`` for a ``(rows, n_random)`` float32 block (see ``map_block_array``)."""
        return self.map_block_array(uniforms).tolist()

    def map(self, uniforms: list[float]) -> list[float]:
        """The plan's values for ``n_random`` uniforms drawn elsewhere (a batch's draws come as one block)."""
        position = 0
        values = []
        for constant, slope, offset in self.entries:
            if constant is not None:
                values.append(constant)
            else:
                values.append(_fma32(uniforms[position], slope, offset))
                position += 1
        return values


def _parse_axis(spec):
    if _is_number(spec):
        return float(spec)
    if isinstance(spec, (Choice, Distribution)):
        return spec
    if isinstance(spec, tuple) and len(spec) == 2 and _is_number(spec[0]) and _is_number(spec[1]):
        return (float(spec[0]), float(spec[1]))
    raise TypeError(
        f"Per-axis spec must be a float, (lo, hi) tuple, Choice, or Distribution, got {type(spec).__name__}"
    )


def _parse_tuple(value: tuple):
    n = len(value)
    if n == 3:
        if all(_is_number(v) for v in value):
            return tuple(float(v) for v in value)
        return tuple(_parse_axis(v) for v in value)
    if not all(_is_number(v) for v in value):
        raise ValueError(f"Mixed per-axis specs require exactly 3 elements, got {n}")
    if n == 1:
        return (float(value[0]),) * 3
    if n == 2:
        return ((float(value[0]), float(value[1])),) * 3
    if n == 6:
        return tuple((float(value[2 * a]), float(value[2 * a + 1])) for a in range(3))
    raise ValueError(f"Tuple must have 1, 2, 3, or 6 elements, got {n}")


class _ParameterRange:
    """Three per-axis specs parsed from ``float | (lo, hi) | (a, b, c) | 6-tuple | Choice | Distribution``."""

    def __init__(self, value) -> None:
        self._original = value
        if _is_number(value):
            self._axes = (float(value),) * 3
        elif isinstance(value, (Choice, Distribution)):
            self._axes = (value,) * 3
        elif isinstance(value, tuple):
            self._axes = _parse_tuple(value)
        else:
            raise TypeError(f"Expected float, tuple, Distribution, or Choice, got {type(value).__name__}")

    @property
    def is_deterministic(self) -> bool:
        return all(_is_number(a) for a in self._axes)

    def is_constant(self, value: float) -> bool:
        for axis in self._axes:
            if _is_number(axis):
                if float(axis) != float(value):
                    return False
            elif isinstance(axis, tuple):
                if not (axis[0] == axis[1] == value):
                    return False
            else:
                return False
        return True

    @property
    def _ranges(self):
        out = []
        for axis in self._axes:
            if _is_number(axis):
                out.append((float(axis), float(axis)))
            elif isinstance(axis, tuple):
                out.append(axis)
            else:
                out.append((0.0, 0.0))
        return tuple(out)

    @property
    def _distribution(self):
        return self._axes[0] if isinstance(self._axes[0], Distribution) else None

    def sample(self, n: int | None = None, *, generator: torch.Generator | None = None):
        """A 3-tuple of floats, or an ``(n, 3)`` tensor (axis 0 drawn first, then 1, then 2)."""
        drawn = [_draw(axis, n, generator) for axis in self._axes]
        return tuple(drawn) if n is None else torch.stack(drawn, dim=-1)

    def sample_1d(self, n: int | None = None, *, generator: torch.Generator | None = None):
        """One float (first axis spec), or an ``(n,)`` tensor."""
        return _draw(self._axes[0], n, generator)

    def __repr__(self) -> str:
        v = self._original
        if isinstance(v, tuple):
            return "(" + ", ".join(repr(x) for x in v) + ")"
        return repr(v) if isinstance(v, (Distribution, Choice)) else str(v)


def to_range(value) -> _ParameterRange:
    return _ParameterRange(value)


def to_nonneg_range(value) -> _ParameterRange:
    parsed = _ParameterRange(value)
    if parsed._distribution is None and any(lo < 0 or hi < 0 for lo, hi in parsed._ranges):
        raise ValueError(f"Value must be non-negative, got {value}")
    return parsed
