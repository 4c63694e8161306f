"""Voxel-to-world affine (mirror of reference ``src/torchio/data/affine.py:20-248``).

Only what the augmentation hot path touches: a float64 4x4 on the host side,
``spacing`` / ``origin`` / ``direction`` and conversion helpers.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor


class AffineMatrix:
    """4x4 float64 matrix mapping voxel indices to world (mm) coordinates."""

    __slots__ = ("_matrix", "_spacing_cache", "_bytes_cache", "_inverse_cache")

    def __init__(self, matrix=None) -> None:
        if matrix is None:
            value = torch.eye(4, dtype=torch.float64)
        elif isinstance(matrix, AffineMatrix):
            value = matrix._matrix.clone()
        elif isinstance(matrix, Tensor):
            value = matrix.detach().to(torch.float64).clone()
        else:
            value = torch.from_numpy(np.array(matrix, dtype=np.float64, copy=True))
        if tuple(value.shape) != (4, 4):
            raise ValueError(f"AffineMatrix must be 4x4, got {tuple(value.shape)}")
        self._matrix = value
        self._spacing_cache = None  # (tensor version, spacing): the 4x4 may be edited in place through .data
        self._bytes_cache = None    # (tensor version, the 128 bytes of the matrix): `same_values`
        self._inverse_cache = None  # (tensor version, np.linalg.inv of the matrix): `inverse_numpy`

    @classmethod
    def from_spacing(cls, spacing, *, origin=(0.0, 0.0, 0.0), direction=None) -> "AffineMatrix":
        matrix = torch.eye(4, dtype=torch.float64)
        if direction is not None:
            matrix[:3, :3] = torch.as_tensor(np.asarray(direction, dtype=np.float64))
        matrix[:3, :3] *= torch.as_tensor(spacing, dtype=torch.float64)
        matrix[:3, 3] = torch.as_tensor(origin, dtype=torch.float64)
        return cls(matrix)

    @property
    def data(self) -> Tensor:
        return self._matrix

    @property
    def device(self) -> torch.device:
        return self._matrix.device

    def _column_norms(self) -> Tensor:
        return torch.sqrt(torch.sum(self._matrix[:3, :3] ** 2, dim=0))

    @property
    def spacing(self) -> tuple[float, float, float]:
        version = self._matrix._version
        cached = self._spacing_cache
        if cached is not None and cached[0] == version:
            return cached[1]
        norms = self._column_norms().tolist()  # (ATen's float64 reduction: numpy adds the three squares in another order)
        value = (float(norms[0]), float(norms[1]), float(norms[2]))
        self._spacing_cache = (version, value)
        return value

    @property
    def origin(self) -> tuple[float, float, float]:
        o = self._matrix[:3, 3]
        return (float(o[0]), float(o[1]), float(o[2]))

    @property
    def direction(self) -> Tensor:
        return self._matrix[:3, :3] / self._column_norms()

    @property
    def orientation(self) -> tuple[str, str, str]:
        """Anatomical orientation codes such as ``('R', 'A', 'S')`` (reference affine.py:124-128).

        The reference asks nibabel (``aff2axcodes``); this is the same published algorithm: the
        closest orthogonal matrix to the zoom-normalised direction block (SVD), then every voxel
        axis claims the world axis on which it has the largest absolute component, each world
        axis being claimed at most once.
        """
        block = self._matrix[:3, :3].cpu().numpy().astype(np.float64)
        zooms = np.sqrt(np.sum(block * block, axis=0))
        zooms[zooms == 0] = 1.0
        left, singular, right = np.linalg.svd(block / zooms)
        keep = singular > singular.max() * 3 * np.finfo(singular.dtype).eps
        rotation = left[:, keep] @ right[keep]
        labels = (("L", "R"), ("P", "A"), ("I", "S"))
        codes: list[str] = []
        for axis in range(3):
            column = rotation[:, axis]
            if np.allclose(column, 0):
                codes.append(None)  # type: ignore[arg-type]
                continue
            world = int(np.argmax(np.abs(column)))
            codes.append(labels[world][0 if column[world] < 0 else 1])
            rotation[world, :] = 0  # a world axis is assigned once
        return (codes[0], codes[1], codes[2])

    def _value_bytes(self) -> bytes:
        version = self._matrix._version
        cached = self._bytes_cache
        if cached is not None and cached[0] == version:
            return cached[1]
        value = self._matrix.numpy().tobytes()
        self._bytes_cache = (version, value)
        return value

    def same_values(self, other: "AffineMatrix") -> bool:
        """``torch.equal(self.data, other.data)`` for host matrices (NaN-free grids: byte equality, except that +0.0 and
        -0.0 compare unequal here — callers fall back to the tensor comparison on ``False``); cached per tensor version."""
        return self is other or self._value_bytes() == other._value_bytes()

    def inverse_numpy(self) -> np.ndarray:
        """``np.linalg.inv(self.numpy())`` (read-only; cached per tensor version: the same grid is inverted by every
        spatial transform of a pipeline)."""
        version = self._matrix._version
        cached = self._inverse_cache
        if cached is not None and cached[0] == version:
            return cached[1]
        value = np.linalg.inv(self._matrix.numpy())
        value.setflags(write=False)
        self._inverse_cache = (version, value)
        return value

    def to(self, *args, **kwargs) -> "AffineMatrix":
        """Affines stay float64 and — unlike image data — on the host.

        The reference moves the 4x4 along with the data (affine.py ``to``); every
        consumer on the hot path immediately reads it back with ``float()`` /
        ``.numpy()`` (spatial.py:1594, affine.py:104-109), i.e. one device sync
        per access.  Keeping it on the CPU removes those syncs and changes no
        value.
        """
        return self

    def clone(self) -> "AffineMatrix":
        new = AffineMatrix.__new__(AffineMatrix)
        new._matrix = self._matrix.clone()
        # The copy inherits what is cached for these values.  The spacing is worked out HERE if nobody asked the source yet:
        # a pipeline copies its input batch every step and asks the COPIES (four tensor ops per element and step, for ever);
        # asked once of the source, every later copy starts with the answer.
        if self._spacing_cache is None or self._spacing_cache[0] != self._matrix._version:
            self.spacing  # noqa: B018 - fills the cache
        cached = self._spacing_cache
        # the copy starts at tensor version 0 with the same values: the spacing carries over
        version = self._matrix._version
        new._spacing_cache = (0, cached[1]) if cached is not None and cached[0] == version else None
        cached = self._bytes_cache
        new._bytes_cache = (0, cached[1]) if cached is not None and cached[0] == version else None
        cached = self._inverse_cache
        new._inverse_cache = (0, cached[1]) if cached is not None and cached[0] == version else None
        return new

    def inverse(self) -> "AffineMatrix":
        return AffineMatrix(torch.linalg.inv(self._matrix))

    def compose(self, other: "AffineMatrix") -> "AffineMatrix":
        """``self @ other`` as a new ``AffineMatrix`` (reference affine.py:178-183)."""
        return AffineMatrix(self._matrix @ other._matrix)

    def apply(self, points) -> Tensor:
        """Map an ``(N, 3)`` set of points through the affine, in float64 (reference affine.py:185-205)."""
        if not isinstance(points, Tensor):
            pts = torch.as_tensor(np.asarray(points, dtype=np.float64), dtype=torch.float64)
        else:
            pts = points.to(torch.float64)
        pts = pts.to(self._matrix.device)
        homogeneous = torch.cat([pts, torch.ones(pts.shape[0], 1, dtype=torch.float64, device=pts.device)], dim=1)
        return (self._matrix @ homogeneous.T).T[:, :3]

    def numpy(self) -> np.ndarray:
        return self._matrix.cpu().numpy()

    def __matmul__(self, other):
        if not isinstance(other, AffineMatrix):
            return NotImplemented
        return AffineMatrix(self._matrix @ other._matrix)

    def __array__(self, dtype=None, copy=None):
        array = self.numpy()
        return array.astype(dtype) if dtype is not None else (array.copy() if copy else array)

    def __eq__(self, other) -> bool:
        if not isinstance(other, AffineMatrix):
            return NotImplemented
        return torch.equal(self._matrix, other._matrix)

    def __hash__(self):  # pragma: no cover - identity hash like the reference's default
        return id(self)

    def __copy__(self):
        return self.clone()

    def __deepcopy__(self, memo):
        new = self.clone()
        memo[id(self)] = new
        return new

    def __repr__(self) -> str:
        sp = ", ".join(f"{s:.2f}" for s in self.spacing)
        o = ", ".join(f"{v:.2f}" for v in self.origin)
        return f"AffineMatrix(spacing=({sp}), origin=({o}))"
