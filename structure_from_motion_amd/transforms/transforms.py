"""Rigid transform value type used at the hot-path boundary.

``triangulate_points`` takes one of these as ``cam2_T_cam1`` and the pose-recovery code builds them
for the cheirality test, so the class keeps the reference's public surface
(reference ``lib/transforms/transforms.py:10-66``): ``from_rmat_t``, ``identity``, ``Tmat``, ``Rmat``,
``t`` (read/write view), ``inv``, ``*`` and ``@``.

The reference assembles the matrix through ``transforms3d.affines.compose(t, R, ones(3))``
(``transforms.py:30``); with unit zooms that is the block matrix ``[[R, t], [0, 1]]``, written out
here directly, so the third-party dependency disappears.
"""
from __future__ import annotations

import numpy as np

_HOMOGENEOUS_SHAPE = (4, 4)


def _block_matrix(rotation: np.ndarray, translation: np.ndarray) -> np.ndarray:
    """[[R, t], [0 0 0 1]] as a fresh float64 array."""
    out = np.zeros(_HOMOGENEOUS_SHAPE, dtype=np.float64)
    out[0:3, 0:3] = rotation
    out[0:3, 3] = translation
    out[3, 3] = 1.0
    return out


class Transform3D:
    """Thin wrapper over a 4x4 homogeneous matrix (no copy is taken of the argument)."""

    __slots__ = ("_Tmat",)

    def __init__(self, Tmat):
        if np.shape(Tmat) != _HOMOGENEOUS_SHAPE:
            raise ValueError("4x4 homogeneous transformation matrix expected")
        self._Tmat = Tmat

    # -- constructors -------------------------------------------------------------------------
    @classmethod
    def from_rmat_t(cls, rmat=None, t=None) -> "Transform3D":
        rotation = np.eye(3, dtype=float) if rmat is None else rmat
        if np.shape(rotation) != (3, 3):
            raise ValueError("3x3 matrix expected")
        translation = np.zeros(3, dtype=float) if t is None else np.asarray(t)
        # Same order of checks as the reference: reshape first (raises for a wrong element count).
        translation = translation.reshape((3,))
        if translation.size != 3:
            raise ValueError("3-element translation vector expected")
        return cls(_block_matrix(rotation, translation))

    @classmethod
    def identity(cls) -> "Transform3D":
        return cls(np.eye(4, dtype=np.float64))

    # -- views --------------------------------------------------------------------------------
    Tmat = property(lambda self: self._Tmat)
    Rmat = property(lambda self: self._Tmat[:3, :3])

    def _get_t(self):
        return self._Tmat[:3, 3]

    def _set_t(self, value):
        self._Tmat[:3, 3] = value

    t = property(_get_t, _set_t)

    # -- algebra ------------------------------------------------------------------------------
    def inv(self) -> "Transform3D":
        return type(self)(np.linalg.inv(self._Tmat))

    def __mul__(self, other) -> "Transform3D":
        if not isinstance(other, Transform3D):
            raise TypeError(
                f"Multiplication is only supported between {type(self)} objects."
            )
        return type(self)(self._Tmat @ other._Tmat)

    __matmul__ = __mul__

    def __str__(self) -> str:
        return f"Homogeneous transformation(\n{self._Tmat})"
