"""Symmetric epipolar distance (reference ``lib/epipolar/sed.py:7-30``), evaluated by the HIP kernel."""
from __future__ import annotations

import numpy as np
import numpy.typing as npt

from .. import device
from ..common.feature import Feature


def calculate_symmetric_epipolar_distance(
    feature_a: Feature, feature_b: Feature, e: npt.NDArray
) -> float:
    """SED (Hartley & Zisserman 11.10) of one correspondence under the essential matrix ``e``.

    Single-pair entry point of the reference; the same device routine (``sfm::sed_value``) is the body
    of the H x N scoring kernel.
    """
    return float(symmetric_epipolar_distances(
        np.array([[feature_a.x, feature_a.y]], dtype=np.float64),
        np.array([[feature_b.x, feature_b.y]], dtype=np.float64), e)[0])


def symmetric_epipolar_distances(coords_a: npt.NDArray, coords_b: npt.NDArray, e: npt.NDArray) -> npt.NDArray:
    """Batched form: (M,2),(M,2) coordinates -> (M,) distances."""
    device.require_gpu()
    corr = device.to_device(np.hstack([coords_a, coords_b]))
    E = device.to_device(np.asarray(e, dtype=np.float64).reshape(9))
    return device.sed_values(corr, E).cpu().numpy()
