"""Linear (DLT) two-view triangulation (reference ``lib/epipolar/triangulation.py:9-62``) on the GPU."""
from __future__ import annotations

import numpy as np
import numpy.typing as npt

from .. import device
from ..common.feature import Feature
from ..transforms.transforms import Transform3D
from ._engine import feature_array


def _camera_rows(P: npt.NDArray) -> np.ndarray:
    """Rows 0..2 of a 3x4 or 4x4 camera matrix as 12 contiguous doubles (only those rows are used by
    the DLT system, reference triangulation.py:23-30)."""
    P = np.asarray(P, dtype=np.float64)
    if P.ndim != 2 or P.shape[0] < 3 or P.shape[1] != 4:
        raise ValueError(f"camera matrix must be 3x4 or 4x4, got {P.shape}")
    return np.ascontiguousarray(P[:3, :]).reshape(12)


def triangulate_point_correspondence(
    feature_a: Feature, feature_b: Feature, P1: npt.NDArray, P2: npt.NDArray
) -> npt.NDArray:
    """3-D position of the point seen as ``feature_a`` by camera ``P1`` and ``feature_b`` by ``P2``."""
    return _triangulate(np.array([[feature_a.x, feature_a.y]]), np.array([[feature_b.x, feature_b.y]]), P1, P2)[0]


def _triangulate(coords_a, coords_b, P1, P2) -> np.ndarray:
    device.require_gpu()
    if len(coords_a) == 0:
        return np.zeros((0, 3), dtype=np.float64)
    corr = device.to_device(np.hstack([coords_a, coords_b]))
    X = device.triangulate(corr, device.to_device(_camera_rows(P1)), device.to_device(_camera_rows(P2)))
    return X.cpu().numpy()


def triangulate_points(
    features_a: list[Feature],
    features_b: list[Feature],
    intrinsic_camera_matrix: npt.NDArray[float],
    cam2_T_cam1: Transform3D,
) -> npt.NDArray[float]:
    """Triangulate matched features (pixel coordinates) of two cameras: camera 1 at the origin, camera 2
    at ``cam2_T_cam1``.  Accepts lists or NumPy object arrays of ``Feature`` (the caller in the
    reference's apps/sfm.py:168-169 passes the latter).  Returns (M, 3) float64."""
    K = intrinsic_camera_matrix
    if (3, 3) != K.shape:
        raise ValueError(f"Camera intrinsic matrix is not 3x3, actual shape: {K.shape}")
    K_ext = np.hstack((K, np.zeros((3, 1))))
    cam1_T_world = Transform3D.identity()
    cam2_T_world = cam2_T_cam1 @ cam1_T_world
    P1 = K_ext @ cam1_T_world.Tmat
    P2 = K_ext @ cam2_T_world.Tmat
    pairs = list(zip(features_a, features_b))  # zip semantics: truncate to the shorter input
    return _triangulate(feature_array([p[0] for p in pairs]), feature_array([p[1] for p in pairs]), P1, P2)
