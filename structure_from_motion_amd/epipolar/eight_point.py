"""Eight-point estimation of F / E and pose recovery from E, with the reference's call signatures
(reference ``lib/epipolar/eight_point.py``).  The arithmetic runs in HIP kernels:

* ``estimate_fundamental_mat`` / ``estimate_essential_mat``  -> ``fit_eight_point_kernel``
  (Hartley normalisation, Y^T Y, Jacobi eigen-solve, rank-2 projection, de-normalisation);
* ``_recover_all_r_t``                                        -> ``decompose_essential_kernel``;
* ``_recover_r_t`` / ``recover_r_t_from_e``                   -> ``cheirality_kernel`` (4 poses x M pairs)
  followed by the reference's vote on the host (a 4 x M byte array).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import numpy.typing as npt
import torch

from .. import device
from .._native import FIT_DEGENERATE
from ..common.feature import Feature
from ..feature_matching.matching import Match
from ._engine import feature_array

DEFAULT_DISTANCE_THRESHOLD = 50.0  # reference eight_point.py:469-470


class EightPointCalculationError(Exception):
    """Raised if the computation cannot proceed due to ill-conditioned input data."""


# ------------------------------------------------------------------------------------------------------
# small host helpers of the reference's public surface
# ------------------------------------------------------------------------------------------------------
def to_normalized_image_coords(feature: Feature, camera_matrix: npt.NDArray) -> Feature:
    """Pixel -> normalised image coordinates: ``(x - cx) / fx, (y - cy) / fy`` (skew ignored, as in the
    reference, eight_point.py:127-133).  Scalar convenience form; the batch form used by the hot path is
    ``normalize_kernel`` (same two IEEE operations per coordinate)."""
    f_x, f_y = camera_matrix[0][0], camera_matrix[1][1]
    c_x, c_y = camera_matrix[0][2], camera_matrix[1][2]
    return Feature(x=(feature.x - c_x) / f_x, y=(feature.y - c_y) / f_y)


def create_trivial_matches(num_features: int) -> list[Match]:
    """``[Match(i, i, 0.0) for i in range(num_features)]``."""
    return [Match(a_index=i, b_index=i, match_score=0.0) for i in range(num_features)]


def _get_matching_coordinates(
    features_a: List[Feature], features_b: List[Feature], matches: List[Match]
) -> Tuple[np.ndarray, np.ndarray]:
    """Two (len(matches), 2) arrays of matched coordinates, in match order."""
    coords_a = feature_array([features_a[m.a_index] for m in matches])
    coords_b = feature_array([features_b[m.b_index] for m in matches])
    return coords_a, coords_b


# ------------------------------------------------------------------------------------------------------
# the reference's private fit helpers, each backed by the corresponding stage of the fit kernel
# ------------------------------------------------------------------------------------------------------
def _normalize_coords(coords: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Translate the centroid to the origin and scale the mean distance to sqrt(2).  Returns
    ``(normalised coords, T)`` with ``T`` the *forward* 3x3 transform (reference eight_point.py:308-338)."""
    device.require_gpu()
    return device.hartley_normalize(np.asarray(coords, dtype=np.float64))


def _get_normalized_match_coordinates(features_a, features_b, matches):
    """``(coords_a, T1, coords_b, T2)`` of the matched, Hartley-normalised coordinates (:341-360)."""
    coords_a, coords_b = _get_matching_coordinates(features_a, features_b, matches)
    return (*_normalize_coords(coords_a), *_normalize_coords(coords_b))


def _get_y_col(coord_a: np.ndarray, coord_b: np.ndarray) -> np.ndarray:
    """Row of the eight-point design matrix for one pair: [xb*xa, xb*ya, xb, yb*xa, yb*ya, yb, xa, ya, 1]
    (reference :378-393).  Four products of two scalars each: host arithmetic, IEEE-identical to the kernel's."""
    assert 2 == len(coord_a) and 2 == len(coord_b)
    xa, ya = coord_a
    xb, yb = coord_b
    return np.array([xb * xa, xb * ya, xb, yb * xa, yb * ya, yb, xa, ya, 1.0], dtype=np.float64)


def _get_yT_y(coords_a: np.ndarray, coords_b: np.ndarray) -> np.ndarray:
    """Y^T Y (9x9) of eight coordinate pairs, accumulated in point order (reference :363-375)."""
    assert len(coords_a) == len(coords_b) == 8
    device.require_gpu()
    return device.fit_stage(0, np.hstack([coords_a, coords_b]), 81).reshape(9, 9)


def _compute_f_est(yT_y: np.ndarray) -> np.ndarray:
    """Null vector of Y^T Y reshaped to 3x3; raises when more than one eigenvalue is ~0 (reference :396-427).
    The sign of the vector is arbitrary (as with LAPACK); callers normalise by F[2,2]."""
    assert (9, 9) == yT_y.shape
    device.require_gpu()
    out = device.fit_stage(1, yT_y, 19)
    if int(out[18]) & FIT_DEGENERATE:
        raise EightPointCalculationError(
            "More than one eigenvalue of Y.T @ Y is small. Cannot confidently estimate"
            " fundamental matrix."
        )
    return out[:9].reshape(3, 3)


def _enforce_fundamental_mat_constraints(f_est: np.ndarray) -> np.ndarray:
    """Closest rank-2 matrix: the smallest singular value set to zero (reference :430-446)."""
    device.require_gpu()
    return device.fit_stage(2, np.asarray(f_est, dtype=np.float64), 9).reshape(3, 3)


# ------------------------------------------------------------------------------------------------------
# fit
# ------------------------------------------------------------------------------------------------------
def _fit_eight(coords_a: np.ndarray, coords_b: np.ndarray) -> np.ndarray:
    """One eight-point fit on the device: coords (8,2)+(8,2) -> (3,3), raising on a degenerate sample."""
    dev = device.require_gpu()
    corr = device.to_device(np.hstack([coords_a, coords_b])).reshape(1, 8, 4)
    S = torch.arange(8, dtype=torch.int32, device=dev).reshape(1, 1, 8)
    E, flags = device.fit_eight_point(corr, S)
    if int(flags.cpu()[0, 0]) & FIT_DEGENERATE:
        raise EightPointCalculationError(
            "More than one eigenvalue of Y.T @ Y is small. Cannot confidently estimate"
            " fundamental matrix."
        )
    return E.cpu().numpy().reshape(3, 3)


def estimate_fundamental_mat(
    features_a: List[Feature],
    features_b: List[Feature],
    matches: List[Match],
) -> np.ndarray:
    """Fundamental matrix from exactly eight matches by the normalised eight-point algorithm; the result
    is scaled so that ``F[2, 2] == 1``.  Raises ``ValueError`` for a match count other than 8 and
    ``EightPointCalculationError`` for a degenerate configuration."""
    if 8 != len(matches):
        raise ValueError("Exactly eight matches are needed")
    coords_a, coords_b = _get_matching_coordinates(features_a, features_b, matches)
    return _fit_eight(coords_a, coords_b)


def estimate_essential_mat(
    *,
    camera_matrix: npt.NDArray,
    features_a: List[Feature],
    features_b: List[Feature],
    matches: List[Match],
):
    """Essential matrix from eight matches and the intrinsic matrix: the eight-point fit applied to
    K-normalised coordinates."""
    if 8 != len(matches):
        raise ValueError("Exactly eight matches are needed")
    coords_a, coords_b = _get_matching_coordinates(features_a, features_b, matches)
    device.require_gpu()
    corr = device.normalize_correspondences(
        device.to_device(coords_a), device.to_device(coords_b), camera_matrix).cpu().numpy()
    return _fit_eight(corr[:, 0:2], corr[:, 2:4])


def estimate_r_t(
    camera_matrix: npt.NDArray,
    features_a: List[Feature],
    features_b: List[Feature],
    matches: List[Match],
):
    """Eight matches -> E -> ``recover_r_t_from_e`` on the matched subset.  Returns ``(R, t, mask)``
    (three values, as the reference actually does: eight_point.py:57-62)."""
    if not features_a or not features_b:
        raise ValueError("Need some matching features")
    e = estimate_essential_mat(
        camera_matrix=camera_matrix, features_a=features_a, features_b=features_b, matches=matches)
    return recover_r_t_from_e(
        e=e,
        camera_matrix=camera_matrix,
        features_a=[features_a[m.a_index] for m in matches],
        features_b=[features_b[m.b_index] for m in matches],
    )


# ------------------------------------------------------------------------------------------------------
# pose recovery
# ------------------------------------------------------------------------------------------------------
def recover_r_t_from_e(
    e: npt.NDArray,
    camera_matrix: npt.NDArray,
    features_a: list[Feature],
    features_b: list[Feature],
    distance_threshold: float | None = None,
):
    """Rotation and unit translation from ``e``: of the four decompositions, the one for which the most
    correspondences triangulate in front of both cameras (and within ``distance_threshold``, default 50).

    ``features_*`` are pixel coordinates.  Returns ``(cam2_R_cam1 (3,3), cam2_t_cam2_cam1 (3,), mask)``
    with ``mask`` the int64 indices of the correspondences that pass for the chosen pose."""
    device.require_gpu()
    corr = device.normalize_correspondences(
        device.to_device(feature_array(features_a)), device.to_device(feature_array(features_b)),
        camera_matrix)
    return _recover_r_t_device(corr, e, distance_threshold)


def _recover_r_t(
    features_a: list[Feature],
    features_b: list[Feature],
    e: np.ndarray,
    distance_threshold: float | None = None,
):
    """As ``recover_r_t_from_e`` for features already in normalised image coordinates."""
    device.require_gpu()
    corr = device.to_device(np.hstack([feature_array(features_a), feature_array(features_b)]))
    return _recover_r_t_device(corr, e, distance_threshold)


def _decompose(e: np.ndarray) -> np.ndarray:
    """(4, 12) candidate poses [R | t] in the order (R1,t),(R1,-t),(R2,t),(R2,-t)."""
    E = device.to_device(np.asarray(e, dtype=np.float64).reshape(1, 9))
    poses, status = device.decompose_essential(E)
    if int(status.cpu()[0]) != 0:
        raise EightPointCalculationError(
            "The smallest singular value of the Essential matrix is expected to be ~0"
        )
    return poses


def _recover_all_r_t(e: np.ndarray):
    """The two rotations and one translation (unit norm) E decomposes into: ``(R_1, R_2, t_1)``."""
    device.require_gpu()
    poses = _decompose(e).cpu().numpy()[0]
    return poses[0, :9].reshape(3, 3), poses[2, :9].reshape(3, 3), poses[0, 9:12].copy()


def _recover_r_t_device(corr: torch.Tensor, e: np.ndarray, distance_threshold):
    if distance_threshold is None:
        distance_threshold = DEFAULT_DISTANCE_THRESHOLD
    poses = _decompose(e)[0]
    m = corr.shape[0]
    if m:
        passes = device.cheirality(corr, poses, distance_threshold).cpu().numpy()
    else:
        passes = np.zeros((4, 0), dtype=np.uint8)
    # Vote exactly as the reference (eight_point.py:213-237): indices via nonzero, votes via
    # count_nonzero of the *index array* (so a passing pair at index 0 is not counted), first maximum.
    index_sets = [np.nonzero(passes[c])[0] for c in range(4)]
    votes = [np.count_nonzero(idx) for idx in index_sets]
    if 0 == np.count_nonzero(votes):
        raise EightPointCalculationError("None of the transformations pass the cheirality check.")
    best = int(np.argmax(votes))
    pose = poses[best].cpu().numpy()
    return pose[:9].reshape(3, 3).copy(), pose[9:12].copy(), index_sets[best]


def _cheirality_check(
    feature_a: Feature,
    feature_b: Feature,
    cam2_R_cam1: npt.NDArray,
    cam2_t_cam2_cam1: npt.NDArray,
    distance_threshold: float | None = None,
    z_axis_index: int = 2,
) -> bool:
    """Whether the pose places the triangulated point in front of both cameras (normalised coords).
    ``z_axis_index`` names the coordinate that points out of the cameras (reference eight_point.py:455,481)."""
    if z_axis_index not in (0, 1, 2):
        raise IndexError(f"index {z_axis_index} is out of bounds for axis 0 with size 3")
    if distance_threshold is None:
        distance_threshold = DEFAULT_DISTANCE_THRESHOLD
    device.require_gpu()
    corr = device.to_device(np.array([[feature_a.x, feature_a.y, feature_b.x, feature_b.y]]))
    R = np.asarray(cam2_R_cam1, dtype=np.float64).reshape(3, 3)
    t = np.asarray(cam2_t_cam2_cam1, dtype=np.float64).reshape(3)
    if z_axis_index == 2:  # the viewing direction the pose kernels are written for
        pose = np.concatenate([R.reshape(9), t])
        return bool(device.cheirality(corr, device.to_device(pose.reshape(1, 12)), distance_threshold).cpu()[0, 0])
    # another outward axis (a diagnostic option of this one-pair helper; the batched pose path is z-only): the point
    # comes from the triangulation kernel, the three comparisons of eight_point.py:478-486 are applied to it here
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = np.hstack([R, t.reshape(3, 1)])
    in_cam1 = device.triangulate(corr, device.to_device(P1.reshape(12)), device.to_device(P2.reshape(12))).cpu().numpy()[0]
    in_cam2 = R @ in_cam1 + t
    tolerance = 1e-8
    return bool(in_cam1[z_axis_index] >= -tolerance and in_cam2[z_axis_index] >= -tolerance
                and np.linalg.norm(in_cam1) <= distance_threshold)
