"""RANSAC estimation of the essential matrix — the function apps/sfm.py calls
(reference ``lib/epipolar/epipolar_ransac.py:18-70``)."""
from __future__ import annotations

from functools import partial
from typing import Tuple

import numpy.typing as npt

from ..common.feature import Feature
from ..feature_matching.matching import Match
from ..ransac.ransac import ErrorAggregationMethod, fit_with_ransac
from . import _engine
from .eight_point import estimate_essential_mat, to_normalized_image_coords
from .sed import calculate_symmetric_epipolar_distance

FeaturePair = Tuple[Feature, Feature]


def calculate_sed_inlier_score(
    e: npt.NDArray, matching_features: FeaturePair, camera_matrix: npt.NDArray
) -> float:
    """SED of one pixel-coordinate pair under ``e`` after K-normalisation (the RANSAC scorer)."""
    feature_a = to_normalized_image_coords(matching_features[0], camera_matrix)
    feature_b = to_normalized_image_coords(matching_features[1], camera_matrix)
    return calculate_symmetric_epipolar_distance(feature_a=feature_a, feature_b=feature_b, e=e)


def eight_point_model_fitter(
    matching_features: list[FeaturePair], camera_matrix: npt.NDArray
) -> npt.NDArray:
    """Essential matrix from exactly eight pixel-coordinate pairs (the RANSAC model fitter)."""
    if 8 != len(matching_features):
        raise ValueError("Eight feature pairs are expected.")
    return estimate_essential_mat(
        camera_matrix=camera_matrix,
        features_a=[pair[0] for pair in matching_features],
        features_b=[pair[1] for pair in matching_features],
        matches=[Match(a_index=i, b_index=i) for i in range(8)],
    )


# fit_with_ransac recognises partials of these two and runs the whole loop on the GPU.
eight_point_model_fitter._sfm_hip_role = "eight_point_fitter"
calculate_sed_inlier_score._sfm_hip_role = "sed_scorer"


def estimate_essential_mat_with_ransac(
    camera_matrix: npt.NDArray,
    features_a: list[Feature],
    features_b: list[Feature],
    matches: list[Match],
    sed_inlier_threshold: float,
    min_num_extra_inliers: int | None = None,
    error_aggregation_method: ErrorAggregationMethod | None = None,
    max_iterations: int | None = None,
) -> Tuple[npt.NDArray, list[FeaturePair]]:
    """Estimate E from matched pixel features with RANSAC over eight-point hypotheses scored by SED in
    K-normalised coordinates.  Returns ``(E with E[2,2] == 1, inlier (Feature, Feature) pairs)``.

    Raises ``ValueError`` when no hypothesis has enough inliers and ``EightPointCalculationError`` when
    a sampled eight-tuple is degenerate (reference behaviour; ``SFM_DEGENERATE=skip`` ignores such
    hypotheses instead)."""
    with _engine.gc_paused():  # bulk creation of pair tuples and inlier copies: see _engine.gc_paused
        feature_pairs = _engine.match_pairs(features_a, features_b, matches)
        e, inlier_feature_pairs = fit_with_ransac(
            feature_pairs,
            model_fit_data_count=8,
            model_fitter=partial(eight_point_model_fitter, camera_matrix=camera_matrix),
            inlier_scorer=partial(calculate_sed_inlier_score, camera_matrix=camera_matrix),
            inlier_threshold=sed_inlier_threshold,
            min_num_extra_inliers=min_num_extra_inliers,
            error_aggregation_method=error_aggregation_method,
            max_iterations=max_iterations,
        )
    if e is None:
        raise ValueError("Could not estimate Essential Matrix with RANSAC.")
    return e, inlier_feature_pairs
