"""Normalised cross-correlation window score (reference ``lib/feature_matching/ncc.py:7-54``)."""
import numpy as np

from ..common import feature as feat
from . import _device_match
from .._native import MATCH_NCC


def calculate_ncc(
    image_a: np.ndarray,
    image_b: np.ndarray,
    feature_a: feat.Feature,
    feature_b: feat.Feature,
    window_size: int = 3,
) -> float:
    """``1 - NCC`` of the two windows, in ``[0, 2]`` with 0 a perfect match (so that lower is better, as
    ``Match.match_score`` expects).  2.0 when a window leaves the image or has no variance."""
    if image_a.shape != image_b.shape:
        raise ValueError("the images must have the same shape")
    score = float(_device_match.score_matrix(MATCH_NCC, image_a, image_b, [feature_a], [feature_b],
                                             window_size).cpu()[0, 0])
    tolerance = 1e-8
    assert -1.0 - tolerance <= 1.0 - score <= 1.0 + tolerance  # the reference's sanity check (ncc.py:47-48)
    return score


calculate_ncc._sfm_hip_metric = MATCH_NCC
