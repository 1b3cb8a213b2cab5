"""Mean squared difference window score (reference ``lib/feature_matching/ssd.py:7-36``).

Deviation, documented: the reference subtracts and squares in the image dtype, so on ``uint8`` images its
result wraps modulo 256 (ssd.py:31-35).  Here images are widened to float64 first — identical to the reference
for float or wide-integer images (the domain of its own ``test_ssd.py``); ``apps/sfm.py`` does not use SSD.
"""
import numpy as np

from ..common import feature as feat
from . import _device_match
from .._native import MATCH_SSD


def calculate_ssd(
    image_a: np.ndarray,
    image_b: np.ndarray,
    feature_a: feat.Feature,
    feature_b: feat.Feature,
    window_size: int = 5,
) -> float:
    """Sum of squared differences of the two windows divided by the window size; ``inf`` when a window
    leaves the image."""
    if image_a.shape != image_b.shape:
        raise ValueError("the images must have the same shape")
    return float(_device_match.score_matrix(MATCH_SSD, image_a, image_b, [feature_a], [feature_b],
                                            window_size).cpu()[0, 0])


calculate_ssd._sfm_hip_metric = MATCH_SSD
