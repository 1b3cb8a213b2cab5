"""Mean squared difference window score (reference ``lib/feature_matching/ssd.py:7-36``).

The reference subtracts and squares in the image dtype (ssd.py:31-35): on integer images both steps wrap modulo
2**bits — modulo 256 on the ``uint8`` images ``apps/sfm.py:222-224`` produces —, the sum runs in int64 / uint64 and the
division by the window size in float64.  The device kernels do exactly that (``SFM_MATCH_SSD_INT``; golden G14: scores
and match lists of the real reference on uint8, and every integer dtype, bit for bit).  Floating-point images are
evaluated in float64 (float16 / float32 images agree with the reference to their own rounding); ``bool`` images raise
NumPy's ``TypeError`` as in the reference.
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
