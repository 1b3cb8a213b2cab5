"""Window helpers of the matcher (reference ``lib/feature_matching/util.py:8-27``); pure host predicates —
the batched forms live in ``patch_extract_kernel``."""
from typing import Tuple

import numpy as np

from ..common import feature as feat


def is_within_bounds(feature: feat.Feature, image_shape: Tuple[int, int], window_size: int) -> bool:
    """Whether a ``window_size`` window centred on the feature lies inside an image of ``image_shape``."""
    half = int(window_size / 2)
    rows_ok = half <= feature.y < (image_shape[0] - half)
    cols_ok = half <= feature.x < (image_shape[1] - half)
    return bool(rows_ok and cols_ok)


def select_window(image: np.ndarray, feature: feat.Feature, window_size: int) -> np.ndarray:
    """The ``(2*half+1)``-square view of ``image`` centred on ``(int(feature.y), int(feature.x))``."""
    half = int(window_size / 2)
    row, col = int(feature.y), int(feature.x)
    return image[row - half:row + half + 1, col - half:col + half + 1]
