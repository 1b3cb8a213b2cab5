"""``lib.harris.harris_detector`` drop-in (reference lib/harris/harris_detector.py)."""
from structure_from_motion_amd.harris.harris_detector import (  # noqa: F401
    _apply_sobel_x,
    _apply_sobel_y,
    _calculate_cornerness_image,
    _non_max_suppress,
    detect_harris_corners,
)
