"""``lib.ransac.ransac`` drop-in (reference lib/ransac/ransac.py)."""
from structure_from_motion_amd.ransac.ransac import (  # noqa: F401
    ErrorAggregationMethod,
    _aggregate_error,
    fit_with_ransac,
)
