"""``lib.feature_matching.util`` drop-in (reference lib/feature_matching/util.py)."""
from structure_from_motion_amd.feature_matching.util import is_within_bounds, select_window  # noqa: F401
