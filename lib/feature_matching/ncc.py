"""``lib.feature_matching.ncc`` drop-in (reference lib/feature_matching/ncc.py)."""
from structure_from_motion_amd.feature_matching.ncc import calculate_ncc  # noqa: F401
