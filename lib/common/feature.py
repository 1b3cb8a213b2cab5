"""``lib.common.feature`` drop-in (reference lib/common/feature.py)."""
from structure_from_motion_amd.common.feature import Feature  # noqa: F401
