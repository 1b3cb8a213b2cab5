"""``lib.common.correlate`` drop-in (reference lib/common/correlate.py)."""
from structure_from_motion_amd.common.correlate import cross_correlate  # noqa: F401
