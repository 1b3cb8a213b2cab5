"""``lib.epipolar.sed`` drop-in (reference lib/epipolar/sed.py)."""
from structure_from_motion_amd.epipolar.sed import calculate_symmetric_epipolar_distance  # noqa: F401
