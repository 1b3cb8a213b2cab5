"""``lib.blur.gaussian`` drop-in (reference lib/blur/gaussian.py)."""
from structure_from_motion_amd.blur.gaussian import create_gaussian_kernel  # noqa: F401
