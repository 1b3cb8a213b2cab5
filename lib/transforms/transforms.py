"""``lib.transforms.transforms`` drop-in (reference lib/transforms/transforms.py)."""
from structure_from_motion_amd.transforms.transforms import Transform3D  # noqa: F401
