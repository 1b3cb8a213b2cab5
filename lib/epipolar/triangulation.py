"""``lib.epipolar.triangulation`` drop-in (reference lib/epipolar/triangulation.py)."""
from structure_from_motion_amd.epipolar.triangulation import (  # noqa: F401
    triangulate_point_correspondence,
    triangulate_points,
)
