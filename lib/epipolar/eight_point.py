"""``lib.epipolar.eight_point`` drop-in (reference lib/epipolar/eight_point.py)."""
from structure_from_motion_amd.epipolar.eight_point import (  # noqa: F401
    EightPointCalculationError,
    _cheirality_check,
    _get_matching_coordinates,
    _recover_all_r_t,
    _recover_r_t,
    create_trivial_matches,
    estimate_essential_mat,
    estimate_fundamental_mat,
    estimate_r_t,
    recover_r_t_from_e,
    to_normalized_image_coords,
)
