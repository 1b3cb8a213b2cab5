"""``lib.epipolar.epipolar_ransac`` drop-in (reference lib/epipolar/epipolar_ransac.py)."""
from structure_from_motion_amd.epipolar.epipolar_ransac import (  # noqa: F401
    FeaturePair,
    calculate_sed_inlier_score,
    eight_point_model_fitter,
    estimate_essential_mat_with_ransac,
)
