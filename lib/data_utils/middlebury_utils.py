"""``lib.data_utils.middlebury_utils`` drop-in (reference lib/data_utils/middlebury_utils.py)."""
from structure_from_motion_amd.data_utils.middlebury_utils import load_camera_k_r_t  # noqa: F401
