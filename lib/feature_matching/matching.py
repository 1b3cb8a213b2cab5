"""``lib.feature_matching.matching`` drop-in: only the ``Match`` value type is on the hot path."""
from structure_from_motion_amd.feature_matching.matching import Match  # noqa: F401
