"""``lib.feature_matching.matching`` drop-in (reference lib/feature_matching/matching.py)."""
from structure_from_motion_amd.feature_matching.matching import (  # noqa: F401
    ImagePairScore,
    Match,
    ScoreFunction,
    ValidationStrategy,
    match_brute_force,
)
