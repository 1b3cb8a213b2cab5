"""``Match`` value type consumed by the epipolar hot path.

Only the dataclass is provided (reference ``lib/feature_matching/matching.py:15-24``);
the brute-force matcher that produces matches is outside the hot-path scope.
"""
from dataclasses import dataclass
from math import inf


@dataclass
class Match:
    a_index: int = -1
    b_index: int = -1
    # Lower is better for every score function.
    match_score: float = inf

    def __lt__(self, other) -> bool:
        return self.match_score < other.match_score
