"""Image feature value type.

Boundary type of the epipolar hot path: mirrors the reference's
``lib/common/feature.py:4-7`` (a mutable two-field dataclass ``Feature(x, y)``).
"""
from dataclasses import dataclass


@dataclass
class Feature:
    x: float
    y: float
