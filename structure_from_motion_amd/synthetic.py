"""Deterministic synthetic two-view correspondence sets (bench / demo inputs).

Host-side input generation only — nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np

BENCH_K = np.array([[1520.4, 0.0, 302.32], [0.0, 1525.9, 246.87], [0.0, 0.0, 1.0]])


def rotation_xy(deg_x: float, deg_y: float) -> np.ndarray:
    """Intrinsic rotation about X then Y (``Rotation.from_euler("XY", ...)``): Rx @ Ry."""
    ax, ay = np.radians(deg_x), np.radians(deg_y)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return rx @ ry


def two_view_scene(n: int, seed: int = 6, outlier_fraction: float = 0.3, noise_px: float = 0.5,
                   K: np.ndarray = BENCH_K):
    """Points uniform in x,y in [-1,1], z in [4,6]; camera 1 = [I|0]; camera 2 = euler XY (-5, -10) deg,
    t = (0.5, 0.05, 0.1); Gaussian pixel noise on every projection; a fraction of image-2 points replaced
    by uniform random pixels.  Returns (pix_a (n,2), pix_b (n,2), K, R, t, is_outlier)."""
    rng = np.random.default_rng(seed)
    X = np.empty((n, 3))
    X[:, 0] = rng.uniform(-1.0, 1.0, n)
    X[:, 1] = rng.uniform(-1.0, 1.0, n)
    X[:, 2] = rng.uniform(4.0, 6.0, n)
    R = rotation_xy(-5.0, -10.0)
    t = np.array([0.5, 0.05, 0.1])

    def project(Xc):
        uvw = Xc @ K.T
        return uvw[:, :2] / uvw[:, 2:3]

    pa = project(X) + rng.normal(0.0, noise_px, (n, 2))
    pb = project(X @ R.T + t) + rng.normal(0.0, noise_px, (n, 2))
    is_out = rng.random(n) < outlier_fraction
    width, height = 2.0 * K[0, 2], 2.0 * K[1, 2]
    rand_px = np.column_stack([rng.uniform(0, width, n), rng.uniform(0, height, n)])
    pb = np.where(is_out[:, None], rand_px, pb)
    return pa, pb, K, R, t, is_out
