"""CPU ORACLE — test infrastructure, NOT product code.

numpy restatement of the reference's Harris corner detector: ``lib/harris/harris_detector.py:11-113`` with
``lib/common/correlate.py:4-39`` as the Sobel filter.  Pinned by ``tests/golden/g12_harris.npz``.

Conventions fixed here: 3x3 correlation sums run row-major left to right; the 2x2 determinant is the plain
``Ix2*Iy2 - IxIy*IxIy`` (the reference goes through ``np.linalg.det`` = LU + sign*exp(sum log), which differs in
the last bits and cannot be reproduced off-CPU); block sums run row-major.
"""
from __future__ import annotations

import numpy as np

SOBEL_X = np.array([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])


def cross_correlate(image: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """correlate.py:4-39: zero 'same' padding, odd square kernels; the border of half a kernel stays 0."""
    if len(image.shape) != 2 or len(kernel.shape) != 2:
        raise ValueError("Only 2D single channel images are supported")
    if kernel.shape[0] != kernel.shape[1] or (kernel.shape[0] % 2) == 0:
        raise ValueError("Only odd-sized square kernels are supported")
    ks = kernel.shape[0]
    h, w = image.shape
    if h < ks or w < ks:
        raise ValueError("Kernel cannot be larger than image")
    half = ks // 2
    img = image.astype(np.float64)
    out = np.zeros(image.shape, dtype=np.float64)
    acc = np.zeros((h - ks + 1, w - ks + 1))
    for r in range(ks):
        for c in range(ks):
            acc = acc + img[r:r + h - ks + 1, c:c + w - ks + 1] * kernel[r, c]
    out[half:half + h - ks + 1, half:half + w - ks + 1] = acc
    return out


def cornerness_image(image: np.ndarray, block_size: int = 2, k: float = 0.04) -> np.ndarray:
    """harris_detector.py:57-86."""
    sx = cross_correlate(image, SOBEL_X)
    sy = cross_correlate(image, SOBEL_X.T)
    ix2, iy2, ixy = sx * sx, sy * sy, sx * sy
    h, w = image.shape
    shrink = int(np.around(block_size / 2))
    out = np.zeros((h - shrink, w - shrink))
    rows, cols = h - block_size, w - block_size
    if rows <= 0 or cols <= 0:
        return out

    def block_sum(m):
        acc = np.zeros((rows, cols))
        for r in range(block_size):
            for c in range(block_size):
                acc = acc + m[r:r + rows, c:c + cols]
        return acc

    a, b, d = block_sum(ix2), block_sum(ixy), block_sum(iy2)
    trace = a + d
    out[:rows, :cols] = (a * d - b * b) - k * (trace * trace)
    return out


def non_max_suppress(image: np.ndarray) -> None:
    """harris_detector.py:95-105, IN PLACE and in raster order: neighbours visited earlier may already be zero."""
    h, w = image.shape
    for r in range(h):
        for c in range(w):
            window = image[max(0, r - 1):min(h, r + 2), max(0, c - 1):min(w, c + 2)]
            if image[r, c] < np.amax(window):
                image[r, c] = 0.0


def detect_harris_corners(image: np.ndarray, num_corners: int = 50, block_size: int = 2, k: float = 0.04):
    """harris_detector.py:11-54 -> (n,2) array of (x, y) in the reference's output order, and the suppressed
    cornerness image."""
    if num_corners <= 0:
        raise ValueError("num_corners needs to be at least 1")
    corn = cornerness_image(image, block_size, k)
    corn[corn < 0] = 0.0
    non_max_suppress(corn)
    order = np.flip(np.argsort(corn, axis=None))[:num_corners]
    order = np.array([i for i in order if corn[np.unravel_index(i, corn.shape)] != 0], dtype=np.int64)
    ys, xs = np.unravel_index(order, corn.shape) if len(order) else (np.zeros(0, dtype=np.int64),) * 2
    pts = np.column_stack([xs.astype(float) + block_size / 2.0, ys.astype(float) + block_size / 2.0])
    return pts.reshape(-1, 2), corn
