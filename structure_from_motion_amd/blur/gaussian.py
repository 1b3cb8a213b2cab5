"""Normalised Gaussian smoothing kernel (API of reference ``lib/blur/gaussian.py:4-25``; host-side utility, used by
nothing on the hot path).  A 2-D Gaussian is separable, so the kernel is the outer product of one 1-D profile with itself,
normalised to unit sum — the constant 1 / (2 pi sigma^2) of the analytic form cancels in that normalisation and is not
applied."""
import numpy as np


def create_gaussian_kernel(kernel_size: int, sigma: float) -> np.ndarray:
    """``kernel_size`` x ``kernel_size`` float64 weights, centred, mirror-symmetric in both axes, summing to one.
    ``kernel_size`` has to be odd and >= 3 (``ValueError`` otherwise, as in the reference)."""
    if kernel_size < 3 or kernel_size % 2 != 1:
        raise ValueError(f"a Gaussian kernel needs an odd size of at least 3, got {kernel_size}")
    radius = kernel_size // 2
    distance = np.abs(np.arange(kernel_size) - radius).astype(np.float64)   # |offset| from the centre tap: exactly symmetric
    profile = np.exp(-0.5 * (distance / sigma) ** 2)
    weights = np.outer(profile, profile)
    return weights / weights.sum()
