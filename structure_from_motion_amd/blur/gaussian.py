"""Normalised Gaussian smoothing kernel factory (reference ``lib/blur/gaussian.py:4-25``); host-side helper."""
import numpy as np


def create_gaussian_kernel(kernel_size: int, sigma: float) -> np.ndarray:
    """A ``kernel_size`` x ``kernel_size`` Gaussian of standard deviation ``sigma`` that sums to one.
    ``kernel_size`` must be odd and at least 3."""
    if kernel_size <= 2:
        raise ValueError("kernel_size must be at least 3")
    if kernel_size % 2 == 0:
        raise ValueError("Only odd-sized kernels are accepted")
    half = int(kernel_size / 2)
    offsets = np.arange(-half, half + 1)
    x_grid, y_grid = np.meshgrid(offsets, offsets)
    kernel = np.exp(-(x_grid ** 2 + y_grid ** 2) / (2 * sigma ** 2))
    kernel /= 2 * np.pi * sigma ** 2
    kernel /= np.sum(kernel)
    return kernel
