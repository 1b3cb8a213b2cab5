"""Harris corner detector with the reference's interface (reference ``lib/harris/harris_detector.py:11-113``).

Sobel filtering, the per-pixel second-moment / cornerness loop and the non-maximum suppression — the three
Python double loops of the reference — run as HIP kernels; the final ``np.argsort`` selection stays the same NumPy
call the reference makes (on the suppressed image copied back), so ties are ordered identically.

The reference's in-place, raster-order suppression (a neighbour visited earlier may already be zero) is kept
exactly (``nms_inplace_kernel``).  The cornerness uses the plain ``Ix2*Iy2 - IxIy^2`` where the reference calls
``np.linalg.det`` on the 2x2 matrix (LU + ``sign*exp(sum(log))``): values agree to ~1e-12 relative.
"""
from typing import List

import numpy as np
import torch

from .. import _native, device
from .._native import check
from ..common import correlate, feature

_sobel_x_kernel = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=float)


def detect_harris_corners(
    image: np.ndarray, num_corners: int = 50, block_size: int = 2, k: float = 0.04
) -> List[feature.Feature]:
    """The ``num_corners`` strongest Harris corners of a grayscale image, strongest first, as ``Feature(x, y)``
    at the centre of their ``block_size`` block.  Fewer are returned if fewer non-zero maxima exist."""
    if num_corners <= 0:
        raise ValueError("num_corners needs to be at least 1")
    suppressed = _suppressed_cornerness(image, block_size, k).cpu().numpy()
    # same NumPy selection as the reference (harris_detector.py:32-42): descending argsort, cut, drop zeros
    order = np.flip(np.argsort(suppressed, axis=None))[:num_corners]
    order = [i for i in order if suppressed[np.unravel_index(i, suppressed.shape)] != 0]
    y_indices, x_indices = np.unravel_index(order, suppressed.shape)
    ys = y_indices.astype(float) + float(block_size) / 2.0
    xs = x_indices.astype(float) + float(block_size) / 2.0
    return [feature.Feature(x=x, y=y) for y, x in zip(ys, xs)]


def _image_tensor(image: np.ndarray) -> torch.Tensor:
    if image.ndim != 2:
        raise ValueError("Only 2D single channel images are supported")
    device.require_gpu()
    return device.to_device(np.ascontiguousarray(image, dtype=np.float64))


def _cornerness_device(image_t: torch.Tensor, block_size: int, k: float, clamp: bool) -> torch.Tensor:
    lib = _native.load()
    h, w = image_t.shape
    sx = correlate.correlate_device(image_t, _sobel_x_kernel)
    sy = correlate.correlate_device(image_t, _sobel_x_kernel.transpose())
    shrink = int(np.around(block_size / 2))
    out = torch.empty((h - shrink, w - shrink), dtype=torch.float64, device=image_t.device)
    check(lib.sfm_harris_cornerness(sx.data_ptr(), sy.data_ptr(), h, w, int(block_size), float(k), int(clamp),
                                    out.shape[0], out.shape[1], out.data_ptr(), device._stream()),
          "sfm_harris_cornerness")
    return out


def _suppressed_cornerness(image: np.ndarray, block_size: int, k: float) -> torch.Tensor:
    lib = _native.load()
    corn = _cornerness_device(_image_tensor(image), block_size, k, clamp=True)
    check(lib.sfm_nms_inplace(corn.data_ptr(), corn.shape[0], corn.shape[1], device._stream()), "sfm_nms_inplace")
    return corn


def _calculate_cornerness_image(image: np.ndarray, block_size: int = 2, k: float = 0.04):
    """Harris response ``det(M) - k trace(M)^2`` of every ``block_size`` block (not clamped, not suppressed)."""
    return _cornerness_device(_image_tensor(image), block_size, k, clamp=False).cpu().numpy()


def _non_max_suppress(image: np.ndarray):
    """Zero, IN PLACE and in raster order, every pixel smaller than the maximum of its 3x3 neighbourhood."""
    lib = _native.load()
    t = _image_tensor(image)
    check(lib.sfm_nms_inplace(t.data_ptr(), t.shape[0], t.shape[1], device._stream()), "sfm_nms_inplace")
    image[...] = t.cpu().numpy()


def _apply_sobel_x(image: np.ndarray) -> np.ndarray:
    return correlate.cross_correlate(image, _sobel_x_kernel)


def _apply_sobel_y(image: np.ndarray) -> np.ndarray:
    return correlate.cross_correlate(image, _sobel_x_kernel.transpose())
