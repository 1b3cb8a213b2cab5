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
    suppressed = _suppressed_cornerness(image, block_size, k)
    order = _strongest_indices_device(suppressed, num_corners)
    y_indices, x_indices = np.unravel_index(order, tuple(suppressed.shape))
    ys = y_indices.astype(float) + float(block_size) / 2.0
    xs = x_indices.astype(float) + float(block_size) / 2.0
    return [feature.Feature(x=x, y=y) for y, x in zip(ys, xs)]


def _strongest_indices(suppressed: np.ndarray, num_corners: int) -> np.ndarray:
    """Flat indices the reference selects (harris_detector.py:32-42): ``np.flip(np.argsort(all pixels))``, cut to
    ``num_corners``, zeros dropped.  After suppression almost every pixel is zero, so only the non-zero pixels
    are sorted; when two of the values that matter are exactly equal their order depends on NumPy's (unstable)
    sort of the full array, and the reference's literal expression is evaluated instead."""
    flat = suppressed.ravel()
    nonzero = np.flatnonzero(flat)
    if len(nonzero) == 0:
        return np.zeros(0, dtype=np.int64)
    values = flat[nonzero]
    by_value = np.argsort(-values, kind="stable")
    keep = min(num_corners, len(nonzero))
    head = values[by_value[:min(keep + 1, len(nonzero))]]
    ties = bool(np.any(head[1:] == head[:-1]))
    if ties:
        order = np.flip(np.argsort(suppressed, axis=None))[:num_corners]
        return np.array([i for i in order if flat[i] != 0], dtype=np.int64)
    return nonzero[by_value[:keep]].astype(np.int64)


_COMPACT_CAPACITY = 1 << 20   # (index, value) slots of the device compaction: 12 MB; integer-valued 1080p images leave ~150 000 maxima
_PRUNE_WORDS = 8192           # the two digit histograms of sfm_prune_top


def _strongest_indices_device(suppressed: torch.Tensor, num_corners: int) -> np.ndarray:
    """``_strongest_indices`` without copying the image back: the non-zero pixels are compacted on the device
    (``sfm_compact_nonzero``), pruned there to the ones that can be among the ``num_corners + 1`` largest
    (``sfm_prune_top``: everything at or above that value, ties included) and only they travel, with the two counters, in ONE
    read-back.  Falls back to the full image when exact ties among the values that matter make the reference's
    order depend on NumPy's sort of the whole array, when a NaN survives, or when more than ``_COMPACT_CAPACITY`` pixels do."""
    lib = _native.load()
    count = suppressed.numel()
    capacity = min(count, _COMPACT_CAPACITY)
    dev = suppressed.device
    index = torch.empty((max(capacity, 1),), dtype=torch.int32, device=dev)
    value = torch.empty((max(capacity, 1),), dtype=torch.float64, device=dev)
    # one buffer: [found, kept, -, -][room values (f64)][room indices][histograms]; its head is what travels
    room = max(2048, 2 * (num_corners + 1))
    words = 4 + 3 * room
    buf = torch.empty((words + _PRUNE_WORDS,), dtype=torch.int32, device=dev)
    check(lib.sfm_compact_nonzero(suppressed.data_ptr(), count, capacity, buf[0:].data_ptr(), index.data_ptr(),
                                  value.data_ptr(), device._stream()), "sfm_compact_nonzero")
    check(lib.sfm_prune_top(value.data_ptr(), index.data_ptr(), buf[0:].data_ptr(), capacity, num_corners + 1,
                            buf[words:].data_ptr(), room, buf[1:].data_ptr(), buf[4 + 2 * room:].data_ptr(),
                            buf[4:].data_ptr(), device._stream()), "sfm_prune_top")
    host = buf[:words].cpu().numpy()
    found, kept = int(host[0]), int(host[1])
    if found == 0:
        return np.zeros(0, dtype=np.int64)
    if found > capacity:
        return _strongest_indices(suppressed.cpu().numpy(), num_corners)
    if kept <= room:   # the leaders (and whatever lies within 2^-12 of the last one): all that the selection below can pick from
        values = host[4:4 + 2 * room].view(np.float64)[:kept]
        nonzero = host[4 + 2 * room:4 + 2 * room + kept].astype(np.int64)  # slot order is arbitrary (atomics)
    else:              # (more than the buffer next to the last leader: e.g. an image whose maxima are all equal)
        nonzero = index[:found].cpu().numpy().astype(np.int64)
        values = value[:found].cpu().numpy()
    if bool(np.isnan(values).any()):
        return _strongest_indices(suppressed.cpu().numpy(), num_corners)
    # the keep + 1 largest, ordered by (value descending, flat index ascending) — what a stable sort of the
    # raster-ordered non-zeros gives; a selection (O(n)) instead of a full sort, then a sort of those few
    have = len(values)
    keep = min(num_corners, found)
    m = min(keep + 1, have)
    cand = np.argpartition(-values, m - 1)[:m] if m < have else np.arange(have)
    order = cand[np.lexsort((nonzero[cand], -values[cand]))]
    head = values[order]
    if bool(np.any(head[1:] == head[:-1])):
        return _strongest_indices(suppressed.cpu().numpy(), num_corners)
    return nonzero[order[:keep]]


_NARROW = (np.uint8, np.int8, np.int16, np.int32, np.float32)   # every value is a float64: widened on the device, bit for bit what NumPy's astype gives


def _image_tensor(image: np.ndarray) -> torch.Tensor:
    """The image as float64 on the device.  An 8 / 16 / 32-bit image travels in its own dtype (a 1080p uint8 frame: 2 MB instead
    of 16.6 MB over PCIe, a third of the whole call) and is widened there."""
    if image.ndim != 2:
        raise ValueError("Only 2D single channel images are supported")
    dev = device.require_gpu()
    image = np.asarray(image)
    if image.dtype.type in _NARROW and image.dtype.isnative:
        return torch.as_tensor(np.ascontiguousarray(image)).to(dev).to(torch.float64)
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


_ROUNDS_PER_CHECK = 12
_MAX_ROUNDS = 96


def _nms_device(t: torch.Tensor) -> None:
    """In-place raster-order suppression of a device image.  Fast path: parallel fixpoint rounds
    (``nms_round_kernel``), checked every few rounds; images with very long dependency chains fall back to the
    single-block wavefront sweep (``nms_inplace_kernel``).  Both reproduce the reference's sequential result."""
    lib = _native.load()
    h, w = t.shape
    if h == 0 or w == 0:
        return
    st = device._stream()
    state = torch.zeros((h, w), dtype=torch.uint8, device=t.device)
    left = torch.zeros((_MAX_ROUNDS,), dtype=torch.int32, device=t.device)
    done = 0
    while done < _MAX_ROUNDS:
        for i in range(done, done + _ROUNDS_PER_CHECK):
            check(lib.sfm_nms_round(t.data_ptr(), state.data_ptr(), h, w, left[i:].data_ptr(), st), "sfm_nms_round")
        done += _ROUNDS_PER_CHECK
        if int(left[done - 1].cpu()) == 0:
            check(lib.sfm_nms_finalize(t.data_ptr(), state.data_ptr(), h, w, st), "sfm_nms_finalize")
            return
    check(lib.sfm_nms_inplace(t.data_ptr(), h, w, st), "sfm_nms_inplace")  # `t` is still untouched here


def _suppressed_cornerness(image: np.ndarray, block_size: int, k: float) -> torch.Tensor:
    corn = _cornerness_device(_image_tensor(image), block_size, k, clamp=True)
    _nms_device(corn)
    return corn


def _calculate_cornerness_image(image: np.ndarray, block_size: int = 2, k: float = 0.04):
    """Harris response ``det(M) - k trace(M)^2`` of every ``block_size`` block (not clamped, not suppressed)."""
    return _cornerness_device(_image_tensor(image), block_size, k, clamp=False).cpu().numpy()


def _non_max_suppress(image: np.ndarray):
    """Zero, IN PLACE and in raster order, every pixel smaller than the maximum of its 3x3 neighbourhood."""
    t = _image_tensor(image)
    _nms_device(t)
    image[...] = t.cpu().numpy()


def _apply_sobel_x(image: np.ndarray) -> np.ndarray:
    return correlate.cross_correlate(image, _sobel_x_kernel)


def _apply_sobel_y(image: np.ndarray) -> np.ndarray:
    return correlate.cross_correlate(image, _sobel_x_kernel.transpose())
