"""Device side of the matcher: score matrices and heap summaries through the C ABI (host glue only)."""
from __future__ import annotations

import operator
from typing import Tuple

import numpy as np
import torch

from .. import _native, device
from .._native import MATCH_NCC, MATCH_SSD, PATCH_MEAN_REMOVED, PATCH_PLAIN, PATCH_RAW64, check

F64 = torch.float64
_feature_x, _feature_y = operator.attrgetter("x"), operator.attrgetter("y")


def _features_tensor(features) -> torch.Tensor:
    n = len(features)
    arr = np.empty((n, 2), dtype=np.float64)   # (two C-level passes over the list: 1.7 ms for 20 000 features, a Python loop took 5.8)
    arr[:, 0] = np.fromiter(map(_feature_x, features), dtype=np.float64, count=n)
    arr[:, 1] = np.fromiter(map(_feature_y, features), dtype=np.float64, count=n)
    return device.to_device(arr)


_NARROW_FLOAT = (np.uint8, np.int8, np.int16, np.int32, np.float32)   # every value is a float64: widened on the device
_NARROW_INT = (np.uint8, np.int8, np.int16, np.int32)                  # ... and an int64 (torch has no uint16 / uint32 tensors)


def _image_tensor(image: np.ndarray, integer_dtype=None) -> torch.Tensor:
    """The image on the device as the score kernels read it: float64, or — integer SSD — int64 bit patterns.  A narrow image
    travels in its own dtype and is widened there (a VGA uint8 frame: 0.3 MB instead of 2.4 MB over PCIe)."""
    if image.ndim != 2:
        raise ValueError("the matcher works on single-channel (2-D) images")
    dev = device.require_gpu()
    if integer_dtype is None:
        if image.dtype.type in _NARROW_FLOAT and image.dtype.isnative:
            return torch.as_tensor(np.ascontiguousarray(image)).to(dev).to(torch.float64)
        return device.to_device(np.ascontiguousarray(image, dtype=np.float64))
    # integer SSD: the pixels in the common dtype NumPy would promote both images to (value-preserving), then as int64 —
    # the two's-complement bit pattern for uint64, whose modular arithmetic is the same
    if image.dtype.type in _NARROW_INT and image.dtype.isnative and np.can_cast(image.dtype, integer_dtype, casting="safe"):
        return torch.as_tensor(np.ascontiguousarray(image)).to(dev).to(torch.int64)   # value-preserving both ways
    wide = np.ascontiguousarray(image).astype(integer_dtype, copy=False)
    wide = wide.view(np.int64) if wide.dtype == np.uint64 else wide.astype(np.int64)
    return device.to_device(np.ascontiguousarray(wide), dtype=torch.int64)


def ssd_arithmetic(dtype_a, dtype_b):
    """The arithmetic the reference's ``calculate_ssd`` runs in for images of these dtypes (``ssd.py:31-35``: ``window_a -
    window_b`` and ``np.square`` stay in NumPy's result type of the two): ``(integer dtype, device metric code)`` for an
    integer result type — differences and squares then wrap modulo 2**bits —, ``(None, MATCH_SSD)`` for floating point
    (computed in float64 here).  ``bool`` images raise the ``TypeError`` NumPy raises for ``bool - bool``."""
    rt = np.result_type(dtype_a, dtype_b)
    if rt == np.bool_:
        raise TypeError("numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, "
                        "or the logical_xor function instead.")
    if rt.kind in "iu":
        return rt, _native.match_ssd_int(8 * rt.itemsize, rt.kind == "i")
    return None, MATCH_SSD


def _extract_patches(metric: int, image_a, image_b, feats_a, feats_b, window_size: int):
    """Window patches of both feature sets on the device: ((patches, ssq, ok, n) for a, same for b, K, device metric).
    ``metric`` MATCH_SSD on integer images becomes the integer-SSD code of their common dtype (``ssd_arithmetic``)."""
    if image_a.shape != image_b.shape:
        raise ValueError("the images must have the same shape")
    integer_dtype = None
    if metric == MATCH_SSD:
        integer_dtype, metric = ssd_arithmetic(np.asarray(image_a).dtype, np.asarray(image_b).dtype)
    patch_mode = PATCH_MEAN_REMOVED if metric == MATCH_NCC else (PATCH_RAW64 if integer_dtype is not None else PATCH_PLAIN)
    lib = _native.load()
    dev = device.require_gpu()
    st = device._stream()
    side = 2 * int(window_size / 2) + 1
    K = side * side
    out = []
    for image, feats in ((image_a, feats_a), (image_b, feats_b)):
        img = _image_tensor(np.asarray(image), integer_dtype)
        ft = feats if isinstance(feats, torch.Tensor) else _features_tensor(feats)
        n = ft.shape[0]
        # rows padded to whole 128-feature tiles: the score kernels then stage them by LDS-DMA (sfm_match.hip)
        patches = torch.empty((K, max(-(-n // 128) * 128, 128)), dtype=F64, device=dev)
        ssq = torch.empty((max(n, 1),), dtype=F64, device=dev)
        ok = torch.empty((max(n, 1),), dtype=torch.uint8, device=dev)
        check(lib.sfm_patch_extract(img.data_ptr(), img.shape[0], img.shape[1], ft.data_ptr(), n, int(window_size),
                                    patch_mode, patches.shape[1], patches.data_ptr(),
                                    ssq.data_ptr(), ok.data_ptr(), st), "sfm_patch_extract")
        out.append((patches, ssq, ok, n))
    return out[0], out[1], K, metric


def score_matrix(metric: int, image_a, image_b, feats_a, feats_b, window_size: int) -> torch.Tensor:
    """(nA, nB) device tensor of window scores for every feature pair."""
    (pa, qa, oka, nA), (pb, qb, okb, nB), K, metric = _extract_patches(metric, image_a, image_b, feats_a, feats_b, window_size)
    lib = _native.load()
    st = device._stream()
    scores = torch.empty((nA, nB), dtype=F64, device=pa.device)
    check(lib.sfm_pair_scores(metric, pa.data_ptr(), pa.shape[1], pb.data_ptr(), pb.shape[1], qa.data_ptr(),
                              qb.data_ptr(), oka.data_ptr(), okb.data_ptr(), nA, nB, K, scores.data_ptr(), st),
          "sfm_pair_scores")
    return scores


def row_summary(scores: torch.Tensor) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """heap[0] score / b index and heap[1] score per row (host arrays)."""
    lib = _native.load()
    nA, nB = scores.shape
    out, best, arg, second = _summary_buffers(nA, scores.device)
    check(lib.sfm_match_row_summary(scores.data_ptr(), nA, nB, best.data_ptr(), arg.data_ptr(),
                                    second.data_ptr(), device._stream()), "sfm_match_row_summary")
    return _summary_to_host(out, nA)


def _summary_buffers(nA: int, dev):
    """best | second | arg of the nA rows in ONE device buffer (read back with one copy): (buffer, best, arg, second)."""
    out = torch.empty((2 * nA + (nA + 1) // 2 + 1,), dtype=F64, device=dev)
    return out, out[:nA], out[2 * nA:].view(torch.int32)[:nA], out[nA:2 * nA]


def _summary_to_host(out: torch.Tensor, nA: int):
    host = out.cpu().numpy()
    return host[:nA], host[2 * nA:].view(np.int32)[:nA].astype(np.int64), host[nA:2 * nA]


def match_summary(metric: int, image_a, image_b, feats_a, feats_b, window_size: int):
    """heap[0] score / b index and heap[1] score per A-feature straight from the images, without materialising
    the score matrix (``sfm_match_summary``); bit-identical to ``row_summary(score_matrix(...))``."""
    (pa, qa, oka, nA), (pb, qb, okb, nB), K, metric = _extract_patches(metric, image_a, image_b, feats_a, feats_b, window_size)
    lib = _native.load()
    dev = pa.device
    out, best, arg, second = _summary_buffers(nA, dev)
    if nA and nB:
        ws_bytes = int(lib.sfm_match_summary_workspace_bytes(nA, nB))
        ws = torch.empty((ws_bytes // 8,), dtype=F64, device=dev)
        check(lib.sfm_match_summary(metric, pa.data_ptr(), pa.shape[1], pb.data_ptr(), pb.shape[1], qa.data_ptr(),
                                    qb.data_ptr(), oka.data_ptr(), okb.data_ptr(), nA, nB, K, ws.data_ptr(), ws_bytes,
                                    best.data_ptr(), arg.data_ptr(), second.data_ptr(), device._stream()),
              "sfm_match_summary")
    return _summary_to_host(out, nA)
