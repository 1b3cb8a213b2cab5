"""CPU ORACLE — test infrastructure, NOT product code.

numpy restatement of the reference's brute-force feature matcher: window scores (NCC / SSD) for every
pair of features and the heap / ratio-test / cross-check logic of ``match_brute_force``.
Reference: ``lib/feature_matching/matching.py:36-118``, ``ncc.py:7-54``, ``ssd.py:7-36``, ``util.py:8-27``.
Pinned by ``tests/golden/g11_matching.npz`` (outputs of the real reference).

Floating-point conventions fixed here (the reference leaves them to NumPy's pairwise sums and BLAS dot):
window sums run sequentially in row-major window order with separate multiply/add roundings.
"""
from __future__ import annotations

import heapq
from typing import List, Optional, Sequence, Tuple

import numpy as np

NCC, SSD = "ncc", "ssd"
RATIO_TEST, CROSSCHECK = "ratio_test", "crosscheck"


def half_window(window_size: int) -> int:
    return int(window_size / 2)  # util.py:11,25


def within_bounds(features: np.ndarray, image_shape, window_size: int) -> np.ndarray:
    """util.py:8-18 for (n,2) features [x, y]: half <= y < H - half and half <= x < W - half."""
    h = half_window(window_size)
    x, y = features[:, 0], features[:, 1]
    return (h <= y) & (y < image_shape[0] - h) & (h <= x) & (x < image_shape[1] - h)


def windows(image: np.ndarray, features: np.ndarray, window_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """util.py:21-27: image[int(y)-h : int(y)+h+1, int(x)-h : int(x)+h+1] flattened row-major, as float64.
    Returns (patches (n, (2h+1)^2), in-bounds flags); out-of-bounds rows are zero."""
    h = half_window(window_size)
    side = 2 * h + 1
    ok = within_bounds(features, image.shape, window_size)
    out = np.zeros((len(features), side * side), dtype=np.float64)
    for i, (x, y) in enumerate(features):
        if ok[i]:
            out[i] = image[int(y) - h:int(y) + h + 1, int(x) - h:int(x) + h + 1].astype(np.float64).ravel()
    return out, ok


def seq_sum(a: np.ndarray) -> np.ndarray:
    """Sequential (left-to-right) sum along the last axis."""
    return np.add.accumulate(a, axis=-1)[..., -1]


def shifted_patches(patches: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """ncc.py:33-37: window - mean(window); plus sum of squares (ncc.py:41)."""
    mean = seq_sum(patches) / patches.shape[-1]
    shifted = patches - mean[..., None]
    return shifted, seq_sum(shifted * shifted)


def ncc_scores(image_a, image_b, feats_a, feats_b, window_size: int = 3) -> np.ndarray:
    """ncc.py:7-54 for every pair -> (nA, nB) scores in [0, 2]; 2.0 when a window leaves the image or the
    denominator is zero."""
    if image_a.shape != image_b.shape:
        raise ValueError("the images must have the same shape")
    pa, oka = windows(image_a, np.asarray(feats_a, dtype=np.float64), window_size)
    pb, okb = windows(image_b, np.asarray(feats_b, dtype=np.float64), window_size)
    sa, qa = shifted_patches(pa)
    sb, qb = shifted_patches(pb)
    num = np.zeros((len(pa), len(pb)))
    for k in range(sa.shape[1]):  # sequential over the window, like a scalar dot product
        num = num + sa[:, k][:, None] * sb[:, k][None, :]
    den = np.sqrt(qa[:, None] * qb[None, :])
    with np.errstate(divide="ignore", invalid="ignore"):
        score = (num / den) * -1.0 + 1.0
    bad = (~oka)[:, None] | (~okb)[None, :] | (den == 0)
    return np.where(bad, 2.0, score)


def ssd_result_kind(dtype_a, dtype_b):
    """What ssd.py:31-35 computes in, given the two image dtypes: NumPy promotes the operands of ``window_a - window_b``
    and every later step stays in that type.  Returns ("int", bits, signed) for an integer result type — difference and
    square then wrap modulo 2**bits (into the signed range for signed types) —, ("float",) otherwise; bool raises the
    TypeError NumPy raises for ``bool - bool``."""
    rt = np.result_type(dtype_a, dtype_b)
    if rt == np.bool_:
        raise TypeError("numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, "
                        "or the logical_xor function instead.")
    if rt.kind in "iu":
        return ("int", 8 * rt.itemsize, rt.kind == "i")
    return ("float",)


def _wrap(values, bits: int, signed: bool):
    """Python-int array reduced modulo 2**bits into the dtype's range (object arrays: exact integers of any size)."""
    m = 1 << bits
    v = values % m
    if signed:
        v = np.where(v >= (m >> 1), v - m, v)
    return v


def ssd_scores(image_a, image_b, feats_a, feats_b, window_size: int = 5) -> np.ndarray:
    """ssd.py:7-36 for every pair: mean squared difference; +inf when a window leaves the image.

    Integer images (ssd.py:31-35 run in the image dtype): ``diff`` wraps modulo 2**bits, ``np.square(diff)`` wraps again,
    ``np.sum`` accumulates in int64 / uint64 (wrapping modulo 2**64 only for 64-bit types) and ``/ size`` converts that
    integer to float64 and divides.  Restated here with exact Python integers and explicit reductions."""
    if image_a.shape != image_b.shape:
        raise ValueError("the images must have the same shape")
    image_a, image_b = np.asarray(image_a), np.asarray(image_b)
    kind = ssd_result_kind(image_a.dtype, image_b.dtype)
    fa, fb = np.asarray(feats_a, dtype=np.float64), np.asarray(feats_b, dtype=np.float64)
    if kind[0] == "float":
        pa, oka = windows(image_a, fa, window_size)
        pb, okb = windows(image_b, fb, window_size)
        acc = np.zeros((len(pa), len(pb)))
        for k in range(pa.shape[1]):
            d = pa[:, k][:, None] - pb[:, k][None, :]
            acc = acc + d * d
        score = acc / pa.shape[1]
        bad = (~oka)[:, None] | (~okb)[None, :]
        return np.where(bad, np.inf, score)
    _, bits, signed = kind
    h = half_window(window_size)
    side = 2 * h + 1
    oka, okb = within_bounds(fa, image_a.shape, window_size), within_bounds(fb, image_b.shape, window_size)

    def int_windows(image, feats, ok):
        out = np.zeros((len(feats), side * side), dtype=object)
        for i, (x, y) in enumerate(feats):
            if ok[i]:
                out[i] = [int(v) for v in image[int(y) - h:int(y) + h + 1, int(x) - h:int(x) + h + 1].ravel()]
        return out

    pa, pb = int_windows(image_a, fa, oka), int_windows(image_b, fb, okb)
    total = np.zeros((len(pa), len(pb)), dtype=object)
    for k in range(side * side):
        d = _wrap(pa[:, k][:, None] - pb[:, k][None, :], bits, signed)
        total = total + _wrap(d * d, bits, signed)
    total = _wrap(total, 64, signed)   # the int64 / uint64 accumulator of np.sum
    score = np.array([[float(v) for v in row] for row in total], dtype=np.float64).reshape(total.shape) / float(side * side)
    bad = (~oka)[:, None] | (~okb)[None, :]
    return np.where(bad, np.inf, score)


def ssd_scores_in_dtype(image_a, image_b, feats_a, feats_b, window_size: int = 5) -> np.ndarray:
    """The same scores for INTEGER images by the other route: NumPy's own fixed-width arithmetic in the promoted dtype,
    vectorised over all pairs (subtract, square — both wrap silently —, sum in the dtype's 64-bit accumulator, float64
    division).  Fast enough for the large GPU parity cases; test_matching_oracle.py checks it against ``ssd_scores`` and G14."""
    image_a, image_b = np.asarray(image_a), np.asarray(image_b)
    kind = ssd_result_kind(image_a.dtype, image_b.dtype)
    assert kind[0] == "int", "integer images only"
    rt = np.result_type(image_a.dtype, image_b.dtype)
    fa, fb = np.asarray(feats_a, dtype=np.float64), np.asarray(feats_b, dtype=np.float64)
    h = half_window(window_size)
    side = 2 * h + 1
    oka, okb = within_bounds(fa, image_a.shape, window_size), within_bounds(fb, image_b.shape, window_size)

    def dtype_windows(image, feats, ok):
        out = np.zeros((len(feats), side * side), dtype=rt)
        for i, (x, y) in enumerate(feats):
            if ok[i]:
                out[i] = image[int(y) - h:int(y) + h + 1, int(x) - h:int(x) + h + 1].astype(rt).ravel()
        return out

    pa, pb = dtype_windows(image_a, fa, oka), dtype_windows(image_b, fb, okb)
    acc_t = np.int64 if rt.kind == "i" else np.uint64
    total = np.zeros((len(pa), len(pb)), dtype=acc_t)
    with np.errstate(over="ignore"):
        for k in range(side * side):
            d = pa[:, k][:, None] - pb[:, k][None, :]
            total = total + np.square(d).astype(acc_t)
    score = total.astype(np.float64) / float(side * side)
    return np.where((~oka)[:, None] | (~okb)[None, :], np.inf, score)


# --------------------------------------------------------------------------------------------------
# heap semantics of match_brute_force (matching.py:55-65, 84-97)
# --------------------------------------------------------------------------------------------------
def left_side(position_1based: np.ndarray) -> np.ndarray:
    """True where a 1-based heap position lies in the LEFT subtree of the root (binary 10...)."""
    p = np.asarray(position_1based, dtype=np.int64)
    k = np.floor(np.log2(np.maximum(p, 1))).astype(np.int64)
    return (p >= 2) & (p < (1 << k) + (1 << np.maximum(k - 1, 0)))


def heap_top_two(row: Sequence[float]) -> Tuple[float, int, float]:
    """(heap[0].score, heap[0].b_index, heap[1].score) after heappush-ing the row in order — literally
    (matching.py:60-65).  heap[1] is the root's LEFT child, not necessarily the second smallest."""
    heap: List[Tuple[float, int]] = []

    class Item:
        __slots__ = ("s", "b")

        def __init__(self, s, b):
            self.s, self.b = s, b

        def __lt__(self, other):  # Match.__lt__ compares scores only (matching.py:22-24)
            return self.s < other.s

    items: List[Item] = []
    for b, s in enumerate(row):
        heapq.heappush(items, Item(s, b))
    second = items[1].s if len(items) > 1 else np.nan
    return items[0].s, items[0].b, second


def row_summary(scores: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Closed form of heap_top_two for every row: root = first minimum; heap[1] = min over pushes landing
    in the left subtree of max(score_i, running minimum before i)."""
    nA, nB = scores.shape
    best = np.full(nA, np.nan)
    arg = np.full(nA, -1, dtype=np.int64)
    second = np.full(nA, np.nan)
    if nB == 0:
        return best, arg, second
    run = scores[:, 0].copy()
    arg[:] = 0
    sec = np.full(nA, np.inf)
    lefts = left_side(np.arange(1, nB + 1))  # lefts[i] <-> 1-based position i + 1
    for i in range(1, nB):
        s = scores[:, i]
        with np.errstate(invalid="ignore"):
            lower = s < run
            if lefts[i]:
                cand = np.where(lower, run, s)  # the displaced root, or the new item itself
                sec = np.where(cand < sec, cand, sec)
        arg = np.where(lower, i, arg)
        run = np.where(lower, s, run)
    second = sec if nB > 1 else second
    return run, arg, second


def match_brute_force(scores: np.ndarray, strategies: Optional[set] = None, ratio_threshold: float = 0.5):
    """matching.py:36-118 on a precomputed score matrix.  Returns [(a_index, b_index, score), ...]."""
    strategies = set() if strategies is None else set(strategies)
    nA, nB = scores.shape
    best, arg, second = row_summary(scores)
    if nB == 0:
        if not strategies:
            raise IndexError("list index out of range")  # matches_for_feature[0] on an empty heap
        return []
    keep = np.ones(nA, dtype=bool)
    if RATIO_TEST in strategies and nB > 1:
        with np.errstate(divide="ignore", invalid="ignore"):
            keep &= (best / second) <= ratio_threshold  # matching.py:90-94 (NaN -> dropped)
    rows = [a for a in range(nA) if keep[a]]
    if CROSSCHECK in strategies:
        best_for_b = {}
        for a in rows:  # matching.py:104-111: strictly better replaces, so the earlier a wins ties
            b = int(arg[a])
            if b not in best_for_b or best[best_for_b[b]] > best[a]:
                best_for_b[b] = a
        rows = [a for a in rows if best_for_b[int(arg[a])] == a]
    return [(a, int(arg[a]), float(best[a])) for a in rows]
