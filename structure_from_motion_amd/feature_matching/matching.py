"""Brute-force feature matching with the reference's interface
(reference ``lib/feature_matching/matching.py:15-118``).

``match_brute_force`` has two routes behind one signature:

* when ``score_function`` is a window score this package knows (NCC / SSD over two images — an
  ``ImagePairScore`` object, a ``functools.partial`` of ``calculate_ncc`` / ``calculate_ssd`` with both images
  bound, or a closure shaped like ``apps/sfm.py:_create_score_function``), all ``|A| x |B|`` scores and the
  per-feature heap summaries are computed by HIP kernels;
* for any other callable the loop is host logic, with a real ``heapq`` like the reference.

Kept quirks: the ratio test divides ``heap[0]`` by ``heap[1]`` of the per-feature heap — the root's left
child, not necessarily the runner-up; features failing it are *dropped* from the result (not replaced by an
empty ``Match``); cross-check keeps a feature only if it is the best (earliest on ties) claimant of its ``b``.
"""
from __future__ import annotations

import dataclasses
import functools
import heapq
from enum import Enum
from math import inf
from typing import Callable, Dict, List, NewType, Optional, Set

import numpy as np

from ..common.feature import Feature

# Interface of a score function: (feature_a, feature_b) -> float, lower is better.
ScoreFunction = NewType("ScoreFunction", Callable[[Feature, Feature], float])


@dataclasses.dataclass
class Match:
    a_index: int = -1
    b_index: int = -1
    # Lower is better for every score function.
    match_score: float = inf

    def __lt__(self, other) -> bool:
        return self.match_score < other.match_score


class ValidationStrategy(Enum):
    """How candidate matches are validated."""

    CROSSCHECK = 1  # (i, j) only if j is i's best and i is the best claimant of j
    RATIO_TEST = 2  # best score / heap[1] score must not exceed a threshold


class ImagePairScore:
    """A window score over two fixed images — callable like any score function, and recognised by
    ``match_brute_force`` so that the whole score matrix is computed on the GPU."""

    def __init__(self, image_a: np.ndarray, image_b: np.ndarray, metric_function, window_size: Optional[int] = None):
        if getattr(metric_function, "_sfm_hip_metric", None) is None:
            raise TypeError("metric_function must be calculate_ncc or calculate_ssd")
        self.image_a, self.image_b = image_a, image_b
        self.metric_function = metric_function
        self.window_size = window_size

    def __call__(self, feature_a: Feature, feature_b: Feature) -> float:
        if self.window_size is None:
            return self.metric_function(self.image_a, self.image_b, feature_a, feature_b)
        return self.metric_function(self.image_a, self.image_b, feature_a, feature_b, self.window_size)


def _default_window(metric_function) -> int:
    return metric_function.__defaults__[0]


def _window_of_partial(p: functools.partial, bound_positional: int) -> Optional[int]:
    """window_size carried by a partial of calculate_ncc/_ssd, given how many positionals are bound."""
    if "window_size" in p.keywords:
        return int(p.keywords["window_size"])
    if len(p.args) > bound_positional:
        return int(p.args[bound_positional])
    return _default_window(p.func)


def _device_score_spec(score_function):
    """(metric, image_a, image_b, window_size) if the callable is a recognised image-pair window score."""
    if isinstance(score_function, ImagePairScore):
        ws = score_function.window_size
        fn = score_function.metric_function
        return fn._sfm_hip_metric, score_function.image_a, score_function.image_b, _default_window(fn) if ws is None else ws
    if isinstance(score_function, functools.partial):
        fn = score_function.func
        metric = getattr(fn, "_sfm_hip_metric", None)
        args = score_function.args
        if metric is not None and len(args) in (2, 3) and all(isinstance(a, np.ndarray) for a in args[:2]):
            if len(args) == 3:
                return None  # feature bound positionally: not a (feature_a, feature_b) function any more
            return metric, args[0], args[1], _window_of_partial(score_function, 4)
        return None
    # closure shaped like apps/sfm.py:_create_score_function: free variables (full_score_function, image_a,
    # image_b) where full_score_function(image_a, image_b, feature_a, feature_b) is calculate_ncc/_ssd or a
    # partial of it that only binds window_size
    code = getattr(score_function, "__code__", None)
    cells = getattr(score_function, "__closure__", None)
    if code is None or not cells or code.co_argcount != 2:
        return None
    free = dict(zip(code.co_freevars, (c.cell_contents for c in cells)))
    if set(free) != {"full_score_function", "image_a", "image_b"}:
        return None
    full, image_a, image_b = free["full_score_function"], free["image_a"], free["image_b"]
    if not (isinstance(image_a, np.ndarray) and isinstance(image_b, np.ndarray)):
        return None
    if isinstance(full, functools.partial):
        metric = getattr(full.func, "_sfm_hip_metric", None)
        if metric is None or full.args or set(full.keywords) - {"window_size"}:
            return None
        return metric, image_a, image_b, _window_of_partial(full, 4)
    metric = getattr(full, "_sfm_hip_metric", None)
    if metric is None:
        return None
    return metric, image_a, image_b, _default_window(full)


def match_brute_force(
    features_a: List[Feature],
    features_b: List[Feature],
    score_function: ScoreFunction,
    *,
    validation_strategies: ValidationStrategy | Set[ValidationStrategy] | None = None,
    ratio_test_threshold: float = 0.5
) -> List[Match]:
    """Match every feature of ``features_a`` against all of ``features_b`` by exhaustive scoring.

    Returns, in ``features_a`` order, the best match of each feature that survives the requested validation
    strategies (``ratio_test_threshold`` is the largest allowed ``heap[0] / heap[1]`` score ratio)."""
    if validation_strategies is None:
        strategies = set()
    elif isinstance(validation_strategies, set):
        strategies = validation_strategies
    else:
        strategies = {validation_strategies}

    spec = _device_score_spec(score_function)
    if spec is not None and len(features_a) > 0 and len(features_b) > 0:
        best, arg, second = _device_rows(spec, features_a, features_b)
        has_second = len(features_b) > 1
    else:
        best, arg, second, has_second = _host_rows(features_a, features_b, score_function)

    if len(features_b) == 0:
        # every heap of the reference is empty here (matching.py:55-65).  Its ratio filter drops empty heaps
        # (:84-97) and an empty feature list has no heap to index; otherwise it indexes heap[0] of an empty heap —
        # in the cross-check (:105) or in the final list comprehension (:79)
        if len(features_a) == 0 or ValidationStrategy.RATIO_TEST in strategies:
            return []
        raise IndexError("list index out of range")
    # plain Python numbers from here on (20 000 rows: NumPy scalars made the loops below three times as slow)
    best_list = best.tolist() if isinstance(best, np.ndarray) else [_plain(v) for v in best]
    arg_list = arg.tolist() if isinstance(arg, np.ndarray) else [int(v) for v in arg]
    rows = range(len(features_a))
    if ValidationStrategy.RATIO_TEST in strategies and has_second:
        with np.errstate(divide="ignore", invalid="ignore"):
            passed = (np.asarray(best, dtype=np.float64) / np.asarray(second, dtype=np.float64)) <= ratio_test_threshold
        rows = np.flatnonzero(passed).tolist()
    if ValidationStrategy.CROSSCHECK in strategies:
        claimant: Dict[int, int] = {}
        for a in rows:  # a strictly better score replaces the claimant, so the earliest wins ties
            b = arg_list[a]
            if b not in claimant or best_list[claimant[b]] > best_list[a]:
                claimant[b] = a
        rows = [a for a in rows if claimant[arg_list[a]] == a]
    return [Match(a, arg_list[a], best_list[a]) for a in rows]


def _plain(value):
    return value.item() if isinstance(value, np.generic) else value


def _device_rows(spec, features_a, features_b):
    from . import _device_match

    metric, image_a, image_b, window_size = spec
    return _device_match.match_summary(metric, image_a, image_b, features_a, features_b, window_size)


def _host_rows(features_a, features_b, score_function):
    """Generic route: one heap per feature of A, exactly like the reference (matching.py:55-65)."""
    best, arg, second = [], [], []
    for a_index, feature_a in enumerate(features_a):
        heap: List[Match] = []
        for b_index, feature_b in enumerate(features_b):
            heapq.heappush(heap, Match(a_index=a_index, b_index=b_index, match_score=score_function(feature_a, feature_b)))
        if heap:
            best.append(heap[0].match_score)
            arg.append(heap[0].b_index)
            second.append(heap[1].match_score if len(heap) > 1 else float("nan"))
        else:
            best.append(float("nan")); arg.append(-1); second.append(float("nan"))
    return best, arg, second, len(features_b) > 1
