"""Generic RANSAC driver with the reference's interface (reference ``lib/ransac/ransac.py:19-108``).

Two execution routes behind one signature:

* when ``model_fitter`` / ``inlier_scorer`` are the eight-point fitter and SED scorer of
  ``epipolar_ransac`` (which is what ``estimate_essential_mat_with_ransac`` passes, exactly like the
  reference's ``epipolar_ransac.py:58-67``), the whole loop — fit, H x N scoring, gate, aggregation,
  selection — runs as HIP kernels on the MI355X (``device.RansacWorkspace``);
* for arbitrary Python callables (e.g. the 2-point line fitter of the reference's own
  ``test_ransac.py``) the loop is host logic: there is nothing to put on a GPU.

Semantics kept from the reference: first ``model_fit_data_count`` items of a cumulative
in-place ``random.shuffle`` of a deep copy are the sample (Q3); sample points are inliers
unconditionally and enter the aggregate (Q2); the model with the strictly lowest aggregated inlier
error among gated hypotheses wins, earliest first, NaN never (Q1); fitter exceptions abort the call
(Q4); defaults 100 iterations / RMS / 0 extra inliers (Q13).
"""
from __future__ import annotations

import copy
import os
import random
from enum import Enum
from math import inf, sqrt
from typing import Any, Callable, Optional, Sequence, Tuple

import numpy as np

DEFAULT_MAX_ITERATIONS = 100


class ErrorAggregationMethod(Enum):
    SUM = "sum"
    SQUARE = "square"
    MEAN = "mean"
    RMS = "rms"


_AGGREGATION_CODE = {"sum": 0, "square": 1, "mean": 2, "rms": 3}


def aggregation_code(method: ErrorAggregationMethod) -> int:
    """SFM_AGG_* code of include/sfm_hip.h for an ErrorAggregationMethod (matched by value, like the
    reference's ``_aggregate_error`` does, so enums from another import path also work)."""
    try:
        return _AGGREGATION_CODE[method.value]
    except (KeyError, AttributeError):
        raise NotImplementedError(method)


def fit_with_ransac(
    data: Sequence,
    model_fit_data_count: int,
    model_fitter: Callable[[Sequence], Any],
    inlier_scorer: Callable[[Any, Any], float],
    inlier_threshold: float,
    min_num_extra_inliers: int | None = None,
    error_aggregation_method: ErrorAggregationMethod | None = None,
    max_iterations: int | None = None,
) -> Tuple[Optional[Any], Sequence]:
    """Fit a model with RANSAC; returns ``(best_model, inliers)`` or raises ``ValueError`` if no
    hypothesis reaches ``min_num_extra_inliers`` extra inliers.  See the module docstring."""
    iterations = DEFAULT_MAX_ITERATIONS if max_iterations is None else max_iterations
    method = ErrorAggregationMethod.RMS if error_aggregation_method is None else error_aggregation_method
    min_extra = 0 if min_num_extra_inliers is None else min_num_extra_inliers

    spec = _device_spec(model_fitter, inlier_scorer, model_fit_data_count)
    if spec is not None:
        from ..epipolar import _engine

        model, inliers = _engine.ransac_feature_pairs(
            data, spec, inlier_threshold, min_extra, aggregation_code(method), iterations)
    else:
        model, inliers = _host_loop(data, model_fit_data_count, model_fitter, inlier_scorer,
                                    inlier_threshold, min_extra, method, iterations)
    if model is None:
        raise ValueError(
            f"No model could be found with at least {min_extra + model_fit_data_count} inliers."
        )
    return model, inliers


def _device_spec(model_fitter, inlier_scorer, model_fit_data_count):
    """Camera matrix if (fitter, scorer) are partials of the eight-point / SED pair, else None."""
    fit_fn = getattr(model_fitter, "func", None)
    score_fn = getattr(inlier_scorer, "func", None)
    if fit_fn is None or score_fn is None or model_fit_data_count != 8:
        return None
    if not getattr(fit_fn, "_sfm_hip_role", None) == "eight_point_fitter":
        return None
    if not getattr(score_fn, "_sfm_hip_role", None) == "sed_scorer":
        return None
    k_fit = model_fitter.keywords.get("camera_matrix") if not model_fitter.args else None
    k_score = inlier_scorer.keywords.get("camera_matrix") if not inlier_scorer.args else None
    if k_fit is None or k_score is None or not np.array_equal(np.asarray(k_fit), np.asarray(k_score)):
        return None
    return np.asarray(k_fit, dtype=np.float64)


def _host_loop(data, k, model_fitter, inlier_scorer, threshold, min_extra, method, iterations):
    """Host driver for arbitrary callables (reference ransac.py:55-86)."""
    pool = copy.deepcopy(data)
    best_model, best_inliers, best_error = None, [], inf
    steps = range(iterations)
    if os.environ.get("SFM_PROGRESS", "0") not in ("", "0"):   # the reference always draws this bar (ransac.py:61)
        from tqdm import tqdm

        steps = tqdm(steps)
    for _ in steps:
        random.shuffle(pool)
        sample, rest = pool[:k], pool[k:]
        model = model_fitter(sample)
        survivors = [item for item in rest if inlier_scorer(model, item) <= threshold]
        if not (min_extra <= len(survivors)):
            continue
        candidates = sample + survivors
        error = _aggregate_error([inlier_scorer(model, item) for item in candidates], method)
        if error < best_error:
            best_model, best_inliers, best_error = model, candidates, error
    return best_model, best_inliers


def _aggregate_error(errors: list, aggregation_method: ErrorAggregationMethod) -> float:
    """reference ransac.py:96-108 (host route only; the device route aggregates in select_best_kernel)."""
    code = aggregation_code(aggregation_method)
    if code == 0:
        return sum(errors)
    squares = np.square(errors)
    if code == 1:
        return np.sum(squares).item()
    if code == 2:
        return np.mean(errors).item()
    return np.sqrt(np.mean(squares)).item()
