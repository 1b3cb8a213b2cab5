"""Glue between the Python call signatures and the device RANSAC engine (host logic only)."""
from __future__ import annotations

import copy
import os
import random
from typing import Sequence

import numpy as np
import torch

from .. import device
from ..common.feature import Feature


def feature_array(features: Sequence[Feature]) -> np.ndarray:
    """(len, 2) float64 array of x, y; accepts lists and NumPy object arrays of Feature."""
    out = np.empty((len(features), 2), dtype=np.float64)
    for i, f in enumerate(features):
        out[i, 0] = f.x
        out[i, 1] = f.y
    return out


def sampler_name() -> str:
    name = os.environ.get("SFM_SAMPLER", "pyshuffle").lower()
    if name not in ("pyshuffle", "philox"):
        raise ValueError(f"SFM_SAMPLER must be 'pyshuffle' or 'philox', got {name!r}")
    return name


def degenerate_policy() -> str:
    policy = os.environ.get("SFM_DEGENERATE", "raise").lower()
    if policy not in ("raise", "skip"):
        raise ValueError(f"SFM_DEGENERATE must be 'raise' or 'skip', got {policy!r}")
    return policy


def local_optimisation_rounds() -> int:
    """``SFM_LOCAL_OPTIMIZATION=<k>``: after RANSAC, refit on all inliers up to k times (0 = off, the reference's
    behaviour: it returns the eight-point model of the winning sample as is)."""
    rounds = int(os.environ.get("SFM_LOCAL_OPTIMIZATION", "0"))
    if rounds < 0:
        raise ValueError("SFM_LOCAL_OPTIMIZATION must be >= 0")
    return rounds


def ransac_feature_pairs(data, camera_matrix, threshold, min_extra, aggregation, iterations):
    """Device route of fit_with_ransac for (Feature, Feature) pairs.  Returns (E or None, inlier pairs).

    Sampler ``pyshuffle`` (default) draws the hypothesis samples from the global ``random`` state
    exactly like the reference's cumulative ``random.shuffle`` (ransac.py:59-64) and advances it;
    ``philox`` (``SFM_SAMPLER=philox``, seed ``SFM_SEED`` or 64 bits from ``random``) is the
    counter-based sampler for large H, generated on the device.
    """
    from .eight_point import EightPointCalculationError

    n = len(data)
    if iterations <= 0:
        return None, []
    if n < 8:
        # reference: data[:8] is short, eight_point_model_fitter raises (epipolar_ransac.py:31-32)
        raise ValueError("Eight feature pairs are expected.")
    dev = device.require_gpu()
    pix_a = feature_array([pair[0] for pair in data])
    pix_b = feature_array([pair[1] for pair in data])
    corr = device.normalize_correspondences(device.to_device(pix_a), device.to_device(pix_b), camera_matrix)
    ws = device.RansacWorkspace(1, n, iterations, dev)
    sampler = sampler_name()
    table = None
    if sampler == "pyshuffle":
        table = device.PyShuffleTable(n, iterations, random, advance=True)
        ws.S.copy_(device.to_device(table.S, dtype=ws.S.dtype).reshape(1, iterations, 8))
    else:
        seed = int(os.environ["SFM_SEED"]) if "SFM_SEED" in os.environ else random.getrandbits(64)
        device.sample_philox(seed, 0, iterations, n, out=ws.S)
    ws.run(corr.reshape(1, n, 4), threshold, min_extra, aggregation)
    outcome = ws.outcome(0)
    if outcome.n_flagged and degenerate_policy() == "raise":
        raise EightPointCalculationError(
            "More than one eigenvalue of Y.T @ Y is small. Cannot confidently estimate"
            f" fundamental matrix. (hypothesis {outcome.first_flagged}, {outcome.n_flagged} in total)"
        )
    if outcome.best_h < 0:
        return None, []
    rounds = local_optimisation_rounds()
    if rounds:
        # extension: the refined model has no "sample", so its inliers come back in
        # index order; an unrefined winner (no refit accepted) falls through to the reference's ordering
        err = ws.result.view(torch.float64)[:, 2]
        E_ref, mask_ref, info = device.refine_inliers(corr.reshape(1, n, 4), ws.E[:, outcome.best_h], ws.mask, err,
                                                      threshold, aggregation, rounds)
        if device.read_refine_info(info)[0][2] > 0:
            keep = np.nonzero(mask_ref.cpu().numpy()[0])[0]
            return E_ref.cpu().numpy().reshape(3, 3), [copy.deepcopy(data[int(i)]) for i in keep]
    survivors = outcome.mask == 1
    if sampler == "pyshuffle":
        perm = table.permutation_after(outcome.best_h)
        order = [int(i) for i in perm[:8]] + [int(i) for i in perm[8:] if survivors[i]]
    else:
        order = [int(i) for i in outcome.sample] + [int(i) for i in np.nonzero(survivors)[0]]
    return outcome.E, [copy.deepcopy(data[i]) for i in order]
