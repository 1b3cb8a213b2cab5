"""Glue between the Python call signatures and the device RANSAC engine (host logic only)."""
from __future__ import annotations

import contextlib
import copy
import gc
import logging
import os
import random
from itertools import starmap
from operator import attrgetter
from typing import Sequence

import numpy as np
import torch

from .. import _native, device
from ..common.feature import Feature


_GET_X, _GET_Y = attrgetter("x"), attrgetter("y")

# SFM_SAMPLER=auto: above this many matches x iterations (SFM_AUTO_PHILOX_WORK overrides) the sampler switches from the
# exact random.shuffle replay to the counter-based device sampler: the replay is inherently sequential (one
# Mersenne-Twister draw per element per iteration, like the reference's own loop), ~3 ns per draw on the host against
# microseconds for the whole pass on the GPU.  The DEFAULT never switches: a drop-in call reproduces the reference's
# samples, winner and inlier list for a given random.seed at every size, and pays the O(matches x iterations) replay.
AUTO_PHILOX_WORK = 10_000_000


@contextlib.contextmanager
def gc_paused():
    """Pause the cyclic garbage collector for a stretch of bulk object creation (pair tuples, Feature copies): every
    few hundred new container objects would otherwise start a collection, and the full ones re-scan each live Feature
    / Match of the caller — at 50 000 matches that is most of a call's time (86 ms of 105 in building the pair list
    alone).  Nothing created here is cyclic; the collector's previous state is restored on exit.

    On exit the objects created meanwhile (95 000 at 50 000 matches) all sit in the youngest generation, and the first
    allocation after ``gc.enable()`` starts a collection that walks every one of them (5-6 ms at that size).  With
    ``SFM_GC_SPLICE=1`` (opt-in since round 4) they are moved to the oldest generation wholesale instead —
    ``gc.freeze(); gc.unfreeze()`` splices the generation lists without visiting an object.  That is a change of the
    EMBEDDING application's collector state — every tracked object of the process is promoted, including the caller's own
    young objects, and a ``gc.freeze()`` another thread takes between the check and the splice would be released by the
    ``unfreeze`` — so a library call does not do it unasked.  Skipped in any case when the process keeps a permanent
    generation of its own (``gc.get_freeze_count() != 0``) and when only a few objects were created."""
    was_enabled = gc.isenabled()
    gc.disable()
    before = gc.get_count()[0]
    try:
        yield
    finally:
        if was_enabled:
            if (os.environ.get("SFM_GC_SPLICE", "0") == "1" and gc.get_count()[0] - before > 20_000
                    and gc.get_freeze_count() == 0):
                gc.freeze()
                gc.unfreeze()
            gc.enable()


def feature_array(features: Sequence[Feature], out: np.ndarray | None = None) -> np.ndarray:
    """(len, 2) float64 array of x, y; accepts lists, tuples and NumPy object arrays of Feature (or of anything
    with ``x`` / ``y`` attributes).  One C-level pass per coordinate (``map`` + ``np.fromiter``)."""
    n = len(features)
    if out is None:
        out = np.empty((n, 2), dtype=np.float64)
    out[:, 0] = np.fromiter(map(_GET_X, features), dtype=np.float64, count=n)
    out[:, 1] = np.fromiter(map(_GET_Y, features), dtype=np.float64, count=n)
    return out


def pair_arrays(data) -> np.ndarray:
    """(2, len, 2) float64: [0] = pixel coordinates of the first features of the pairs, [1] = of the second.  Plain
    lists of (Feature, Feature) tuples are read by the CPython helper (csrc/hostfast.c: 50 000 pairs 11 -> 3 ms)."""
    n = len(data)
    out = np.empty((2, n, 2), dtype=np.float64)
    if n:
        fast = _native.hostfast()
        if fast is not None and fast.pair_arrays(Feature, data, out):
            return out
        first, second = zip(*data)
        feature_array(first, out[0])
        feature_array(second, out[1])
    return out


def match_pairs(features_a, features_b, matches) -> list:
    """``[(features_a[m.a_index], features_b[m.b_index]) for m in matches]`` (reference epipolar_ransac.py:55-57)."""
    fast = _native.hostfast()
    if fast is not None:
        pairs = fast.match_pairs(features_a, features_b, matches)
        if pairs is not None:
            return pairs
    return [(features_a[m.a_index], features_b[m.b_index]) for m in matches]


def copy_pairs(data, order) -> list:
    """Fresh copies of ``data[i]`` for i in ``order`` (an int64 array) — the reference hands back deep copies of the
    caller's features (ransac.py:59 copies the data before shuffling).  Plain ``Feature`` pairs whose attribute values
    are atomic are copied by the CPython helper (csrc/hostfast.c: a new instance per feature holding the same immutable
    values — exactly what ``copy.deepcopy`` produces for them; 32 000 pairs out of 50 000: 25 -> 6 ms); without the
    helper they are rebuilt by their constructor, and anything else is deep-copied.  Runs with the cyclic garbage
    collector paused (``gc_paused``: a collection every few hundred new containers would re-scan every live Feature)."""
    order = np.ascontiguousarray(order, dtype=np.int64)
    with gc_paused():
        fast = _native.hostfast()
        if fast is not None:
            copies = fast.copy_pairs(Feature, data, order) if isinstance(data, list) else None
            if copies is not None:
                return copies
            return [copy.deepcopy(data[i]) for i in order.tolist()]   # anything the helper declines: the general copy
        picked = [data[i] for i in order.tolist()]
        plain = Feature
        # C-level loops throughout (type checks by set(map(type, ...)), attribute reads by map(attrgetter), construction
        # by starmap): 51 -> 32 ms for 32 000 pairs on the build container against comprehensions calling the class
        if picked and set(map(type, picked)) == {tuple} and set(map(len, picked)) == {2}:
            firsts, seconds = zip(*picked)
            if set(map(type, firsts)) == {plain} and set(map(type, seconds)) == {plain}:
                get_x, get_y = attrgetter("x"), attrgetter("y")
                copies_a = starmap(plain, zip(map(get_x, firsts), map(get_y, firsts)))
                copies_b = starmap(plain, zip(map(get_x, seconds), map(get_y, seconds)))
                return list(zip(copies_a, copies_b))
        return [copy.deepcopy(p) for p in picked]


def sampler_name(n: int = 0, iterations: int = 0) -> str:
    """``SFM_SAMPLER=pyshuffle|philox|auto``.  Default ``pyshuffle`` at EVERY size: the exact replay of the reference's
    cumulative ``random.shuffle`` (ransac.py:59-64) from the global ``random`` state, which it advances exactly as the
    reference's loop would — ``random.seed(k)`` reproduces the reference's samples, winner and ordered inlier list, and
    later ``random``-dependent code of the caller sees the same stream.  ``philox``: the counter-based device sampler
    (seed ``SFM_SEED`` or ``random.getrandbits(64)``; consumes 64 bits of the global stream instead of one shuffle per
    iteration).  ``auto``: ``pyshuffle`` up to ``SFM_AUTO_PHILOX_WORK`` (default 10^7) matches x iterations, ``philox``
    beyond — an explicit opt-in to trade the reference's sample stream for speed on large problems."""
    name = os.environ.get("SFM_SAMPLER", "pyshuffle").lower()
    if name == "auto":
        limit = int(float(os.environ.get("SFM_AUTO_PHILOX_WORK", AUTO_PHILOX_WORK)))
        return "pyshuffle" if n * iterations <= limit else "philox"
    if name not in ("pyshuffle", "philox"):
        raise ValueError(f"SFM_SAMPLER must be 'pyshuffle', 'philox' or 'auto', got {name!r}")
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


logger = logging.getLogger(__name__)   # one line per call, never per hypothesis


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
    pix = device.to_device(pair_arrays(data))   # one upload: [2, n, 2]
    corr = device.normalize_correspondences(pix[0], pix[1], camera_matrix)
    ws = device.RansacWorkspace(1, n, iterations, dev)
    sampler = sampler_name(n, iterations)
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
    if logger.isEnabledFor(logging.DEBUG):
        logger.debug("RANSAC-E: %d matches x %d hypotheses (%s sampler): best hypothesis %d, %d extra inliers, "
                     "aggregated error %.6g, %d degenerate sample(s)", n, iterations, sampler, outcome.best_h,
                     outcome.extra_inliers, outcome.error, outcome.n_flagged)
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
            return E_ref.cpu().numpy().reshape(3, 3), copy_pairs(data, keep)
    survivors = outcome.mask == 1
    if sampler == "pyshuffle":
        perm = table.permutation_after(outcome.best_h)
        rest = perm[8:]
        order = np.concatenate([perm[:8], rest[survivors[rest]]])
    else:
        order = np.concatenate([outcome.sample, np.nonzero(survivors)[0]])
    return outcome.E, copy_pairs(data, order)
