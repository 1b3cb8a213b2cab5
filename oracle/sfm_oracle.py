"""CPU ORACLE — test infrastructure, NOT product code.

A numpy (float64) restatement of the reference's RANSAC essential-matrix / pose-recovery /
triangulation path, written so it can be driven by an explicit sample table ``S[H, 8]`` and compared
with the HIP kernels on identical inputs.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under
``structure_from_motion_amd/`` does.

Every function cites the reference lines it follows (paths relative to the reference repo root).
Parity pin: ``tests/golden/*.npz`` hold outputs of the *real* reference (imported unmodified in the
build container by ``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this file
against them.

Floating-point conventions fixed here (the reference leaves them to NumPy/BLAS):
  * every 3-term dot product is evaluated left to right with separate multiply and add roundings
    (no FMA), e.g. ``(xb*E00 + yb*E10) + E20``;
  * ``r = (b^T E) a`` is associated the way ``coord_b.T @ e @ coord_a`` parses (``sed.py:21``);
  * per-hypothesis error sums run over the 8 sample points first (in sample order) and then over the
    surviving non-sample points in the order given (index order for an explicit table), sequentially.
"""
from __future__ import annotations

import itertools
import random as _pyrandom
from typing import Optional, Sequence, Tuple

import numpy as np

SUM, SQUARE, MEAN, RMS = "sum", "square", "mean", "rms"
MODEL_POINTS = 8
VERY_SMALL = 1e-10  # eight_point.py:414
CHEIRALITY_TOLERANCE = 1e-8  # eight_point.py:477
DEFAULT_DISTANCE_THRESHOLD = 50.0  # eight_point.py:469-470


class OracleDegenerateSample(Exception):
    """Mirror of EightPointCalculationError (eight_point.py:20-23) inside the oracle."""


# --------------------------------------------------------------------------------------------------
# coordinates
# --------------------------------------------------------------------------------------------------
def to_normalized_image_coords(pix: np.ndarray, K: np.ndarray) -> np.ndarray:
    """(x-cx)/fx, (y-cy)/fy using only K[0][0], K[1][1], K[0][2], K[1][2] (eight_point.py:127-133)."""
    pix = np.asarray(pix, dtype=np.float64)
    out = np.empty_like(pix)
    out[..., 0] = (pix[..., 0] - K[0][2]) / K[0][0]
    out[..., 1] = (pix[..., 1] - K[1][2]) / K[1][1]
    return out


def pack_correspondences(norm_a: np.ndarray, norm_b: np.ndarray) -> np.ndarray:
    """(N,2),(N,2) -> (N,4) rows [xa, ya, xb, yb]: the layout the HIP kernels stream."""
    return np.ascontiguousarray(np.hstack([norm_a, norm_b]), dtype=np.float64)


# --------------------------------------------------------------------------------------------------
# symmetric epipolar distance (sed.py:7-30)
# --------------------------------------------------------------------------------------------------
def sed_values(E: np.ndarray, corr: np.ndarray) -> np.ndarray:
    """SED of every correspondence under every E.

    E: (..., 3, 3); corr: (N, 4).  Returns (..., N).
    line_b = E^T b (sed.py:25), r = (b^T E) a = line_b . a (sed.py:21), line_a = E a (sed.py:24),
    sed = (1/(la0^2+la1^2) + 1/(lb0^2+lb1^2)) * r^2 (sed.py:27-29).
    """
    E = np.asarray(E, dtype=np.float64)
    e = E[..., None, :, :]  # broadcast over points
    xa, ya, xb, yb = (corr[:, k] for k in range(4))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        lb0 = (xb * e[..., 0, 0] + yb * e[..., 1, 0]) + e[..., 2, 0]
        lb1 = (xb * e[..., 0, 1] + yb * e[..., 1, 1]) + e[..., 2, 1]
        lb2 = (xb * e[..., 0, 2] + yb * e[..., 1, 2]) + e[..., 2, 2]
        r = (lb0 * xa + lb1 * ya) + lb2
        la0 = (e[..., 0, 0] * xa + e[..., 0, 1] * ya) + e[..., 0, 2]
        la1 = (e[..., 1, 0] * xa + e[..., 1, 1] * ya) + e[..., 1, 2]
        da = la0 * la0 + la1 * la1
        db = lb0 * lb0 + lb1 * lb1
        return (1.0 / da + 1.0 / db) * (r * r)


# --------------------------------------------------------------------------------------------------
# normalised eight-point fit (eight_point.py:136-170 and helpers)
# --------------------------------------------------------------------------------------------------
def hartley_normalize(coords: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """eight_point.py:308-338.  coords (..., n, 2) -> normalised coords, forward transform T (..., 3, 3)."""
    centroid = np.mean(coords, axis=-2)
    centered = coords - centroid[..., None, :]
    norms = np.sqrt(centered[..., 0] * centered[..., 0] + centered[..., 1] * centered[..., 1])
    scale = np.sqrt(2.0) / np.mean(norms, axis=-1)
    normalized = centered * scale[..., None, None]
    T = np.zeros(coords.shape[:-2] + (3, 3), dtype=np.float64)
    T[..., 0, 0] = scale
    T[..., 1, 1] = scale
    T[..., 0, 2] = -scale * centroid[..., 0]
    T[..., 1, 2] = -scale * centroid[..., 1]
    T[..., 2, 2] = 1.0
    return normalized, T


def y_columns(ca: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """eight_point.py:378-393: [xb*xa, xb*ya, xb, yb*xa, yb*ya, yb, xa, ya, 1].  (..., n, 2) -> (..., n, 9)."""
    xa, ya, xb, yb = ca[..., 0], ca[..., 1], cb[..., 0], cb[..., 1]
    one = np.ones_like(xa)
    return np.stack([xb * xa, xb * ya, xb, yb * xa, yb * ya, yb, xa, ya, one], axis=-1)


def yty(ca: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """eight_point.py:363-375: sum of the 8 outer products, accumulated in point order."""
    cols = y_columns(ca, cb)
    acc = np.zeros(cols.shape[:-2] + (9, 9), dtype=np.float64)
    for i in range(cols.shape[-2]):
        c = cols[..., i, :]
        acc = acc + c[..., :, None] * c[..., None, :]
    return acc


def fit_from_sample_coords(ca: np.ndarray, cb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Batched normalised eight-point fit.

    ca, cb: (H, 8, 2) coordinates of the sampled pairs (image a / image b).
    Returns (E (H,3,3) with E[2,2]=1, degenerate flag (H,) bool, second-smallest eigenvalue (H,)).
    Follows estimate_fundamental_mat (eight_point.py:136-170): Hartley-normalise (:154-156),
    Y^T Y (:158), eig + degeneracy predicate + argmin|w| (:410-425), rank-2 via SVD (:440-445),
    T2.T @ F @ T1 (:163), divide by [2,2] (:166).
    """
    na, T1 = hartley_normalize(ca)
    nb, T2 = hartley_normalize(cb)
    A = yty(na, nb)
    w, v = np.linalg.eig(A)  # eight_point.py:410 (general dgeev, as the reference)
    w = np.real(w)
    v = np.real(v)
    sorted_w = np.sort(w, axis=-1)
    degenerate = np.any(sorted_w[..., 1:] <= VERY_SMALL, axis=-1)  # :415-416
    min_index = np.argmin(np.abs(w), axis=-1)  # :423
    v_min = np.take_along_axis(v, min_index[..., None, None], axis=-1)[..., 0]
    f_est = v_min.reshape(v_min.shape[:-1] + (3, 3))
    u, s, vh = np.linalg.svd(f_est)  # :440
    s = s.copy()
    s[..., 2] = 0.0  # :443
    f = (u * s[..., None, :]) @ vh  # u @ diag(s) @ vh (:444-445)
    e = np.swapaxes(T2, -1, -2) @ f @ T1  # :163
    with np.errstate(divide="ignore", invalid="ignore"):
        e = e / e[..., 2:3, 2:3]  # :166 (unguarded)
    return e, degenerate, sorted_w[..., 1]


def fit_hypotheses(corr: np.ndarray, S: np.ndarray):
    """Fit one E per row of the sample table S (H, 8) of indices into corr (N, 4)."""
    pts = corr[S]  # (H, 8, 4)
    return fit_from_sample_coords(pts[..., 0:2], pts[..., 2:4])


_POOL_CORR = None  # the correspondence set of a FitPool worker (set once per worker process)


def _fit_pool_init(corr: np.ndarray) -> None:
    global _POOL_CORR
    _POOL_CORR = corr


def _fit_pool_block(S_block: np.ndarray):
    return fit_hypotheses(_POOL_CORR, S_block)


def _fit_pool_philox_block(args):
    seed, h_begin, h_count = args
    S = philox_sample_table(seed, h_begin, h_count, _POOL_CORR.shape[0])
    return (S,) + tuple(fit_hypotheses(_POOL_CORR, S))


class FitPool:
    """fit_hypotheses over contiguous hypothesis blocks on ``workers`` processes (BASELINE.md §3 ii: the all-core
    form of the same numpy restatement; results are those of fit_hypotheses block by block, i.e. identical).
    Workers are spawned, not forked: the parent may hold an initialised GPU runtime."""

    def __init__(self, corr: np.ndarray, workers: int):
        import multiprocessing as mp
        import os

        self.workers = max(1, int(workers))
        saved = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS")}
        os.environ["OPENBLAS_NUM_THREADS"] = "1"   # one LAPACK thread per worker: the pool is the parallelism
        os.environ["OMP_NUM_THREADS"] = "1"
        try:
            self.pool = mp.get_context("spawn").Pool(self.workers, initializer=_fit_pool_init,
                                                     initargs=(np.ascontiguousarray(corr),))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def fit(self, S: np.ndarray, block: int = 0):
        h = S.shape[0]
        if block <= 0:   # ~2000 hypotheses per block keep the batched LAPACK working set in cache (2x faster than 80 000)
            block = min(2500, max(256, -(-h // (4 * self.workers))))
        parts = self.pool.map(_fit_pool_block, [S[i:i + block] for i in range(0, h, block)])
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    def sample_and_fit(self, seed: int, h_begin: int, h_count: int, block: int = 0):
        """(S, E, degenerate, lambda2) for hypotheses [h_begin, h_begin + h_count) of the Philox stream: the table is
        counter-based, so every worker draws its own block of it."""
        if block <= 0:
            block = min(2500, max(256, -(-h_count // (4 * self.workers))))
        jobs = [(seed, h_begin + i, min(block, h_count - i)) for i in range(0, h_count, block)]
        parts = self.pool.map(_fit_pool_philox_block, jobs)
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(4))

    def close(self):
        self.pool.close()
        self.pool.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def estimate_fundamental_mat(coords_a: np.ndarray, coords_b: np.ndarray) -> np.ndarray:
    """Single 8-point fit; raises like eight_point.py:417-421 on a degenerate sample."""
    if coords_a.shape != (8, 2) or coords_b.shape != (8, 2):
        raise ValueError("Exactly eight matches are needed")  # eight_point.py:151-152
    e, deg, _ = fit_from_sample_coords(coords_a[None], coords_b[None])
    if deg[0]:
        raise OracleDegenerateSample("More than one eigenvalue of Y.T @ Y is small.")
    return e[0]


# --------------------------------------------------------------------------------------------------
# RANSAC scoring / selection (ransac.py:61-86)
# --------------------------------------------------------------------------------------------------
def score_hypotheses(
    corr: np.ndarray, E: np.ndarray, S: np.ndarray, thr: float, block: int = 0
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Per hypothesis: (#non-sample points with sed <= thr, sum sed, sum sed^2) where both sums run over
    the 8 sample points (unconditionally, ransac.py:76-79) followed by the surviving non-sample
    points in index order (ransac.py:70-74), accumulated sequentially."""
    N = corr.shape[0]
    H = E.shape[0]
    cnt = np.zeros(H, dtype=np.int32)
    s1 = np.zeros(H, dtype=np.float64)
    s2 = np.zeros(H, dtype=np.float64)
    if block <= 0:
        block = max(1, min(H, (1 << 21) // max(N, 1)))
    rows = np.arange(block)
    for h0 in range(0, H, block):
        h1 = min(H, h0 + block)
        nb = h1 - h0
        sed = sed_values(E[h0:h1], corr)  # (nb, N)
        sample = S[h0:h1]
        sample_sed = np.take_along_axis(sed, sample, axis=1)  # (nb, 8)
        with np.errstate(invalid="ignore"):
            keep = sed <= thr
        keep[rows[:nb, None], sample] = False  # sample points are not "rest of data"
        cnt[h0:h1] = keep.sum(axis=1)
        rest = np.where(keep, sed, 0.0)
        seq = np.concatenate([sample_sed, rest], axis=1)
        s1[h0:h1] = np.add.accumulate(seq, axis=1)[:, -1]
        with np.errstate(over="ignore", invalid="ignore"):
            s2[h0:h1] = np.add.accumulate(seq * seq, axis=1)[:, -1]
    return cnt, s1, s2


def aggregate(cnt: np.ndarray, s1: np.ndarray, s2: np.ndarray, method: str) -> np.ndarray:
    """ransac.py:96-108 expressed on (count, sum, sum of squares); n = 8 + count."""
    n = (cnt.astype(np.float64) + float(MODEL_POINTS))
    with np.errstate(invalid="ignore", over="ignore"):
        if method == SUM:
            return s1.copy()
        if method == SQUARE:
            return s2.copy()
        if method == MEAN:
            return s1 / n
        if method == RMS:
            return np.sqrt(s2 / n)
    raise NotImplementedError(method)


def select_best(err: np.ndarray, cnt: np.ndarray, min_extra: float) -> Tuple[int, float]:
    """ransac.py:75-86: first hypothesis (lowest index) with the strictly smallest error among the
    gated ones; NaN and +inf never win.  Returns (-1, inf) if none."""
    gated = (cnt >= min_extra) & (err < np.inf)  # NaN < inf is False
    if not np.any(gated):
        return -1, float("inf")
    masked = np.where(gated, err, np.inf)
    best = int(np.argmin(masked))  # argmin returns the first minimum
    return best, float(masked[best])


def inlier_indices(corr: np.ndarray, E: np.ndarray, sample: np.ndarray, thr: float,
                   rest_order: Optional[np.ndarray] = None) -> np.ndarray:
    """Indices of the returned inliers: the sample (in sample order) then the surviving rest
    (ransac.py:76) in ``rest_order`` (default: increasing index)."""
    sed = sed_values(E, corr)
    with np.errstate(invalid="ignore"):
        keep = sed <= thr
    keep[sample] = False
    if rest_order is None:
        rest = np.nonzero(keep)[0]
    else:
        rest = np.asarray([i for i in rest_order if keep[i]], dtype=np.int64)
    return np.concatenate([np.asarray(sample, dtype=np.int64), rest.astype(np.int64)])


def ransac_essential(corr, S, thr, min_extra=0, method=RMS):
    """Whole estimate_essential_mat_with_ransac (epipolar_ransac.py:45-70) for an explicit sample
    table.  Returns dict(best, E, err, inliers, cnt, s1, s2, Eall, degenerate)."""
    Eall, deg, _ = fit_hypotheses(corr, S)
    cnt, s1, s2 = score_hypotheses(corr, Eall, S, thr)
    err = aggregate(cnt, s1, s2, method)
    best, best_err = select_best(err, cnt, min_extra)
    out = dict(best=best, err=best_err, cnt=cnt, s1=s1, s2=s2, Eall=Eall, degenerate=deg,
               errs=err, E=None, inliers=None)
    if best >= 0:
        out["E"] = Eall[best]
        out["inliers"] = inlier_indices(corr, Eall[best], S[best], thr)
    return out


def refit_on_points(corr: np.ndarray, indices: np.ndarray):
    """The reference's eight-point pipeline (estimate_fundamental_mat, eight_point.py:154-166) applied to the
    M >= 8 correspondences ``indices`` instead of exactly 8: _normalize_coords (:308-338), the outer-product
    accumulation of _get_y_col columns (:363-393), _compute_f_est (:396-427) and
    _enforce_fundamental_mat_constraints (:430-446) are written for any number of points; only the entry check
    (:151-152) and an assert in _get_yT_y (:365) pin 8.  Pinned by tests/golden/g13_refit.npz (the same composition
    of the real helpers).  Returns (E (3,3), degenerate flag)."""
    pts = corr[np.asarray(indices, dtype=np.int64)]
    e, deg, _ = fit_from_sample_coords(pts[None, :, 0:2], pts[None, :, 2:4])
    return e[0], bool(deg[0])


def local_optimisation(corr, E0, mask0, err0, thr, method=RMS, iterations=1):
    """Local optimisation of a RANSAC winner — an EXTENSION; the reference has no such step,
    so this function is the definition the device kernel is held to (the fit inside it is pinned to the reference's
    helpers by tests/golden/g13_refit.npz).  Up to ``iterations`` times: refit on all current inliers, re-score
    all points (sed <= thr), keep the refit iff it has more inliers, or as many and a lower aggregated error
    (ransac.py:96-108 over exactly the inliers); stop at the first refit not kept, degenerate, or with < 8 inliers.
    Returns (E, boolean mask, count, error, accepted refits)."""
    best_E = np.asarray(E0, dtype=np.float64).reshape(3, 3)
    best_mask = np.asarray(mask0) != 0
    best_cnt, best_err, accepted = int(best_mask.sum()), float(err0), 0
    for _ in range(iterations):
        idx = np.nonzero(best_mask)[0]
        if len(idx) < MODEL_POINTS:
            break
        E, degenerate = refit_on_points(corr, idx)
        if degenerate:
            break
        sed = sed_values(E, corr)
        with np.errstate(invalid="ignore"):
            keep = sed <= thr
        cnt = int(keep.sum())
        kept = sed[keep]
        s1 = np.add.accumulate(kept)[-1] if cnt else 0.0
        s2 = np.add.accumulate(kept * kept)[-1] if cnt else 0.0
        with np.errstate(invalid="ignore", divide="ignore"):
            err = {SUM: s1, SQUARE: s2, MEAN: np.float64(s1) / cnt if cnt else np.nan,
                   RMS: np.sqrt(np.float64(s2) / cnt) if cnt else np.nan}[method]
        if not (cnt > best_cnt or (cnt == best_cnt and err < best_err)):
            break
        best_E, best_mask, best_cnt, best_err, accepted = E, keep, cnt, float(err), accepted + 1
    return best_E, best_mask, best_cnt, best_err, accepted


def aggregate_literal(errors: Sequence[float], method: str) -> float:
    """ransac.py:96-108 verbatim semantics on a compact error list (python sum / numpy pairwise)."""
    if method == SUM:
        return sum(errors)
    if method == SQUARE:
        return np.sum(np.square(errors)).item()
    if method == MEAN:
        return np.mean(errors).item()
    if method == RMS:
        return np.sqrt(np.mean(np.square(errors))).item()
    raise NotImplementedError(method)


# --------------------------------------------------------------------------------------------------
# samplers
# --------------------------------------------------------------------------------------------------
def pyshuffle_sample_table(n: int, iterations: int, k: int = MODEL_POINTS, rng=_pyrandom):
    """Replay of ransac.py:59-64: a cumulative in-place ``random.shuffle`` of the data, first k taken
    as the sample.  Works on an index permutation; returns (S (iterations,k) int32, list of full
    permutations).  Uses (and advances) the given ``random`` state exactly as the reference does."""
    perm = list(range(n))
    S = np.empty((iterations, k), dtype=np.int32)
    perms = []
    for it in range(iterations):
        rng.shuffle(perm)
        S[it] = perm[:k]
        perms.append(list(perm))
    return S, perms


_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, key: Tuple[int, int]) -> np.ndarray:
    """Philox-4x32-10 (Salmon et al., SC'11).  ctr: (..., 4) uint32; returns (..., 4) uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _PHILOX_M0 * c[0]
        p1 = _PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def philox_sample_table(seed: int, h_begin: int, h_count: int, n: int) -> np.ndarray:
    """Counter-based sampler: hypothesis h draws 8 distinct indices in [0, n) as the first 8 positions
    of a Fisher-Yates shuffle of range(n) driven by Philox(key=seed, counter=(h, 0, block, 0)).
    Draw k: j = k + ((u_k * (n - k)) >> 32) with u_k the k-th 32-bit word (multiply-shift range
    reduction, no rejection); position k then swaps with position j.  This is the build's own sampler
    (the reference's cumulative shuffle is inherently sequential, ransac.py:62)."""
    if n < MODEL_POINTS:
        raise ValueError("need at least 8 correspondences")
    h = np.arange(h_begin, h_begin + h_count, dtype=np.uint64)
    ctr = np.zeros((h_count, 4), dtype=np.uint32)
    ctr[:, 0] = (h & _MASK32).astype(np.uint32)
    ctr[:, 1] = (h >> np.uint64(32)).astype(np.uint32)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ctr[:, 2] = 0
    w0 = philox4x32_10(ctr, key)
    ctr[:, 2] = 1
    w1 = philox4x32_10(ctr, key)
    u = np.concatenate([w0, w1], axis=1).astype(np.uint64)  # (H, 8)
    S = np.empty((h_count, MODEL_POINTS), dtype=np.int64)
    # sparse Fisher-Yates: value at position p is p unless displaced by an earlier swap
    disp_pos = np.full((h_count, MODEL_POINTS), -1, dtype=np.int64)
    disp_val = np.zeros((h_count, MODEL_POINTS), dtype=np.int64)
    for k in range(MODEL_POINTS):
        j = k + ((u[:, k] * np.uint64(n - k)) >> np.uint64(32)).astype(np.int64)
        # current value at position j (latest displacement wins) and at position k
        vj = j.copy()
        vk = np.full(h_count, k, dtype=np.int64)
        for m in range(k):
            hit_j = disp_pos[:, m] == j
            vj = np.where(hit_j, disp_val[:, m], vj)
            hit_k = disp_pos[:, m] == k
            vk = np.where(hit_k, disp_val[:, m], vk)
        S[:, k] = vj
        # position j now holds the old value of position k
        disp_pos[:, k] = j
        disp_val[:, k] = vk
    return S.astype(np.int32)


# --------------------------------------------------------------------------------------------------
# pose recovery (eight_point.py:181-280, 449-488) and triangulation (triangulation.py:9-62)
# --------------------------------------------------------------------------------------------------
def recover_all_r_t(e: np.ndarray):
    """eight_point.py:245-280."""
    u, s, vh = np.linalg.svd(e)
    det_u = np.linalg.det(u)
    det_vh = np.linalg.det(vh)
    if not np.isclose(abs(det_u), 1):
        raise OracleDegenerateSample("U is not a rotation matrix")
    if not np.isclose(abs(det_vh), 1):
        raise OracleDegenerateSample("V_h is not a rotation matrix")
    if np.isclose(-1, det_u):
        u = u * -1
    if np.isclose(-1, det_vh):
        vh = vh * -1
    if not np.isclose(0.0, s[-1]):
        raise OracleDegenerateSample("smallest singular value of E expected ~0")
    w = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=float)
    z = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0]], dtype=float)
    t_x = u @ z @ u.T
    t_1 = np.array([-t_x[1, 2], t_x[0, 2], -t_x[0, 1]])
    return u @ w.T @ vh, u @ w @ vh, t_1


def dlt_matrix(corr: np.ndarray, P1: np.ndarray, P2: np.ndarray) -> np.ndarray:
    """triangulation.py:23-30: rows ya*P1[2]-P1[1]; P1[0]-xa*P1[2]; yb*P2[2]-P2[1]; P2[0]-xb*P2[2]."""
    xa, ya, xb, yb = (corr[:, k][:, None] for k in range(4))
    return np.stack(
        [ya * P1[2, :] - P1[1, :], P1[0, :] - xa * P1[2, :],
         yb * P2[2, :] - P2[1, :], P2[0, :] - xb * P2[2, :]], axis=1)


def triangulate_dlt(corr: np.ndarray, P1: np.ndarray, P2: np.ndarray) -> np.ndarray:
    """triangulation.py:9-39 for M pairs: last right-singular vector of the 4x4 DLT matrix,
    de-homogenised by its last component (unguarded, :38)."""
    if corr.shape[0] == 0:
        return np.zeros((0, 3), dtype=np.float64)
    A = dlt_matrix(corr, P1, P2)
    _, _, vh = np.linalg.svd(A)
    x = vh[:, -1, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        return (x / x[:, 3:4])[:, :3]


def cheirality_pass(corr_n: np.ndarray, R: np.ndarray, t: np.ndarray,
                    distance_threshold: Optional[float] = None) -> np.ndarray:
    """eight_point.py:449-488 for M normalised pairs -> bool (M,)."""
    if distance_threshold is None:
        distance_threshold = DEFAULT_DISTANCE_THRESHOLD
    P1 = np.eye(4)
    P2 = np.eye(4)
    P2[:3, :3] = R
    P2[:3, 3] = t
    X = triangulate_dlt(corr_n, P1, P2)
    Xh = np.hstack([X, np.ones((X.shape[0], 1))])
    X2 = (Xh @ P2.T)[:, :3]
    with np.errstate(invalid="ignore"):
        ok = (X[:, 2] >= -CHEIRALITY_TOLERANCE) & (X2[:, 2] >= -CHEIRALITY_TOLERANCE)
        ok &= np.sqrt((X * X).sum(axis=1)) <= distance_threshold
    return ok


def recover_r_t(corr_n: np.ndarray, e: np.ndarray, distance_threshold: Optional[float] = None):
    """eight_point.py:181-242 including the vote quirk: votes = count_nonzero(index array), so a
    passing pair at index 0 is not counted (:228-230)."""
    R1, R2, t1 = recover_all_r_t(e)
    votes, poses, masks = [], [], []
    for R, t in itertools.product([R1, R2], [t1, -t1]):
        idx = np.nonzero(cheirality_pass(corr_n, R, t, distance_threshold))[0]
        masks.append(idx)
        votes.append(int(np.count_nonzero(idx)))
        poses.append((R, t))
    if 0 == np.count_nonzero(votes):
        raise OracleDegenerateSample("None of the transformations pass the cheirality check.")
    best = int(np.argmax(votes))
    return poses[best][0], poses[best][1], masks[best], votes


def triangulate_points(pix_a: np.ndarray, pix_b: np.ndarray, K: np.ndarray, cam2_T_cam1: np.ndarray):
    """triangulation.py:42-62 with pixel-unit 3x4 projection matrices."""
    if (3, 3) != K.shape:
        raise ValueError(f"Camera intrinsic matrix is not 3x3, actual shape: {K.shape}")
    K_ext = np.hstack((K, np.zeros((3, 1))))
    P1 = K_ext @ np.eye(4)
    P2 = K_ext @ (cam2_T_cam1 @ np.eye(4))
    return triangulate_dlt(pack_correspondences(pix_a, pix_b), P1, P2)


# --------------------------------------------------------------------------------------------------
# synthetic two-view scene — shared by tests, smoke and the cpu_baseline leg
# --------------------------------------------------------------------------------------------------
BENCH_K = np.array([[1520.4, 0.0, 302.32], [0.0, 1525.9, 246.87], [0.0, 0.0, 1.0]])


def euler_xy(deg_x: float, deg_y: float) -> np.ndarray:
    """Intrinsic rotation about X then Y (scipy ``Rotation.from_euler("XY", ...)``): Rx @ Ry."""
    ax, ay = np.radians(deg_x), np.radians(deg_y)
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    return rx @ ry


def synthetic_two_view(n: int, seed: int = 6, outlier_fraction: float = 0.3, noise_px: float = 0.5,
                       K: np.ndarray = BENCH_K):
    """Deterministic scene of SURVEY.md §8d.  Returns (pix_a (n,2), pix_b (n,2), K, R, t, is_outlier)."""
    rng = np.random.default_rng(seed)
    X = np.empty((n, 3))
    X[:, 0] = rng.uniform(-1.0, 1.0, n)
    X[:, 1] = rng.uniform(-1.0, 1.0, n)
    X[:, 2] = rng.uniform(4.0, 6.0, n)
    R = euler_xy(-5.0, -10.0)
    t = np.array([0.5, 0.05, 0.1])

    def project(Xc):
        uvw = Xc @ K.T
        return uvw[:, :2] / uvw[:, 2:3]

    pa = project(X) + rng.normal(0.0, noise_px, (n, 2))
    pb = project(X @ R.T + t) + rng.normal(0.0, noise_px, (n, 2))
    is_out = rng.random(n) < outlier_fraction
    width, height = 2.0 * K[0, 2], 2.0 * K[1, 2]
    rand_px = np.column_stack([rng.uniform(0, width, n), rng.uniform(0, height, n)])
    pb = np.where(is_out[:, None], rand_px, pb)
    return pa, pb, K, R, t, is_out
