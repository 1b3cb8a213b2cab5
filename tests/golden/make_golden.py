"""Generate golden vectors by running the REAL reference (``/root/reference``, imported unmodified).

Run in the build container only:  ``python tests/golden/make_golden.py``.
Writes ``tests/golden/*.npz`` (data only: inputs and the reference's outputs).  The reference never
travels to the GPU box; the committed ``.npz`` files do.

Harness-side shims (live only in this process, no reference file is touched):
  * ``np.Infinity = np.inf``            (reference ``matching.py:20`` predates NumPy 2)
  * stub ``transforms3d.affines.compose(T, R, Z) -> [[R diag(Z), T], [0, 1]]`` (``transforms.py:30``)
OpenCV is absent here, so scenes are projected with a plain pinhole model instead of
``cv.projectPoints`` (no distortion, identical geometry).
Produced under NumPy {np_version}; the reference pins 1.21.2 — differences are LAPACK rounding level.
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def _install_shims():
    np.Infinity = np.inf
    t3 = types.ModuleType("transforms3d")
    aff = types.ModuleType("transforms3d.affines")

    def compose(T, R, Z):
        out = np.eye(4)
        out[:3, :3] = np.asarray(R) @ np.diag(Z)
        out[:3, 3] = T
        return out

    aff.compose = compose
    t3.affines = aff
    sys.modules["transforms3d"] = t3
    sys.modules["transforms3d.affines"] = aff
    sys.path.insert(0, REFERENCE)


_install_shims()
import logging  # noqa: E402

logging.disable(logging.CRITICAL)

from lib.common.feature import Feature  # noqa: E402
from lib.epipolar import eight_point, epipolar_ransac, sed, triangulation  # noqa: E402
from lib.ransac import ransac as ref_ransac  # noqa: E402
from lib.ransac.ransac import ErrorAggregationMethod  # noqa: E402
from lib.transforms.transforms import Transform3D  # noqa: E402

sys.path.insert(0, REPO)
from oracle import sfm_oracle as orc  # noqa: E402  (scene generator + philox only)

assert eight_point.__file__.startswith(REFERENCE), eight_point.__file__

# silence tqdm progress bars of the reference
import tqdm  # noqa: E402

ref_ransac.tqdm.tqdm = lambda it, *a, **k: it


def feats(arr):
    return [Feature(x=float(p[0]), y=float(p[1])) for p in arr]


def camera_matrix(f, w, h):
    return np.array([[f, 0.0, w / 2.0], [0.0, f, h / 2.0], [0.0, 0.0, 1.0]])


def euler_xy(dx, dy):
    return orc.euler_xy(dx, dy)


def project(world_pts, cam_R_world, world_t_cam, K):
    cam = (world_pts - world_t_cam) @ cam_R_world.T
    uvw = cam @ K.T
    return uvw[:, :2] / uvw[:, 2:3]


def eight_point_scene():
    """Scene of the reference's EightPointFixture (test_epipolar.py:109-143)."""
    rng = np.random.default_rng(seed=6)
    pts = rng.random((8, 3), dtype=np.float64) + np.array([1.0, 0.0, 0.0])
    K = camera_matrix(50.0, 512, 256)
    c1 = np.array([1.5, 0.25, -1.0])
    R1 = np.eye(3)
    c2 = np.array([2.5, 0.1, -1.5])
    R2 = euler_xy(-20.0, -50.0)
    return rng, pts, K, c1, R1, c2, R2, project(pts, R1, c1, K), project(pts, R2, c2, K)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **arrays)
    print("wrote", path, {k: np.shape(v) for k, v in arrays.items()})


# --------------------------------------------------------------------------------------------------
def g1_eight_point_pipeline():
    _, pts, K, c1, R1, c2, R2, pa, pb = eight_point_scene()
    fa, fb = feats(pa), feats(pb)
    matches = eight_point.create_trivial_matches(8)
    F = eight_point.estimate_fundamental_mat(fa, fb, matches)
    E = eight_point.estimate_essential_mat(camera_matrix=K, features_a=fa, features_b=fb, matches=matches)
    Ra, Rb, t = eight_point._recover_all_r_t(E.copy())
    na = [eight_point.to_normalized_image_coords(f, K) for f in fa]
    nb = [eight_point.to_normalized_image_coords(f, K) for f in fb]
    R, tt, mask = eight_point._recover_r_t(na, nb, E)
    R_e2e, t_e2e, mask_e2e = eight_point.estimate_r_t(K, fa, fb, matches)
    # ground truth (test_epipolar.py:256-269)
    t_gt = R2 @ (c1 - c2)
    t_gt = t_gt / np.linalg.norm(t_gt)
    R_gt = R2 @ R1.T
    # analytic essential matrix [t]x R / e22
    tx = np.array([[0, -t_gt[2], t_gt[1]], [t_gt[2], 0, -t_gt[0]], [-t_gt[1], t_gt[0], 0]])
    E_gt = tx @ R_gt
    E_gt = E_gt / E_gt[2, 2]
    sed_perfect = np.array([
        sed.calculate_symmetric_epipolar_distance(a, b, E_gt) for a, b in zip(na, nb)])
    save("g1_eight_point", K=K, pix_a=pa, pix_b=pb, F=F, E=E, R1=Ra, R2=Rb, t1=t, R=R, t=tt,
         mask=mask, R_e2e=R_e2e, t_e2e=t_e2e, mask_e2e=mask_e2e, R_gt=R_gt, t_gt=t_gt, E_gt=E_gt,
         sed_perfect=sed_perfect)


def drive_reference_ransac(pix_a, pix_b, K, thr, method, min_extra, iters, shuffle_source):
    """Run the reference's fit_with_ransac on *indices* with recording wrappers around the reference's
    own fitter and scorer, so the per-hypothesis samples / models / scores are captured."""
    fa, fb = feats(pix_a), feats(pix_b)
    rec = dict(samples=[], models=[], scores=[], cur=None)

    def fitter(idx_list):
        rec["samples"].append(list(idx_list))
        pairs = [(fa[i], fb[i]) for i in idx_list]
        model = epipolar_ransac.eight_point_model_fitter(pairs, camera_matrix=K)
        rec["models"].append(np.array(model))
        rec["scores"].append({})
        return model

    def scorer(model, i):
        s = epipolar_ransac.calculate_sed_inlier_score(model, (fa[i], fb[i]), camera_matrix=K)
        rec["scores"][-1][i] = s
        return s

    saved = ref_ransac.random
    ref_ransac.random = shuffle_source
    try:
        model, inliers = ref_ransac.fit_with_ransac(
            list(range(len(fa))), 8, fitter, scorer, thr,
            min_num_extra_inliers=min_extra, error_aggregation_method=method, max_iterations=iters)
    finally:
        ref_ransac.random = saved
    return model, inliers, rec


class TableShuffle:
    """Stand-in for the ``random`` module inside the reference's ransac.py: ``shuffle`` arranges the
    data so that its first 8 entries are the next row of a given sample table and the rest follow in
    increasing order."""

    def __init__(self, table):
        self.table = table
        self.row = 0

    def shuffle(self, data):
        head = [int(v) for v in self.table[self.row]]
        self.row += 1
        rest = sorted(set(data) - set(head))
        data[:] = head + rest


def g2_ransac_known_answer():
    """test_epipolar.py:367-415: fixture + 2 noise matches, random.seed(5), thr 0.01, SUM, 100 iters."""
    rng, pts, K, c1, R1, c2, R2, pa, pb = eight_point_scene()
    max_x, min_x = pa[:, 0].max(), pa[:, 0].min()
    max_y, min_y = pa[:, 1].max(), pa[:, 1].min()
    nx = rng.random(4) * (max_x - min_x) + min_x
    ny = rng.random(4) * (max_y - min_y) + min_y
    pa2 = np.vstack([pa, np.column_stack([nx[:2], ny[:2]])])
    pb2 = np.vstack([pb, np.column_stack([nx[2:], ny[2:]])])
    fa, fb = feats(pa2), feats(pb2)
    matches = eight_point.create_trivial_matches(len(fa))
    random.seed(5)
    E, pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
        camera_matrix=K, features_a=fa, features_b=fb, matches=matches,
        sed_inlier_threshold=0.01, error_aggregation_method=ErrorAggregationMethod.SUM)
    inl_a = np.array([[p[0].x, p[0].y] for p in pairs])
    inl_b = np.array([[p[1].x, p[1].y] for p in pairs])
    # same run on indices to capture the sample table
    random.seed(5)
    E2, inliers, rec = drive_reference_ransac(
        pa2, pb2, K, 0.01, ErrorAggregationMethod.SUM, None, None, random)
    assert np.array_equal(E, E2)
    tx = R2 @ (c1 - c2)
    tx = tx / np.linalg.norm(tx)
    Tx = np.array([[0, -tx[2], tx[1]], [tx[2], 0, -tx[0]], [-tx[1], tx[0], 0]])
    E_gt = Tx @ (R2 @ R1.T)
    E_gt = E_gt / E_gt[2, 2]
    save("g2_ransac_seed5", K=K, pix_a=pa2, pix_b=pb2, E=E, inlier_a=inl_a, inlier_b=inl_b,
         inlier_idx=np.array(inliers), S=np.array(rec["samples"], dtype=np.int32),
         Eall=np.array(rec["models"]), E_gt=E_gt)


def g3_per_hypothesis():
    """Synthetic N=200 with 30 % outliers, H=50 hypotheses from the reference's own cumulative shuffle
    (random.seed(5)); per-hypothesis E / count / error for all four aggregation modes."""
    N, H, thr, min_extra = 200, 50, 1.5e-6, 10
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(N, seed=6)
    out = dict(K=K, pix_a=pa, pix_b=pb, thr=thr, min_extra=min_extra)
    for method in ErrorAggregationMethod:
        random.seed(5)
        E, inliers, rec = drive_reference_ransac(pa, pb, K, thr, method, min_extra, H, random)
        S = np.array(rec["samples"], dtype=np.int32)
        cnt = np.zeros(H, dtype=np.int32)
        err = np.full(H, np.nan)
        rest_scores = np.full((H, N), np.nan)
        for h in range(H):
            sc = rec["scores"][h]
            sample = set(rec["samples"][h])
            for i, s in sc.items():
                rest_scores[h, i] = s
            # ungated hypotheses never get their sample re-scored by ransac.py:77-79; fill those in
            # with the reference's own scorer so every hypothesis has a reference-computed error
            fa_, fb_ = feats(pa), feats(pb)
            for i in sample:
                if np.isnan(rest_scores[h, i]):
                    rest_scores[h, i] = epipolar_ransac.calculate_sed_inlier_score(
                        rec["models"][h], (fa_[i], fb_[i]), camera_matrix=K)
            rest = [i for i in range(N) if i not in sample and sc[i] <= thr]
            cnt[h] = len(rest)
        out[f"E_{method.value}"] = np.array(E)
        out[f"inliers_{method.value}"] = np.array(inliers)
        if method == ErrorAggregationMethod.RMS:
            out["S"] = S
            out["Eall"] = np.array(rec["models"])
            out["cnt"] = cnt
            out["sed_all"] = rest_scores
    # per-hypothesis aggregated error, through the reference's own _aggregate_error, on the data
    # order the reference used (sample first, then survivors in shuffled order)
    random.seed(5)
    perm = list(range(N))
    for method in ErrorAggregationMethod:
        errs = np.full(H, np.nan)
        random.seed(5)
        perm = list(range(N))
        for h in range(H):
            random.shuffle(perm)
            assert perm[:8] == list(out["S"][h])
            sed_h = out["sed_all"][h]
            surv = [i for i in perm[8:] if sed_h[i] <= thr]
            e_list = [sed_h[i] for i in perm[:8]] + [sed_h[i] for i in surv]
            errs[h] = ref_ransac._aggregate_error(e_list, method)
        out[f"err_{method.value}"] = errs
    save("g3_per_hypothesis", **out)


def g4_sed_values():
    rng = np.random.default_rng(11)
    E = rng.normal(size=(6, 3, 3))
    E[:, 2, 2] = 1.0
    pa = rng.uniform(-0.3, 0.3, (40, 2))
    pb = rng.uniform(-0.3, 0.3, (40, 2))
    vals = np.empty((6, 40))
    for k in range(6):
        for i in range(40):
            vals[k, i] = sed.calculate_symmetric_epipolar_distance(
                Feature(*pa[i]), Feature(*pb[i]), E[k])
    save("g4_sed", E=E, norm_a=pa, norm_b=pb, sed=vals)


def g5_cheirality():
    """Pose recovery on a noisy synthetic scene: per-candidate pass masks, quirk votes, chosen pose."""
    N = 120
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(N, seed=9, outlier_fraction=0.2)
    tn = t / np.linalg.norm(t)
    Tx = np.array([[0, -tn[2], tn[1]], [tn[2], 0, -tn[0]], [-tn[1], tn[0], 0]])
    E = Tx @ R
    E = E / E[2, 2]
    fa, fb = feats(pa), feats(pb)
    R_out, t_out, mask = eight_point.recover_r_t_from_e(e=E, camera_matrix=K, features_a=fa, features_b=fb)
    R_a, R_b, t1 = eight_point._recover_all_r_t(E.copy())
    na = [eight_point.to_normalized_image_coords(f, K) for f in fa]
    nb = [eight_point.to_normalized_image_coords(f, K) for f in fb]
    import itertools
    passes = np.zeros((4, N), dtype=np.uint8)
    votes = np.zeros(4, dtype=np.int64)
    for c, (Rc, tc) in enumerate(itertools.product([R_a, R_b], [t1, -t1])):
        for i in range(N):
            passes[c, i] = eight_point._cheirality_check(na[i], nb[i], Rc, tc)
        votes[c] = np.count_nonzero(np.nonzero(passes[c])[0])
    # tighter distance threshold variant
    R_d, t_d, mask_d = eight_point.recover_r_t_from_e(
        e=E, camera_matrix=K, features_a=fa, features_b=fb, distance_threshold=5.2)
    save("g5_cheirality", K=K, pix_a=pa, pix_b=pb, E=E, R=R_out, t=t_out, mask=mask,
         R1=R_a, R2=R_b, t1=t1, passes=passes, votes=votes, R_d=R_d, t_d=t_d, mask_d=mask_d,
         R_gt=R, t_gt=tn)


def g6_triangulate():
    """triangulate_points on clean + noisy + low-parallax points, and the reference's test_triangulate
    known answer (test_epipolar.py:418-496)."""
    N = 60
    pa, pb, K, R, t, _ = orc.synthetic_two_view(N, seed=4, outlier_fraction=0.0, noise_px=0.2)
    # low-parallax: push a few points far away (z ~ 200..2000 baseline units)
    rng = np.random.default_rng(21)
    far = np.column_stack([rng.uniform(-20, 20, 6), rng.uniform(-20, 20, 6), rng.uniform(200, 2000, 6)])

    def proj(Xc):
        uvw = Xc @ K.T
        return uvw[:, :2] / uvw[:, 2:3]

    pa = np.vstack([pa, proj(far)])
    pb = np.vstack([pb, proj(far @ R.T + t)])
    T = Transform3D.from_rmat_t(R, t)
    X = triangulation.triangulate_points(feats(pa), feats(pb), K, T)
    # object-array inputs as produced by np.take in apps/sfm.py:168-169
    Xobj = triangulation.triangulate_points(
        np.take(feats(pa), np.arange(len(pa))), np.take(feats(pb), np.arange(len(pb))), K, T)
    assert np.array_equal(X, Xobj)
    # known-answer case
    K2 = camera_matrix(50.0, 512, 256)
    world_pt = np.array([[0.0, 0.0, 10.0]])
    c1 = np.array([0.0, 0.0, 5.0])
    c2 = np.array([3.0, 0.0, 5.0])
    ay = np.radians(30.0)
    world_R_cam2 = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    cam1_T_world = Transform3D.from_rmat_t(np.eye(3), -c1).Tmat
    cam2_T_world = Transform3D.from_rmat_t(world_R_cam2.T, -c2).Tmat
    K_ext = np.hstack((K2, np.zeros((3, 1))))
    P1 = K_ext @ cam1_T_world
    P2 = K_ext @ cam2_T_world

    def proj_p(P):
        h = P @ np.append(world_pt[0], 1.0)
        return h[:2] / h[2]

    qa, qb = proj_p(P1), proj_p(P2)
    X_known = triangulation.triangulate_point_correspondence(Feature(*qa), Feature(*qb), P1, P2)
    save("g6_triangulate", K=K, pix_a=pa, pix_b=pb, cam2_T_cam1=T.Tmat, X=X,
         known_P1=P1, known_P2=P2, known_a=qa, known_b=qb, known_X=X_known)


def g7_degenerate():
    """test_epipolar.py:272-364: eight points on two planar rectangles -> EightPointCalculationError."""
    rect = np.array([[0, 0, 0], [1, 0, 0], [1, 0.5, 0], [0, 0.5, 0]], dtype=float)
    rect_b = rect + np.array([2.0, 0.0, 0.0])

    def rot_y(deg):
        a = np.radians(deg)
        return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])

    def rotate(r, Rm):
        c = r.mean(axis=0)
        return (r - c) @ Rm.T + c

    rect_a = rotate(rect, rot_y(-40.0))
    rect_b = rotate(rect_b, rot_y(40.0))
    pts = np.vstack([rect_a, rect_b])
    fov = 100
    base = np.radians((180 - fov) / 2.0)
    f = min(np.tan(base) * 512 / 2, np.tan(base) * 256 / 2)
    K = camera_matrix(f, 512, 256)
    c1 = np.array([1.5, 0.25, -1.0])
    c2 = np.array([2.5, 0.1, -1.5])
    R2 = euler_xy(-20.0, -50.0)
    # the reference test passes tvec = -camera position for BOTH cameras (test_epipolar.py:330-348)
    pa = project(pts, np.eye(3), c1, K)
    cam2 = pts @ R2.T - c2
    uvw = cam2 @ K.T
    pb = uvw[:, :2] / uvw[:, 2:3]
    raised = False
    try:
        eight_point.estimate_fundamental_mat(feats(pa), feats(pb), eight_point.create_trivial_matches(8))
    except eight_point.EightPointCalculationError:
        raised = True
    ca, cb = eight_point._get_matching_coordinates(feats(pa), feats(pb), eight_point.create_trivial_matches(8))
    na, _ = eight_point._normalize_coords(ca)
    nb, _ = eight_point._normalize_coords(cb)
    w = np.sort(np.real(np.linalg.eig(eight_point._get_yT_y(na, nb))[0]))
    save("g7_degenerate", pix_a=pa, pix_b=pb, raised=np.array(raised), sorted_w=w)


def g8_unit_vectors():
    """test_epipolar.py:49-106 helper known answers."""
    inp = np.array([[10.0, 10.0], [15.0, 10.0], [5.0, 10.0]])
    nc, T = eight_point._normalize_coords(inp)
    ycol = eight_point._get_y_col(np.array([2.0, 3.0]), np.array([7.0, 6.0]))
    K = camera_matrix(50.0, 512, 256)
    nf = eight_point.to_normalized_image_coords(Feature(x=50, y=60), K)
    rng = np.random.default_rng(3)
    ca = rng.uniform(-1, 1, (8, 2))
    cb = rng.uniform(-1, 1, (8, 2))
    na, Ta = eight_point._normalize_coords(ca)
    nb, Tb = eight_point._normalize_coords(cb)
    A = eight_point._get_yT_y(na, nb)
    save("g8_units", norm_in=inp, norm_out=nc, norm_T=T, ycol=ycol, K=K, nf=np.array([nf.x, nf.y]),
         ca=ca, cb=cb, na=na, nb=nb, Ta=Ta, Tb=Tb, yty=A)


def g9_explicit_table():
    """Reference driven by an explicit (philox) sample table at N=300, H=200: final E, ordered inlier
    indices, per-hypothesis models.  Demo-like RANSAC parameters (config.yaml:6-9)."""
    N, H, thr, min_extra = 300, 200, 1.5e-6, 10
    pa, pb, K, R, t, _ = orc.synthetic_two_view(N, seed=6)
    S = orc.philox_sample_table(5, 0, H, N)
    out = dict(K=K, pix_a=pa, pix_b=pb, S=S, thr=thr, min_extra=min_extra)
    for method in ErrorAggregationMethod:
        E, inliers, rec = drive_reference_ransac(pa, pb, K, thr, method, min_extra, H, TableShuffle(S))
        assert np.array_equal(np.array(rec["samples"]), S)
        out[f"E_{method.value}"] = np.array(E)
        out[f"inliers_{method.value}"] = np.array(inliers)
        best = [h for h in range(H) if np.array_equal(rec["models"][h], E)]
        out[f"best_{method.value}"] = np.array(best[0])
    out["Eall"] = np.array(rec["models"])
    save("g9_explicit_table", **out)


def g10_line_ransac():
    """lib/ransac/tests/test_ransac.py:70-123 known answer for the generic driver (2-point line fit)."""
    from math import sqrt
    line_start = np.array([4, 5])
    slope, n_line, dx = 0.6, 50, 0.3
    line_points = np.array([line_start + np.array([i * dx, i * slope * dx]) for i in range(n_line)]).reshape((-1, 2))
    rng = np.random.default_rng(seed=6)
    noise = rng.random(size=(25, 2))
    noise[:, 0] = noise[:, 0] * (line_points[:, 0].max() - line_points[:, 0].min()) + line_points[:, 0].min()
    noise[:, 1] = noise[:, 1] * (line_points[:, 1].max() - line_points[:, 1].min()) + line_points[:, 1].min()
    all_points = np.vstack([line_points, noise])

    def fitter(points):
        dxx = points[1][0] - points[0][0]
        if abs(dxx) <= 1e-6:
            return (1.0, 0.0, -points[0][0])
        s = (points[1][1] - points[0][1]) / dxx
        return (s, -1.0, points[0][1] - s * points[0][0])

    def scorer(m, p):
        return abs(m[0] * p[0] + m[1] * p[1] + m[2]) / sqrt(m[0] ** 2 + m[1] ** 2)

    random.seed(5)
    model, inliers = ref_ransac.fit_with_ransac(
        data=list(all_points), model_fit_data_count=2, model_fitter=fitter, inlier_scorer=scorer,
        inlier_threshold=0.2, min_num_extra_inliers=len(all_points) / 2,
        error_aggregation_method=ErrorAggregationMethod.RMS)
    save("g10_line_ransac", points=all_points, model=np.array(model), inliers=np.array(inliers))


def g11_matching():
    """Brute-force matcher (matching.py:36-118) with the real NCC / SSD score functions on synthetic uint8
    images, wired like apps/sfm.py:73-87 (closure around a functools.partial of calculate_ncc)."""
    import functools
    from lib.feature_matching import matching, ncc, ssd

    rng = np.random.default_rng(42)
    H, W = 72, 104
    base = rng.integers(0, 256, size=(H + 8, W + 8)).astype(np.float64)
    # smooth a little so neighbouring windows correlate, then quantise to uint8
    k = np.array([1.0, 2.0, 1.0]) / 4.0
    for _ in range(2):
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, base)
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, base)
    image_a = np.clip(base[4:4 + H, 4:4 + W], 0, 255).astype(np.uint8)
    image_b = np.clip(base[2:2 + H, 1:1 + W] + rng.normal(0, 2.0, (H, W)), 0, 255).astype(np.uint8)  # shifted + noise
    nA, nB = 40, 50
    fa = np.column_stack([rng.integers(0, W, nA), rng.integers(0, H, nA)]).astype(np.float64)
    fb = np.column_stack([rng.integers(0, W, nB), rng.integers(0, H, nB)]).astype(np.float64)
    fb[:25] = fa[:25] + np.array([3.0, 2.0])  # true correspondences for the first 25
    fa[30] = [2.0, 40.0]   # out of bounds for window 9, in bounds for window 5
    fb[40] = [W - 1.0, 5.0]
    fa[31] = fa[0]          # duplicate feature -> tied scores
    feats_a, feats_b = feats(fa), feats(fb)
    out = dict(image_a=image_a, image_b=image_b, feats_a=fa, feats_b=fb)

    def score_fn(full):
        def score(feature_a, feature_b):  # same shape as apps/sfm.py:_create_score_function
            return full(image_a, image_b, feature_a, feature_b)
        return score

    combos = {
        "none": None,
        "ratio": matching.ValidationStrategy.RATIO_TEST,
        "cross": {matching.ValidationStrategy.CROSSCHECK},
        "both": {matching.ValidationStrategy.RATIO_TEST, matching.ValidationStrategy.CROSSCHECK},
    }
    for metric, fn, ws in (("ncc", ncc.calculate_ncc, 9), ("ncc5", ncc.calculate_ncc, 5), ("ssd", ssd.calculate_ssd, 5)):
        full = functools.partial(fn, window_size=ws)
        if metric == "ssd":
            # ssd.py:31-35 subtracts and squares in the image dtype: on uint8 images the reference wraps modulo
            # 256.  The SSD vectors are therefore taken on float images (the reference's own test_ssd.py uses
            # wide integers), where the arithmetic is the plain one.
            ia, ib = image_a.astype(np.float64), image_b.astype(np.float64)
            full = functools.partial(lambda A, B, a, b, _f=full, _ia=ia, _ib=ib: _f(_ia, _ib, a, b))
        scores = np.array([[full(image_a, image_b, a, b) for b in feats_b] for a in feats_a], dtype=np.float64)
        out[f"scores_{metric}"] = scores
        for name, strat in combos.items():
            for thr in (0.7, 0.95):
                ms = matching.match_brute_force(feats_a, feats_b, score_fn(full),
                                                validation_strategies=strat, ratio_test_threshold=thr)
                out[f"matches_{metric}_{name}_{thr}"] = np.array(
                    [[m.a_index, m.b_index, m.match_score] for m in ms], dtype=np.float64).reshape(-1, 3)
    # the reference's own unit vectors (test_ncc.py, test_ssd.py, test_util.py)
    img = np.array([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [9, 8, 7, 6, 5], [4, 3, 2, 1, 0], [1, 2, 3, 4, 5]])
    out["unit_ncc_perfect"] = np.array(ncc.calculate_ncc(img, img.copy(), Feature(2, 2), Feature(2, 2), 5))
    out["unit_ncc_worst"] = np.array(ncc.calculate_ncc(img, -img, Feature(2, 2), Feature(2, 2), 5))
    np.random.seed(55)
    ra, rb, rs = [], [], []
    for _ in range(20):
        ia = np.random.rand(5, 5) + np.random.randint(-100, 100)
        ib = np.random.rand(5, 5) + np.random.randint(-100, 100)
        ra.append(ia); rb.append(ib)
        rs.append(ncc.calculate_ncc(ia, ib, Feature(2, 2), Feature(2, 2), 5))
    out["unit_ncc_random_a"] = np.array(ra)
    out["unit_ncc_random_b"] = np.array(rb)
    out["unit_ncc_random_scores"] = np.array(rs)
    save("g11_matching", **out)


def g12_harris():
    """Harris detector (harris_detector.py:11-113) on small synthetic images: the filled rectangle of the
    reference's test_harris_detector.py (drawn without OpenCV) and two random textured uint8 images."""
    from lib.common import correlate
    from lib.harris import harris_detector as harris

    harris.tqdm.trange = lambda n, **k: range(n)
    out = {}
    rect = np.zeros((100, 200), dtype=float)
    rect[25:76, 50:151] = 255.0  # cv.rectangle(background, (50, 25), (150, 75), 255, -1) fills inclusive bounds
    rng = np.random.default_rng(7)
    tex = rng.integers(0, 256, size=(48, 64)).astype(np.float64)
    kern = np.array([1.0, 2.0, 1.0]) / 4.0
    for _ in range(2):
        tex = np.apply_along_axis(lambda r: np.convolve(r, kern, mode="same"), 0, tex)
        tex = np.apply_along_axis(lambda r: np.convolve(r, kern, mode="same"), 1, tex)
    tex_u8 = np.clip(tex, 0, 255).astype(np.uint8)
    blobs = np.zeros((40, 56))
    for (cy, cx) in rng.integers(5, 35, size=(12, 2)):
        yy, xx = np.mgrid[0:40, 0:56]
        blobs += 200.0 * np.exp(-((yy - cy) ** 2 + (xx - cx * 1.4) ** 2) / 6.0)
    blobs_u8 = np.clip(blobs, 0, 255).astype(np.uint8)
    for name, img, n in (("rect", rect, 50), ("tex", tex_u8, 60), ("blobs", blobs_u8, 25)):
        corners = harris.detect_harris_corners(img, num_corners=n)
        corn = harris._calculate_cornerness_image(img, 2, 0.04)
        supp = corn.copy()
        supp[supp < 0] = 0.0
        harris._non_max_suppress(supp)
        out[f"{name}_image"] = img
        out[f"{name}_corners"] = np.array([[c.x, c.y] for c in corners]).reshape(-1, 2)
        out[f"{name}_cornerness"] = corn
        out[f"{name}_suppressed"] = supp
        out[f"{name}_sobel_x"] = harris._apply_sobel_x(img)
        out[f"{name}_sobel_y"] = harris._apply_sobel_y(img)
        out[f"{name}_n"] = np.array(n)
    img = np.array([[1, 5, 4, 3, 7], [2, 5, 7, 4, -10], [9, -5, 4, 3, 2]], dtype=float)  # test_correlate.py
    ker = np.array([[1, -2, 3], [2, 1, 0], [7, -5, 1]], dtype=float)
    out["cc_image"], out["cc_kernel"], out["cc_out"] = img, ker, correlate.cross_correlate(img, ker)
    out["cc_ones5"] = correlate.cross_correlate(np.ones((5, 10)), np.ones((5, 5)))
    save("g12_harris", **out)


def g13_refit():
    """N-point refit for the local-optimisation extension.  The reference itself only fits
    8 pairs (``estimate_fundamental_mat`` raises otherwise, eight_point.py:151-152, and ``_get_yT_y`` asserts 8,
    :365), so the M-point fit is COMPOSED here from the reference's own helpers in the order
    ``estimate_fundamental_mat`` uses them (:154-166): ``_normalize_coords`` per image, the ``np.outer`` accumulation
    of ``_get_y_col`` columns (the loop of ``_get_yT_y`` without its length assert), ``_compute_f_est``,
    ``_enforce_fundamental_mat_constraints``, ``T2.T @ e @ T1``, ``e /= e[2, 2]``."""
    out = {}
    for name, n, m, seed in (("a", 400, 40, 11), ("b", 3000, 1500, 12), ("c", 64, 8, 13)):
        pa, pb, K, R, t, is_out = orc.synthetic_two_view(n, seed=seed)
        idx = np.nonzero(~is_out)[0][:m]
        na = np.array([[f.x, f.y] for f in (eight_point.to_normalized_image_coords(f, K) for f in feats(pa[idx]))])
        nb = np.array([[f.x, f.y] for f in (eight_point.to_normalized_image_coords(f, K) for f in feats(pb[idx]))])
        ca, T1 = eight_point._normalize_coords(na)
        cb, T2 = eight_point._normalize_coords(nb)
        yty = np.zeros((9, 9), dtype=np.float64)
        for i in range(len(ca)):
            col = eight_point._get_y_col(ca[i, :], cb[i, :])
            yty += np.outer(col, col)
        e = eight_point._enforce_fundamental_mat_constraints(eight_point._compute_f_est(yty))
        e = T2.T @ e @ T1
        e /= e[2, 2]
        out[f"{name}_pix_a"], out[f"{name}_pix_b"], out[f"{name}_K"] = pa, pb, K
        out[f"{name}_idx"], out[f"{name}_E"], out[f"{name}_yty"] = idx, e, yty
        if m == 8:  # with exactly eight pairs the composition must BE the reference's public entry point
            E8 = eight_point.estimate_essential_mat(camera_matrix=K, features_a=feats(pa[idx]),
                                                    features_b=feats(pb[idx]),
                                                    matches=eight_point.create_trivial_matches(8))
            assert np.array_equal(E8, e), np.abs(E8 - e).max()
    save("g13_refit", **out)


def g14_ssd_integer():
    """ssd.py:31-36 on INTEGER images: the reference subtracts and squares in the image dtype (uint8 wraps modulo 256,
    int16 modulo 65536 into the signed range, ...), sums in int64 / uint64 and divides by the window size in float64.
    Same images, features and matcher wiring as g11 — taken on the uint8 images themselves, no float cast — plus
    small known-answer grids for every integer dtype, for a mixed-dtype pair (NumPy promotes) and for bool (TypeError)."""
    import functools
    from lib.feature_matching import matching, ssd

    d = np.load(os.path.join(HERE, "g11_matching.npz"))
    image_a, image_b, fa, fb = d["image_a"], d["image_b"], d["feats_a"], d["feats_b"]
    assert image_a.dtype == np.uint8 and image_b.dtype == np.uint8
    feats_a, feats_b = feats(fa), feats(fb)
    out = dict(image_a=image_a, image_b=image_b, feats_a=fa, feats_b=fb)

    def score_fn(full):
        def score(feature_a, feature_b):  # same shape as apps/sfm.py:_create_score_function
            return full(image_a, image_b, feature_a, feature_b)
        return score

    combos = {
        "none": None,
        "ratio": matching.ValidationStrategy.RATIO_TEST,
        "cross": {matching.ValidationStrategy.CROSSCHECK},
        "both": {matching.ValidationStrategy.RATIO_TEST, matching.ValidationStrategy.CROSSCHECK},
    }
    for ws in (5, 9):
        full = functools.partial(ssd.calculate_ssd, window_size=ws)
        with np.errstate(over="ignore"):
            out[f"scores_u8_w{ws}"] = np.array([[full(image_a, image_b, a, b) for b in feats_b] for a in feats_a],
                                               dtype=np.float64)
            for name, strat in combos.items():
                for thr in (0.7, 0.95):
                    ms = matching.match_brute_force(feats_a, feats_b, score_fn(full), validation_strategies=strat,
                                                    ratio_test_threshold=thr)
                    out[f"matches_u8_w{ws}_{name}_{thr}"] = np.array(
                        [[m.a_index, m.b_index, m.match_score] for m in ms], dtype=np.float64).reshape(-1, 3)
    # every integer dtype: random images over the dtype's full range (differences and squares wrap all the time),
    # a 6 x 7 grid of features, windows of 3 and 5
    rng = np.random.default_rng(1414)
    H, W = 24, 30
    gx, gy = np.meshgrid(np.array([0, 2, 5, 11, 17, 27, 29]), np.array([1, 2, 7, 12, 21, 23]))
    grid = np.column_stack([gx.ravel(), gy.ravel()]).astype(np.float64)
    out["grid_feats"] = grid
    gf = feats(grid)
    for name in ("uint8", "int8", "uint16", "int16", "uint32", "int32", "uint64", "int64"):
        dt = np.dtype(name)
        info = np.iinfo(dt)
        ia = rng.integers(info.min, info.max, size=(H, W), dtype=dt, endpoint=True)
        ib = rng.integers(info.min, info.max, size=(H, W), dtype=dt, endpoint=True)
        out[f"{name}_a"], out[f"{name}_b"] = ia, ib
        for ws in (3, 5):
            with np.errstate(over="ignore"):
                out[f"{name}_scores_w{ws}"] = np.array(
                    [[ssd.calculate_ssd(ia, ib, a, b, ws) for b in gf[::3]] for a in gf], dtype=np.float64)
    # mixed dtypes: NumPy promotes uint8 - int16 to int16, uint8 - int8 to int16, uint32 - int32 to int64, uint64 - int64 to float64
    for na, nb in (("uint8", "int16"), ("uint8", "int8"), ("uint32", "int32"), ("uint64", "int64"), ("uint8", "float64")):
        ia, ib = out[f"{na}_a"], (out[f"{nb}_b"] if nb != "float64" else out["uint8_b"].astype(np.float64) + 0.25)
        with np.errstate(over="ignore"):
            out[f"mixed_{na}_{nb}_scores"] = np.array(
                [[ssd.calculate_ssd(ia, ib, a, b, 3) for b in gf[::3]] for a in gf], dtype=np.float64)
    out["mixed_uint8_float64_b"] = out["uint8_b"].astype(np.float64) + 0.25
    # the reference's own test_ssd.py vector (wide integers: platform int) and bool images (NumPy refuses `-` on bool)
    img = np.array([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [9, 8, 7, 6, 5], [4, 3, 2, 1, 0], [1, 2, 3, 4, 5]])
    out["unit_int_image"] = img
    out["unit_int_same"] = np.array(ssd.calculate_ssd(img, img.copy(), Feature(2, 2), Feature(2, 2), 5))
    out["unit_int_negated"] = np.array(ssd.calculate_ssd(img, -img, Feature(2, 2), Feature(2, 2), 5))
    try:
        ssd.calculate_ssd(img > 4, img > 5, Feature(2, 2), Feature(2, 2), 3)
        out["bool_raises"] = np.array(0)
    except TypeError:
        out["bool_raises"] = np.array(1)
    save("g14_ssd_integer", **out)


if __name__ == "__main__":
    everything = [g1_eight_point_pipeline, g2_ransac_known_answer, g3_per_hypothesis, g4_sed_values, g5_cheirality,
                  g6_triangulate, g7_degenerate, g8_unit_vectors, g9_explicit_table, g10_line_ransac, g11_matching,
                  g12_harris, g13_refit, g14_ssd_integer]
    wanted = sys.argv[1:]  # e.g. ``make_golden.py g13`` regenerates one fixture
    for fn in everything:
        if not wanted or fn.__name__.split("_")[0] in wanted:
            fn()
    print("numpy", np.__version__)
