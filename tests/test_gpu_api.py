"""GPU tests of the drop-in Python surface (``lib.epipolar`` / ``lib.ransac`` import paths), written to
read like the reference's own ``lib/epipolar/tests/test_epipolar.py`` with OpenCV replaced by golden
vectors of the real reference and by analytic ground truth."""
import os
import random

import numpy as np
import pytest

from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu(native_lib):
    from structure_from_motion_amd import device

    device.require_gpu()


from lib.common.feature import Feature  # noqa: E402
from lib.epipolar import eight_point, epipolar_ransac  # noqa: E402
from lib.epipolar.sed import calculate_symmetric_epipolar_distance  # noqa: E402
from lib.epipolar.triangulation import triangulate_point_correspondence, triangulate_points  # noqa: E402
from lib.feature_matching.matching import Match  # noqa: E402
from lib.ransac.ransac import ErrorAggregationMethod  # noqa: E402
from lib.transforms.transforms import Transform3D  # noqa: E402


def feats(arr):
    return [Feature(x=float(p[0]), y=float(p[1])) for p in arr]


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def test_epipolar_pipeline(golden):
    """reference test_epipolar.py:151-269."""
    d = golden("g1_eight_point")
    features_1, features_2 = feats(d["pix_a"]), feats(d["pix_b"])
    matches = eight_point.create_trivial_matches(len(features_1))
    f = eight_point.estimate_fundamental_mat(features_1, features_2, matches)
    np.testing.assert_almost_equal(d["F"], f, decimal=5)  # the reference's bar (vs OpenCV)
    assert rel(f, d["F"]) <= 1e-6                          # north_star bar vs the reference itself

    e = eight_point.estimate_essential_mat(
        camera_matrix=d["K"], features_a=features_1, features_b=features_2, matches=matches)
    np.testing.assert_almost_equal(d["E_gt"], e, decimal=5)
    assert rel(e, d["E"]) <= 1e-6

    R1, R2, t = eight_point._recover_all_r_t(e)

    def close(x, y):
        return np.allclose(x, y, atol=1e-4)

    assert close(d["t1"], t) or close(-d["t1"], t)
    assert (close(d["R1"], R1) and close(d["R2"], R2)) or (close(d["R2"], R1) and close(d["R1"], R2))

    na = [eight_point.to_normalized_image_coords(f_, d["K"]) for f_ in features_1]
    nb = [eight_point.to_normalized_image_coords(f_, d["K"]) for f_ in features_2]
    R, tt, mask = eight_point._recover_r_t(na, nb, e)
    assert np.all(np.array(range(len(na))) == mask)
    assert mask.dtype == np.int64
    R2e, t2e, mask2 = eight_point.estimate_r_t(d["K"], features_1, features_2, matches)
    np.testing.assert_equal(mask, mask2)
    np.testing.assert_allclose(R2e, R)
    np.testing.assert_allclose(t2e, tt)
    np.testing.assert_allclose(d["t_gt"], tt / abs(np.linalg.norm(tt)), atol=1e-5, rtol=0.0)
    np.testing.assert_allclose(d["R_gt"], R, atol=1e-5, rtol=0.0)
    np.testing.assert_allclose(R, d["R"], atol=1e-9)
    np.testing.assert_allclose(tt, d["t"], atol=1e-9)


def test_estimate_essential_matrix_degenerate(golden):
    """reference test_epipolar.py:272-364."""
    d = golden("g7_degenerate")
    with pytest.raises(eight_point.EightPointCalculationError):
        eight_point.estimate_fundamental_mat(feats(d["pix_a"]), feats(d["pix_b"]),
                                             eight_point.create_trivial_matches(8))


def test_wrong_match_count_raises(golden):
    d = golden("g1_eight_point")
    with pytest.raises(ValueError):
        eight_point.estimate_fundamental_mat(feats(d["pix_a"]), feats(d["pix_b"]),
                                             eight_point.create_trivial_matches(7))
    with pytest.raises(ValueError):
        epipolar_ransac.eight_point_model_fitter([(Feature(0, 0), Feature(1, 1))] * 7, camera_matrix=d["K"])
    with pytest.raises(ValueError):
        eight_point.estimate_r_t(d["K"], [], [], [])


def test_estimate_essential_mat_with_ransac(golden):
    """reference test_epipolar.py:367-415: random.seed(5), thr 0.01, SUM, default 100 iterations.
    With the pyshuffle sampler the hypothesis samples are the reference's, so E and the ordered
    inlier list must equal the real reference's output."""
    d = golden("g2_ransac_seed5")
    features_1, features_2 = feats(d["pix_a"]), feats(d["pix_b"])
    matches = eight_point.create_trivial_matches(len(features_1))
    random.seed(5)
    e, inlier_feature_pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
        camera_matrix=d["K"], features_a=features_1, features_b=features_2, matches=matches,
        sed_inlier_threshold=0.01, error_aggregation_method=ErrorAggregationMethod.SUM)
    np.testing.assert_almost_equal(d["E_gt"], e, decimal=5)
    assert rel(e, d["E"]) <= 1e-6
    assert len(inlier_feature_pairs) == len(d["inlier_a"])
    # This fixture is noise-free: the inliers' SEDs are ~1e-30 (pure rounding), so WHICH of the equivalent
    # all-inlier samples wins is decided by rounding noise and only the inlier *set* is well defined
    # (the ordered list is pinned on the noisy fixture in test_ransac_per_method_equals_reference).
    got_a = np.array(sorted([p[0].x, p[0].y] for p in inlier_feature_pairs))
    got_b = np.array(sorted([p[1].x, p[1].y] for p in inlier_feature_pairs))
    np.testing.assert_array_equal(got_a, np.array(sorted(d["inlier_a"].tolist())))
    np.testing.assert_array_equal(got_b, np.array(sorted(d["inlier_b"].tolist())))
    # the returned pairs are copies, inputs untouched (ransac.py:59 deepcopy)
    assert all(p[0] is not f for p in inlier_feature_pairs for f in features_1)
    # the global random state advanced exactly as 100 reference shuffles would
    after = random.random()
    random.seed(5)
    perm = list(range(len(features_1)))
    for _ in range(100):
        random.shuffle(perm)
    assert random.random() == after


def test_default_sampler_is_the_reference_stream_above_the_old_switch(monkeypatch):
    """2 000 matches x 6 000 iterations = 1.2e7 draws — above the size where rounds 1-2 silently switched the default to
    the device sampler.  With the environment untouched the drop-in call must still draw the reference's samples
    (cumulative random.shuffle from the global state, ransac.py:59-64): winner, E and the ORDERED inlier list equal the
    oracle driven by CPython's own random.shuffle, and the global random state ends where 6 000 reference shuffles
    leave it."""
    monkeypatch.delenv("SFM_SAMPLER", raising=False)
    monkeypatch.delenv("SFM_SEED", raising=False)
    n, h, thr, min_extra = 2000, 6000, 1.5e-6, 10
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=8)
    matches = eight_point.create_trivial_matches(n)
    random.seed(77)
    e, pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
        K, feats(pa), feats(pb), matches, thr, min_num_extra_inliers=min_extra,
        error_aggregation_method=ErrorAggregationMethod.RMS, max_iterations=h)
    after = random.random()
    random.seed(77)
    perm = list(range(n))
    S = np.empty((h, 8), dtype=np.int32)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    for it in range(h):
        random.shuffle(perm)
        S[it] = perm[:8]
    assert random.random() == after          # the call advanced the global stream exactly like the reference's loop
    ref = orc.ransac_essential(corr, S, thr, min_extra, orc.RMS)
    assert ref["best"] >= 0 and rel(e, ref["E"]) <= 1e-6
    # the reference's list order: the winning iteration's sample, then the survivors in that iteration's shuffled order
    random.seed(77)
    perm = list(range(n))
    for it in range(ref["best"] + 1):
        random.shuffle(perm)
    order = orc.inlier_indices(corr, ref["E"], np.array(perm[:8]), thr, rest_order=perm[8:])
    np.testing.assert_array_equal(np.array([[p[0].x, p[0].y] for p in pairs]), pa[order])
    np.testing.assert_array_equal(np.array([[p[1].x, p[1].y] for p in pairs]), pb[order])


def test_ransac_per_method_equals_reference(golden):
    d = golden("g3_per_hypothesis")
    features_1, features_2 = feats(d["pix_a"]), feats(d["pix_b"])
    matches = eight_point.create_trivial_matches(len(features_1))
    for method in ErrorAggregationMethod:
        random.seed(5)
        e, pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
            d["K"], features_a=features_1, features_b=features_2, matches=matches,
            sed_inlier_threshold=float(d["thr"]), error_aggregation_method=method,
            min_num_extra_inliers=int(d["min_extra"]), max_iterations=50)
        assert rel(e, d["E_" + method.value]) <= 1e-6
        idx = d["inliers_" + method.value]
        np.testing.assert_array_equal(np.array([[p[0].x, p[0].y] for p in pairs]), d["pix_a"][idx])


def test_ransac_no_model_and_short_input(golden):
    d = golden("g3_per_hypothesis")
    features_1, features_2 = feats(d["pix_a"]), feats(d["pix_b"])
    matches = eight_point.create_trivial_matches(len(features_1))
    random.seed(1)
    with pytest.raises(ValueError, match="No model could be found with at least 208 inliers"):
        epipolar_ransac.estimate_essential_mat_with_ransac(
            d["K"], features_1, features_2, matches, 1.5e-6, min_num_extra_inliers=200, max_iterations=20)
    with pytest.raises(ValueError):
        epipolar_ransac.estimate_essential_mat_with_ransac(d["K"], features_1, features_2, matches[:5], 1.5e-6)


def test_ransac_degenerate_sample_policy(golden, monkeypatch):
    """A degenerate eight-tuple aborts the call like the reference (Q4); SFM_DEGENERATE=skip ignores it."""
    d = golden("g7_degenerate")
    g = golden("g1_eight_point")
    fa = feats(d["pix_a"])
    fb = feats(d["pix_b"])
    matches = eight_point.create_trivial_matches(8)
    random.seed(0)
    with pytest.raises(eight_point.EightPointCalculationError):
        epipolar_ransac.estimate_essential_mat_with_ransac(g["K"], fa, fb, matches, 0.01, max_iterations=3)
    monkeypatch.setenv("SFM_DEGENERATE", "skip")
    with pytest.raises(ValueError):  # every hypothesis is degenerate -> no model
        epipolar_ransac.estimate_essential_mat_with_ransac(g["K"], fa, fb, matches, 0.01, max_iterations=3)


def test_ransac_philox_sampler(monkeypatch, caplog):
    monkeypatch.setenv("SFM_SAMPLER", "philox")
    monkeypatch.setenv("SFM_SEED", "5")
    n, h = 400, 300
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=6)
    matches = eight_point.create_trivial_matches(n)
    import logging

    with caplog.at_level(logging.DEBUG, logger="structure_from_motion_amd.epipolar._engine"):
        e, pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
            K, feats(pa), feats(pb), matches, 1.5e-6, min_num_extra_inliers=10, max_iterations=h)
    # one log line per call, never per hypothesis
    lines = [r.getMessage() for r in caplog.records if r.name.endswith("epipolar._engine")]
    assert len(lines) == 1 and f"{n} matches x {h} hypotheses (philox sampler)" in lines[0], lines
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    ref = orc.ransac_essential(corr, orc.philox_sample_table(5, 0, h, n), 1.5e-6, 10, orc.RMS)
    assert rel(e, ref["E"]) <= 1e-6
    np.testing.assert_array_equal(np.array([[p[0].x, p[0].y] for p in pairs]), pa[ref["inliers"]])


def test_match_indirection():
    """matches index into the feature lists (epipolar_ransac.py:55-57)."""
    n = 60
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=2, outlier_fraction=0.1)
    rng = np.random.default_rng(0)
    perm_a, perm_b = rng.permutation(n), rng.permutation(n)
    fa = feats(pa[perm_a])
    fb = feats(pb[perm_b])
    inv_a, inv_b = np.argsort(perm_a), np.argsort(perm_b)
    matches = [Match(a_index=int(inv_a[i]), b_index=int(inv_b[i])) for i in range(n)]
    random.seed(3)
    e1, _ = epipolar_ransac.estimate_essential_mat_with_ransac(K, fa, fb, matches, 1.5e-6, max_iterations=40)
    random.seed(3)
    e2, _ = epipolar_ransac.estimate_essential_mat_with_ransac(
        K, feats(pa), feats(pb), eight_point.create_trivial_matches(n), 1.5e-6, max_iterations=40)
    np.testing.assert_array_equal(e1, e2)


def test_triangulate(golden):
    """reference test_epipolar.py:418-496."""
    d = golden("g6_triangulate")
    est = triangulate_point_correspondence(Feature(*d["known_a"]), Feature(*d["known_b"]), d["known_P1"], d["known_P2"])
    np.testing.assert_allclose([0.0, 0.0, 10.0], est, atol=1e-10, rtol=0)
    T = Transform3D(d["cam2_T_cam1"].copy())
    X = triangulate_points(feats(d["pix_a"]), feats(d["pix_b"]), d["K"], T)
    assert X.shape == (66, 3) and X.dtype == np.float64
    assert np.max(np.abs(X[:60] - d["X"][:60]) / np.abs(d["X"][:60])) <= 1e-6
    # NumPy object arrays of Feature, as apps/sfm.py:168-169 passes them
    idx = np.arange(0, 60, 3)
    X2 = triangulate_points(np.take(feats(d["pix_a"]), idx), np.take(feats(d["pix_b"]), idx), d["K"], T)
    np.testing.assert_array_equal(X2, X[idx])
    with pytest.raises(ValueError):
        triangulate_points(feats(d["pix_a"]), feats(d["pix_b"]), np.eye(4), T)
    assert triangulate_points([], [], d["K"], T).shape == (0, 3)
    # 4x4 camera matrices are accepted (only rows 0..2 are used), as in _cheirality_check
    P1 = np.vstack([d["known_P1"], [0, 0, 0, 1]])
    P2 = np.vstack([d["known_P2"], [0, 0, 0, 1]])
    np.testing.assert_array_equal(
        triangulate_point_correspondence(Feature(*d["known_a"]), Feature(*d["known_b"]), P1, P2), est)


def test_calculate_symmetric_epipolar_distance(golden):
    """reference test_epipolar.py:501-515: perfect correspondences under the true E."""
    d = golden("g1_eight_point")
    for p1, p2 in zip(d["pix_a"], d["pix_b"]):
        sed = calculate_symmetric_epipolar_distance(
            eight_point.to_normalized_image_coords(Feature(*p1), d["K"]),
            eight_point.to_normalized_image_coords(Feature(*p2), d["K"]), d["E_gt"])
        assert isinstance(sed, float) and sed < 1e-20
    s = epipolar_ransac.calculate_sed_inlier_score(d["E"], (Feature(*d["pix_a"][0]), Feature(*d["pix_b"][0])), d["K"])
    assert 0.0 <= s < 1e-15


def test_recover_r_t_from_e_golden(golden):
    d = golden("g5_cheirality")
    R, t, mask = eight_point.recover_r_t_from_e(
        e=d["E"], camera_matrix=d["K"], features_a=feats(d["pix_a"]), features_b=feats(d["pix_b"]))
    np.testing.assert_allclose(R, d["R"], atol=1e-9)
    np.testing.assert_allclose(t, d["t"], atol=1e-9)
    np.testing.assert_array_equal(mask, d["mask"])
    Rd, td, maskd = eight_point.recover_r_t_from_e(
        e=d["E"], camera_matrix=d["K"], features_a=feats(d["pix_a"]), features_b=feats(d["pix_b"]),
        distance_threshold=5.2)
    np.testing.assert_array_equal(maskd, d["mask_d"])
    # np.take on the mask, as apps/sfm.py:168 does
    kept = np.take(feats(d["pix_a"]), mask)
    assert len(kept) == len(d["mask"])
    with pytest.raises(eight_point.EightPointCalculationError):
        eight_point.recover_r_t_from_e(e=np.eye(3), camera_matrix=d["K"], features_a=feats(d["pix_a"]),
                                       features_b=feats(d["pix_b"]))


def test_vote_quirk_index_zero(golden):
    """Q9: a passing pair at index 0 is not counted; a single pair at index 0 therefore raises."""
    d = golden("g1_eight_point")
    fa, fb = feats(d["pix_a"][:1]), feats(d["pix_b"][:1])
    with pytest.raises(eight_point.EightPointCalculationError, match="cheirality"):
        eight_point.recover_r_t_from_e(e=d["E"], camera_matrix=d["K"], features_a=fa, features_b=fb)
    R, t, mask = eight_point.recover_r_t_from_e(
        e=d["E"], camera_matrix=d["K"], features_a=feats(d["pix_a"][:2]), features_b=feats(d["pix_b"][:2]))
    np.testing.assert_array_equal(mask, [0, 1])
    assert eight_point._cheirality_check(
        eight_point.to_normalized_image_coords(fa[0], d["K"]),
        eight_point.to_normalized_image_coords(fb[0], d["K"]), d["R"], d["t"]) is True


def test_sfm_call_sequence():
    """The three hot-path calls in the order and style of apps/sfm.py:110-186 on a synthetic pair."""
    n = 300
    pa, pb, K, R_gt, t_gt, is_out = orc.synthetic_two_view(n, seed=6)
    features_a, features_b = feats(pa), feats(pb)
    matches = eight_point.create_trivial_matches(n)
    random.seed(5)
    e, inlier_feature_pairs = epipolar_ransac.estimate_essential_mat_with_ransac(
        K, features_a=features_a, features_b=features_b, matches=matches, sed_inlier_threshold=1.5e-6,
        error_aggregation_method=ErrorAggregationMethod.RMS, min_num_extra_inliers=10, max_iterations=2000)
    inlier_features_a = [pair[0] for pair in inlier_feature_pairs]
    inlier_features_b = [pair[1] for pair in inlier_feature_pairs]
    r, t, inlier_mask = eight_point.recover_r_t_from_e(
        e=e, camera_matrix=K, features_a=inlier_features_a, features_b=inlier_features_b)
    cam2_T_cam1 = Transform3D.from_rmat_t(r, t)
    inlier_features_a = np.take(inlier_features_a, inlier_mask)
    inlier_features_b = np.take(inlier_features_b, inlier_mask)
    pts = triangulate_points(inlier_features_a, inlier_features_b, intrinsic_camera_matrix=K, cam2_T_cam1=cam2_T_cam1)
    assert pts.shape == (len(inlier_mask), 3)
    # against the oracle run on the same samples (pyshuffle replay of seed 5)
    random.seed(5)
    S, _ = orc.pyshuffle_sample_table(n, 2000)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    ref = orc.ransac_essential(corr, S, 1.5e-6, 10, orc.RMS)
    assert rel(e, ref["E"]) <= 1e-6
    assert len(inlier_feature_pairs) == len(ref["inliers"])
    # pose and structure against the oracle on the same inliers (sample first, then shuffled survivors)
    inl = np.array([[p[0].x, p[0].y, p[1].x, p[1].y] for p in inlier_feature_pairs])
    corr_inl = orc.pack_correspondences(orc.to_normalized_image_coords(inl[:, 0:2], K),
                                        orc.to_normalized_image_coords(inl[:, 2:4], K))
    R_o, t_o, mask_o, _ = orc.recover_r_t(corr_inl, ref["E"])
    np.testing.assert_allclose(r, R_o, atol=1e-6)
    np.testing.assert_allclose(t, t_o, atol=1e-6)
    np.testing.assert_array_equal(inlier_mask, mask_o)
    T = np.eye(4)
    T[:3, :3] = R_o
    T[:3, 3] = t_o
    pts_o = orc.triangulate_points(inl[mask_o, 0:2], inl[mask_o, 2:4], K, T)
    assert np.max(np.abs(pts - pts_o) / np.linalg.norm(pts_o, axis=1, keepdims=True)) <= 1e-6
    assert np.median(pts[:, 2]) > 0


# ------------------------------------------------------------------------------------------------------
# the reference's private fit helpers (reference test_epipolar.py:30-85) and the fused stages behind them
# ------------------------------------------------------------------------------------------------------
def test_get_matching_coordinates():
    """reference test_epipolar.py:30-46."""
    features_a = [Feature(256, 128), Feature(128, 64)]
    features_b = [Feature(32, 64), Feature(16, 8)]
    matches = [Match(0, 1), Match(1, 0)]
    coords_a, coords_b = eight_point._get_matching_coordinates(features_a, features_b, matches)
    np.testing.assert_allclose(np.array([[256, 128], [128, 64]]), coords_a)
    np.testing.assert_allclose(np.array([[16, 8], [32, 64]]), coords_b)


def test_normalize_coords(golden):
    """reference test_epipolar.py:49-61, plus the real reference's values (golden g8)."""
    input_coords = np.array([[10, 10], [15, 10], [5, 10]])
    normalized_coords, t = eight_point._normalize_coords(input_coords)
    expected_distance = np.sqrt(2.0) * 3.0 / 2.0
    np.testing.assert_allclose(np.array([[0, 0], [expected_distance, 0], [-expected_distance, 0]]),
                               normalized_coords, atol=1e-15)
    homo = np.hstack([normalized_coords, np.ones((3, 1))])
    np.testing.assert_allclose(input_coords, (homo @ np.linalg.inv(t).T)[:, :-1])
    d = golden("g8_units")
    np.testing.assert_allclose(normalized_coords, d["norm_out"], rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose(t, d["norm_T"], rtol=1e-15)
    na, Ta = eight_point._normalize_coords(d["ca"])
    np.testing.assert_allclose(na, d["na"], rtol=2e-16, atol=1e-16)
    np.testing.assert_allclose(Ta, d["Ta"], rtol=2e-16, atol=1e-16)


def test_get_y_col(golden):
    """reference test_epipolar.py:64-85."""
    y_col = eight_point._get_y_col(np.array([2.0, 3.0]), np.array([7.0, 6.0]))
    assert (9,) == y_col.shape
    np.testing.assert_allclose(np.array([14.0, 21.0, 7.0, 12.0, 18.0, 6.0, 2.0, 3.0, 1.0]), y_col)
    np.testing.assert_array_equal(y_col, golden("g8_units")["ycol"])


def test_fit_stages_against_reference_golden(golden):
    """Y^T Y bit-identical to the real reference's _get_yT_y; f_est / rank-2 step against LAPACK."""
    d = golden("g8_units")
    yty = eight_point._get_yT_y(d["na"], d["nb"])
    np.testing.assert_array_equal(yty, d["yty"])                       # bit-exact accumulation order
    f_est = eight_point._compute_f_est(d["yty"])
    w, v = np.linalg.eig(d["yty"])
    ref = np.real(v[:, np.argmin(np.abs(w))]).reshape(3, 3)
    ref = ref * np.sign(ref.ravel()[np.argmax(np.abs(ref))]) * np.sign(f_est.ravel()[np.argmax(np.abs(ref))])
    np.testing.assert_allclose(f_est, ref, atol=1e-12)
    assert abs(np.linalg.norm(f_est) - 1.0) <= 1e-14
    f2 = eight_point._enforce_fundamental_mat_constraints(f_est)
    u, s_, vh = np.linalg.svd(f_est)
    s_[2] = 0.0
    np.testing.assert_allclose(f2, u @ np.diag(s_) @ vh, atol=1e-14)
    assert abs(np.linalg.det(f2)) <= 1e-16
    degenerate = golden("g7_degenerate")
    na, _ = eight_point._normalize_coords(degenerate["pix_a"])
    nb, _ = eight_point._normalize_coords(degenerate["pix_b"])
    with pytest.raises(eight_point.EightPointCalculationError):
        eight_point._compute_f_est(eight_point._get_yT_y(na, nb))
    ca, T1, cb, T2 = eight_point._get_normalized_match_coordinates(
        feats(d["ca"]), feats(d["cb"]), eight_point.create_trivial_matches(8))
    np.testing.assert_allclose(ca, d["na"], rtol=2e-16, atol=1e-16)
    np.testing.assert_allclose(T2, d["Tb"], rtol=2e-16, atol=1e-16)


def test_enforce_rank2_on_hard_matrices():
    """_enforce_fundamental_mat_constraints (eight_point.py:430-446) against LAPACK's truncated SVD on matrices the
    eight-point fit rarely sees: graded singular values down to a nearly rank-ONE matrix (where the kernel's
    projection from the two large Jacobi columns would lose sigma_1 / sigma_2 in accuracy and the wave takes the
    V-accumulating route instead), exact rank one, exact rank two, zero, and random well-conditioned ones."""
    rng = np.random.default_rng(12)

    def with_singular_values(s):
        u, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        v, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        return u @ np.diag(s) @ v.T

    cases = [rng.normal(size=(3, 3)) for _ in range(6)]
    cases += [with_singular_values(s) for s in ([1.0, 0.5, 1e-3], [1.0, 1e-3, 1e-9], [1.0, 1e-5, 1e-6], [1.0, 1e-7, 1e-12],
                                                [1.0, 1e-9, 1e-10], [1.0, 1e-13, 1e-16], [2.0, 1.0, 0.0], [3.0, 0.0, 0.0])]
    cases += [np.outer([1.0, -2.0, 0.5], [0.25, 4.0, -1.0]), np.zeros((3, 3))]
    # any magnitude: the SVD is scale-equivariant, the kernel scales by an exact power of two and back
    cases += [cases[0] * 1e-100, cases[1] * 1e-140, cases[2] * 1e100, cases[3] * 1e140, cases[7] * 1e-200]
    for f in cases:
        got = eight_point._enforce_fundamental_mat_constraints(f)
        u, s_, vh = np.linalg.svd(f)
        s_[2] = 0.0
        want = u @ np.diag(s_) @ vh
        scale = max(np.abs(f).max(), 1e-300)
        assert np.abs(got - want).max() <= 1e-13 * scale, (f, got, want)
        assert np.linalg.matrix_rank(got, tol=1e-12 * scale) <= 2


def test_decompose_essential_at_any_magnitude():
    """_recover_all_r_t (eight_point.py:245-280) is scale-free below the absolute sigma_3 ~ 0 test (:268-271): an E
    scaled down by 1e-60 ... 1e-140 gives the same rotations and translation direction; scaled up by 1e60 its third
    singular value (1e-16 of the first) exceeds the reference's atol of 1e-8 and the reference raises — so do we."""
    R = orc.euler_xy(-5.0, -10.0)
    t = np.array([0.5, 0.05, 0.1])
    t /= np.linalg.norm(t)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    for scale in (1.0, 1e-60, 1e-100, 1e-140, 1e-250):
        r1, r2, tt = eight_point._recover_all_r_t(E * scale)
        assert min(np.abs(r1 - R).max(), np.abs(r2 - R).max()) <= 1e-12, scale
        assert min(np.abs(tt - t).max(), np.abs(tt + t).max()) <= 1e-12, scale
    with pytest.raises(eight_point.EightPointCalculationError):
        eight_point._recover_all_r_t(E * 1e60)


def test_traced_fit_intermediates_bit_exact(golden):
    """Inside the fused fit kernel: Hartley normalisation and Y^T Y equal the real reference's values bit for
    bit (golden g8: _normalize_coords / _get_yT_y on the same 8 pairs); eigenvalues and E to LAPACK accuracy."""
    from structure_from_motion_amd import device

    d = golden("g8_units")
    corr = device.to_device(np.hstack([d["ca"], d["cb"]])).reshape(1, 8, 4)
    S = torch.arange(8, dtype=torch.int32, device=corr.device).reshape(1, 1, 8)
    E, flags, tr = device.fit_eight_point_traced(corr, S)
    np.testing.assert_array_equal(tr["norm_a"][0, 0], d["na"])
    np.testing.assert_array_equal(tr["norm_b"][0, 0], d["nb"])
    scale, cx, cy = tr["T1"][0, 0]
    np.testing.assert_array_equal(np.array([[scale, 0, -scale * cx], [0, scale, -scale * cy], [0, 0, 1.0]]), d["Ta"])
    np.testing.assert_array_equal(tr["yty"][0, 0], d["yty"])
    np.testing.assert_allclose(np.sort(tr["eigenvalues"][0, 0]), np.sort(np.real(np.linalg.eig(d["yty"])[0])),
                               atol=1e-13)
    E_plain, _ = device.fit_eight_point(corr, S)
    assert torch.equal(E, E_plain)                                      # tracing does not change the result
    E_o = orc.estimate_fundamental_mat(d["ca"], d["cb"])
    assert rel(E.cpu().numpy().reshape(3, 3), E_o) <= 1e-10


def test_headless_demo_end_to_end():
    """The whole reference pipeline (Harris -> NCC matching -> RANSAC E -> pose -> triangulation) on a rendered
    pair, checked against the scene's ground truth (config "make demo", headless)."""
    from apps import sfm_headless

    summary = sfm_headless.run_sfm({"num_harris_corners": 400,
                                    "ransac": {**sfm_headless.DEFAULT_CONFIG["ransac"], "max_iterations": 500}})
    assert summary["corners"] == [400, 400]
    assert summary["matches"] >= 60
    assert summary["ransac_inliers"] >= 18
    assert summary["points"] == summary["cheirality_inliers"] >= 15
    assert summary["rotation_error_deg"] < 2.0
    assert summary["translation_direction_error_deg"] < 12.0
    assert summary["fraction_of_points_in_true_depth_range"] > 0.8


def test_local_optimisation_env_option(monkeypatch):
    """SFM_LOCAL_OPTIMIZATION=k (extension): default 0 keeps the reference's result; k > 0 returns the refined
    model with at least as many inliers, identical to the oracle's local optimisation of the same winner."""
    n, thr = 1200, 1.5e-6
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(n, seed=4)
    fa = [Feature(x=float(p[0]), y=float(p[1])) for p in pa]
    fb = [Feature(x=float(p[0]), y=float(p[1])) for p in pb]
    matches = [Match(i, i, 0.0) for i in range(n)]
    kw = dict(sed_inlier_threshold=thr, min_num_extra_inliers=150,
              error_aggregation_method=ErrorAggregationMethod.RMS, max_iterations=300)
    random.seed(5)
    E0, pairs0 = epipolar_ransac.estimate_essential_mat_with_ransac(K, fa, fb, matches, **kw)
    monkeypatch.setenv("SFM_LOCAL_OPTIMIZATION", "5")
    random.seed(5)
    E1, pairs1 = epipolar_ransac.estimate_essential_mat_with_ransac(K, fa, fb, matches, **kw)
    assert len(pairs1) > len(pairs0)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    lookup = {(f.x, f.y): i for i, f in enumerate(fa)}
    mask0 = np.zeros(n, dtype=bool)
    mask0[[lookup[(a.x, a.y)] for a, _ in pairs0]] = True
    sed0 = orc.sed_values(E0, corr)[mask0]
    err0 = float(np.sqrt(np.mean(sed0 * sed0)))
    E_o, m_o, cnt_o, _, acc_o = orc.local_optimisation(corr, E0, mask0, err0, thr, orc.RMS, 5)
    assert acc_o >= 1 and len(pairs1) == cnt_o
    assert [lookup[(a.x, a.y)] for a, _ in pairs1] == list(np.nonzero(m_o)[0])
    assert np.max(np.abs(E1 - E_o)) / np.max(np.abs(E_o)) <= 1e-9


def test_rccl_accepts_the_record_all_gather():
    """The one collective of a sharded pass — all_gather_into_tensor of int64 [batch, 5] records — through the "nccl"
    backend (RCCL) itself, on a single-rank group (one GPU here; the multi-rank logic is covered by the gloo tests):
    dtype, shapes and the flat views are what RCCL is given at N = 8."""
    import socket

    import torch.distributed as dist

    from structure_from_motion_amd import distributed

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        record = torch.arange(10, dtype=torch.int64, device="cuda").reshape(2, 5)
        out = torch.empty((1, 2, 5), dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(out.view(-1), record.reshape(-1))
        torch.cuda.synchronize()
        assert torch.equal(out[0], record)
        glob, best_h, single = distributed.fold_records(out)
        assert glob.shape == (2, 5) and best_h.shape == (2,)
    finally:
        dist.destroy_process_group()


def test_bench_two_rank_rehearsal(tmp_path):
    """`python bench.py --gpus 2` end to end, typed without a launcher (bench.py starts one child process per rank through
    torch.distributed.run, exactly the command the driver would type), both
    ranks sharing the one GPU of the box with the collective staged through gloo: the N > 1 path — shard_range
    partition, 40-byte all-gather, fold, winner re-derivation, max-over-ranks timing — produces one JSON line whose
    winner equals a single-rank run over the same global hypothesis range."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--matches", "3000", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, SFM_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # typed the plain way, with no launcher: bench.py starts its two ranks itself (child processes through
    # torch.distributed.run) and relays rank 0's line
    two = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--hypotheses", "4000"] + common,
                         env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert two.returncode == 0, two.stderr[-2000:]
    assert len(two.stdout.strip().splitlines()) == 1, two.stdout[:600]   # ONE line on stdout (gloo's own chatter goes to stderr)
    line2 = json.loads(two.stdout.strip().splitlines()[-1])
    one = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--hypotheses", "8000"] + common,
                         capture_output=True, text=True, timeout=600, cwd=repo)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert line2["n_gpus"] == 2 and line2["config"]["global_hypotheses"] == 8000
    assert line2["config"]["exchange"].startswith("one all_gather")
    assert line2["result"] == line1["result"]        # same winner, error, inlier count from either partition
    assert line2["value"] > 0 and line2["roofline"]["bound"] == "valu-issue"
    # 3000 x 4000 per rank runs as the lean small pass: its scoring launch is bracketed by the events all the same,
    # and there is no separate scoring call to time
    assert line2["roofline"]["kernel_ms"] > 0 and line2["roofline"]["score_call_ms"] is None


def test_rccl_exchange_on_one_rank(tmp_path):
    """The multi-GPU path under the REAL backend, as far as one GPU allows: a child process — started
    before anything in it touches the GPU — with WORLD_SIZE=1 initialises the "nccl" (= RCCL) process group exactly as
    bench.py does (`device_id=cuda:0`, HSA_ENABLE_IPC_MODE_LEGACY=0) and runs `ShardedRansac(..., force_exchange=True)`:
    the selection kernel writes global indices, `dist.all_gather_into_tensor` runs as an RCCL collective on the int64 DEVICE
    tensor, the records are folded and the winner re-derived from (seed, h*).  Winner, error, E, sample and mask equal the
    single-GPU engine's bit for bit, over three seeds and with a hypothesis offset.  It cannot measure scaling; it makes sure
    the first 8-GPU run is not the first time RCCL sees this code."""
    import os
    import socket
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "from structure_from_motion_amd import device, distributed, synthetic\n"
        "from structure_from_motion_amd._native import AGG_RMS\n"
        "n, h = 3000, 6000\n"
        "pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)\n"
        "corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)\n"
        "plain = distributed.ShardedRansac(corr, None, 1.5e-6, 10, AGG_RMS, 0, 1, total_hypotheses=h)\n"
        "forced = distributed.ShardedRansac(corr, None, 1.5e-6, 10, AGG_RMS, 0, 1, total_hypotheses=h, force_exchange=True)\n"
        "assert forced.gathered.is_cuda and forced.gathered.dtype == torch.int64 and forced.gathered.shape == (1, 1, 5)\n"
        "for seed in (5, 6, 2**63 + 11):\n"
        "    plain.step(seed); a = plain.outcome()\n"
        "    forced.gathered.fill_(-7)\n"
        "    forced.step(seed); b = forced.outcome()\n"
        "    assert torch.equal(forced.gathered[0], forced.ws.result), 'the collective did not deliver the record'\n"
        "    assert a.best_h == b.best_h >= 0 and a.error == b.error, (a.best_h, b.best_h)\n"
        "    assert np.array_equal(a.E, b.E) and np.array_equal(a.sample, b.sample) and np.array_equal(a.mask, b.mask)\n"
        "# a rank that does not own hypothesis 0: global indices through the same exchange\n"
        "late = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS, rank=3, world=1, force_exchange=True)\n"
        "late.step(5); c = late.outcome()\n"
        "assert 3 * h <= c.best_h < 4 * h, c.best_h\n"
        "dist.barrier()\n"
        "dist.destroy_process_group()\n"
        "print('rccl one-rank exchange ok', a.best_h, c.best_h)\n"
    ) % repo
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=repo)
    assert out.returncode == 0 and "rccl one-rank exchange ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
