"""CPU tests: the oracle (oracle/sfm_oracle.py, oracle/sed_score.c) against golden vectors produced by
the REAL reference (tests/golden/make_golden.py).  These pin the oracle before it is trusted as the
checker of the HIP path."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import sfm_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corr_of(d):
    return orc.pack_correspondences(orc.to_normalized_image_coords(d["pix_a"], d["K"]),
                                    orc.to_normalized_image_coords(d["pix_b"], d["K"]))


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def test_g8_unit_vectors(golden):
    d = golden("g8_units")
    nc, T = orc.hartley_normalize(d["norm_in"])
    np.testing.assert_array_equal(nc, d["norm_out"])
    np.testing.assert_array_equal(T, d["norm_T"])
    # reference test_normalize_coords: T is the forward transform
    back = np.hstack([nc, np.ones((3, 1))]) @ np.linalg.inv(T).T
    np.testing.assert_allclose(back[:, :2], d["norm_in"])
    ycol = orc.y_columns(np.array([[2.0, 3.0]]), np.array([[7.0, 6.0]]))[0]
    np.testing.assert_array_equal(ycol, d["ycol"])
    nf = orc.to_normalized_image_coords(np.array([50.0, 60.0]), d["K"])
    np.testing.assert_array_equal(nf, d["nf"])
    na, Ta = orc.hartley_normalize(d["ca"])
    nb, Tb = orc.hartley_normalize(d["cb"])
    np.testing.assert_array_equal(na, d["na"])
    np.testing.assert_array_equal(Tb, d["Tb"])
    np.testing.assert_array_equal(orc.yty(na, nb), d["yty"])


def test_g1_eight_point_pipeline(golden):
    d = golden("g1_eight_point")
    F = orc.estimate_fundamental_mat(d["pix_a"], d["pix_b"])
    assert rel(F, d["F"]) <= 1e-12
    na = orc.to_normalized_image_coords(d["pix_a"], d["K"])
    nb = orc.to_normalized_image_coords(d["pix_b"], d["K"])
    E = orc.estimate_fundamental_mat(na, nb)
    assert rel(E, d["E"]) <= 1e-12
    assert rel(E, d["E_gt"]) <= 1e-8  # analytic [t]x R / e22 (the reference compares to OpenCV at 1e-5)
    R1, R2, t1 = orc.recover_all_r_t(d["E"])
    np.testing.assert_allclose(R1, d["R1"], atol=1e-12)
    np.testing.assert_allclose(R2, d["R2"], atol=1e-12)
    np.testing.assert_allclose(t1, d["t1"], atol=1e-12)
    R, t, mask, votes = orc.recover_r_t(orc.pack_correspondences(na, nb), d["E"])
    np.testing.assert_allclose(R, d["R"], atol=1e-12)
    np.testing.assert_allclose(t, d["t"], atol=1e-12)
    np.testing.assert_array_equal(mask, d["mask"])
    assert votes[int(np.argmax(votes))] == 7  # quirk Q9: index 0 passes but is not counted
    np.testing.assert_allclose(R, d["R_gt"], atol=1e-8)
    np.testing.assert_allclose(t, d["t_gt"], atol=1e-8)
    sed = orc.sed_values(d["E_gt"], orc.pack_correspondences(na, nb))
    assert np.all(sed < 1e-20)


def test_g2_ransac_seed5(golden):
    d = golden("g2_ransac_seed5")
    random.seed(5)
    S, perms = orc.pyshuffle_sample_table(len(d["pix_a"]), 100)
    np.testing.assert_array_equal(S, d["S"])
    corr = corr_of(d)
    out = orc.ransac_essential(corr, S, 0.01, 0, orc.SUM)
    assert rel(out["Eall"], d["Eall"]) <= 1e-12
    assert rel(out["E"], d["E"]) <= 1e-12
    ordered = orc.inlier_indices(corr, out["E"], S[out["best"]], 0.01, rest_order=perms[out["best"]][8:])
    np.testing.assert_array_equal(ordered, d["inlier_idx"])
    np.testing.assert_array_equal(d["pix_a"][ordered], d["inlier_a"])
    assert rel(out["E"], d["E_gt"]) <= 1e-5


@pytest.mark.parametrize("method", [orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS])
def test_g3_per_hypothesis(golden, method):
    d = golden("g3_per_hypothesis")
    corr = corr_of(d)
    thr, min_extra = float(d["thr"]), int(d["min_extra"])
    out = orc.ransac_essential(corr, d["S"], thr, min_extra, method)
    assert rel(out["Eall"], d["Eall"]) <= 1e-12
    np.testing.assert_array_equal(out["cnt"], d["cnt"])
    ref_err = d["err_" + method]
    assert np.max(np.abs(out["errs"] - ref_err) / ref_err) <= 1e-11
    assert rel(out["E"], d["E_" + method]) <= 1e-12
    assert set(out["inliers"].tolist()) == set(d["inliers_" + method].tolist())
    np.testing.assert_array_equal(out["inliers"][:8], d["inliers_" + method][:8])
    # value-level SED agreement and the decision margin of this fixture
    sed = orc.sed_values(d["Eall"], corr)
    ok = ~np.isnan(d["sed_all"])
    assert np.max(np.abs(sed[ok] - d["sed_all"][ok]) / d["sed_all"][ok]) <= 1e-8
    assert np.min(np.abs(d["sed_all"][ok] - thr) / thr) > 1e-6


def test_g3_literal_aggregation(golden):
    """ransac.py:96-108 verbatim on the compact list, in the reference's shuffled order."""
    d = golden("g3_per_hypothesis")
    thr = float(d["thr"])
    random.seed(5)
    S, perms = orc.pyshuffle_sample_table(len(d["pix_a"]), len(d["S"]))
    np.testing.assert_array_equal(S, d["S"])
    for method in (orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS):
        for h in (0, 7, 31, 49):
            sed_h = d["sed_all"][h]
            errs = [sed_h[i] for i in perms[h][:8]] + [sed_h[i] for i in perms[h][8:] if sed_h[i] <= thr]
            assert orc.aggregate_literal(errs, method) == d["err_" + method][h]


def test_g4_sed_values(golden):
    d = golden("g4_sed")
    v = orc.sed_values(d["E"], orc.pack_correspondences(d["norm_a"], d["norm_b"]))
    assert np.max(np.abs(v - d["sed"]) / d["sed"]) <= 1e-13


def test_g5_cheirality(golden):
    d = golden("g5_cheirality")
    corr = corr_of(d)
    R1, R2, t1 = orc.recover_all_r_t(d["E"])
    import itertools
    for c, (R, t) in enumerate(itertools.product([R1, R2], [t1, -t1])):
        np.testing.assert_array_equal(orc.cheirality_pass(corr, R, t).astype(np.uint8), d["passes"][c])
    R, t, mask, votes = orc.recover_r_t(corr, d["E"])
    np.testing.assert_array_equal(votes, d["votes"])
    np.testing.assert_array_equal(mask, d["mask"])
    np.testing.assert_allclose(R, d["R"], atol=1e-12)
    np.testing.assert_allclose(t, d["t"], atol=1e-12)
    Rd, td, maskd, _ = orc.recover_r_t(corr, d["E"], 5.2)
    np.testing.assert_array_equal(maskd, d["mask_d"])
    np.testing.assert_allclose(R, d["R_gt"], atol=1e-9)


def test_g6_triangulate(golden):
    d = golden("g6_triangulate")
    X = orc.triangulate_points(d["pix_a"], d["pix_b"], d["K"], d["cam2_T_cam1"])
    assert np.max(np.abs(X - d["X"]) / np.abs(d["X"])) <= 1e-9
    Xk = orc.triangulate_dlt(orc.pack_correspondences(d["known_a"][None], d["known_b"][None]),
                             d["known_P1"], d["known_P2"])[0]
    np.testing.assert_allclose(Xk, [0.0, 0.0, 10.0], atol=1e-10)
    np.testing.assert_allclose(Xk, d["known_X"], atol=1e-12)
    with pytest.raises(ValueError):
        orc.triangulate_points(d["pix_a"], d["pix_b"], np.eye(4), d["cam2_T_cam1"])


def test_g7_degenerate(golden):
    d = golden("g7_degenerate")
    assert bool(d["raised"])
    with pytest.raises(orc.OracleDegenerateSample):
        orc.estimate_fundamental_mat(d["pix_a"], d["pix_b"])


@pytest.mark.parametrize("method", [orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS])
def test_g9_explicit_table(golden, method):
    d = golden("g9_explicit_table")
    corr = corr_of(d)
    S = d["S"]
    np.testing.assert_array_equal(orc.philox_sample_table(5, 0, len(S), len(corr)), S)
    out = orc.ransac_essential(corr, S, float(d["thr"]), int(d["min_extra"]), method)
    assert out["best"] == int(d["best_" + method])
    assert rel(out["E"], d["E_" + method]) <= 1e-12
    np.testing.assert_array_equal(out["inliers"], d["inliers_" + method])


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    def run(ctr, key):
        return [int(x) for x in orc.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]]

    assert run([0, 0, 0, 0], (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run([0xFFFFFFFF] * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], (0xA4093822, 0x299F31D0)) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_sampler_properties():
    S = orc.philox_sample_table(123, 1000, 5000, 37)
    assert S.shape == (5000, 8) and S.min() >= 0 and S.max() < 37
    assert all(len(set(row)) == 8 for row in S.tolist())
    # counter-based: a sub-range equals the same rows of a bigger table
    np.testing.assert_array_equal(orc.philox_sample_table(123, 1200, 50, 37), S[200:250])
    S8 = orc.philox_sample_table(1, 0, 200, 8)  # n == 8: a permutation
    assert all(sorted(r) == list(range(8)) for r in S8.tolist())
    # roughly uniform first index
    counts = np.bincount(S[:, 0], minlength=37)
    assert counts.min() > 60 and counts.max() < 220


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle")], check=True)
    lib = C.CDLL(os.path.join(REPO, "oracle", "libsfm_oracle.so"))
    lib.sfm_oracle_score.restype = C.c_int
    lib.sfm_oracle_score.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.sfm_oracle_sed_values.restype = None
    lib.sfm_oracle_sed_values.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    return lib


def test_c_oracle_matches_numpy_oracle(c_oracle):
    """The plain-C scoring loop (cpu_baseline leg) is bit-identical to the numpy oracle."""
    n, h, thr = 1000, 96, 1.5e-6
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=3)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(9, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    cnt, s1, s2 = orc.score_hypotheses(corr, E, S, thr)
    E_c = np.ascontiguousarray(E.reshape(h, 9))
    c_cnt = np.zeros(h, dtype=np.int32)
    c_s1 = np.zeros(h)
    c_s2 = np.zeros(h)
    used = c_oracle.sfm_oracle_score(corr.ctypes.data, n, E_c.ctypes.data, S.ctypes.data, h, thr,
                                     c_cnt.ctypes.data, c_s1.ctypes.data, c_s2.ctypes.data, 2)
    assert used >= 1
    np.testing.assert_array_equal(c_cnt, cnt)
    np.testing.assert_array_equal(c_s1, s1)
    np.testing.assert_array_equal(c_s2, s2)
    vals = np.zeros(n)
    c_oracle.sfm_oracle_sed_values(corr.ctypes.data, n, E_c[3].ctypes.data, vals.ctypes.data)
    np.testing.assert_array_equal(vals, orc.sed_values(E[3], corr))


def test_oracle_fit_against_multiprecision():
    """The LAPACK-based oracle fit agrees with a 40-digit evaluation of the same algorithm."""
    from oracle.fit_mp import fit_eight_point_mp

    n, h = 500, 24
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=6)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(7, 0, h, n)
    E, deg, lam2 = orc.fit_hypotheses(corr, S)
    for i in range(h):
        truth, lam_true = fit_eight_point_mp(corr[S[i]][:, 0:2], corr[S[i]][:, 2:4])
        # null-vector conditioning ~ eps * lambda_max / lambda_2 (lambda_max ~ 10 after Hartley scaling)
        assert np.max(np.abs(E[i] - truth)) / np.max(np.abs(truth)) <= 1e-13 * max(100.0, 100.0 / lam_true)
        assert abs(lam2[i] - lam_true) <= 1e-13


# ------------------------------------------------------------------------------------------------------
# local optimisation (extension): the N-point fit against the reference's helpers (G13)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_refit_matches_reference_helpers(golden, case):
    d = golden("g13_refit")
    K = d[f"{case}_K"]
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(d[f"{case}_pix_a"], K),
                                    orc.to_normalized_image_coords(d[f"{case}_pix_b"], K))
    E, degenerate = orc.refit_on_points(corr, d[f"{case}_idx"])
    assert not degenerate
    want = d[f"{case}_E"]
    assert np.max(np.abs(E - want)) / np.max(np.abs(want)) <= 1e-9


def test_local_optimisation_semantics():
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(2000, seed=6)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(5, 0, 500, 2000)
    thr = 1.5e-6
    ref = orc.ransac_essential(corr, S, thr, 200, orc.RMS)
    mask = np.zeros(2000, dtype=bool)
    mask[ref["inliers"]] = True
    counts = []
    for rounds in (0, 1, 2, 3, 10):
        E, m, cnt, err, accepted = orc.local_optimisation(corr, ref["E"], mask, ref["err"], thr, orc.RMS, rounds)
        counts.append(cnt)
        assert accepted <= rounds and cnt == m.sum()
        if rounds == 0:
            np.testing.assert_array_equal(E, ref["E"])
            np.testing.assert_array_equal(m, mask)
        else:  # the mask is exactly the threshold set of the returned model
            np.testing.assert_array_equal(m, orc.sed_values(E, corr) <= thr)
    assert counts == sorted(counts) and counts[-1] > counts[0]  # never loses inliers; here it gains
    # far fewer outliers among the inliers than in the data
    assert (m & is_out).sum() <= 0.01 * m.sum()
    # fewer than eight inliers, or a degenerate inlier set (all points identical): nothing happens
    few = np.zeros(2000, dtype=bool)
    few[:7] = True
    assert orc.local_optimisation(corr, ref["E"], few, 1.0, thr, orc.RMS, 3)[4] == 0
    flat = corr.copy()
    flat[:] = corr[0]
    assert orc.local_optimisation(flat, ref["E"], mask, 1.0, thr, orc.RMS, 3)[4] == 0
