"""GPU parity tests (run with ``-m gpu`` on an MI355X): HIP kernels through the C ABI against the CPU
oracle on the same seeded inputs, and against the golden vectors of the real reference.

Bars: bit-exact for integer / index work (sample tables, inlier counts, inlier index sets, masks,
selected hypothesis) and for element-wise fp64 values whose operation order is fixed (K-normalisation,
SED); relative tolerances written next to each assertion for the iterative eigen/SVD routines and the
order-dependent sums."""
import os

import numpy as np
import pytest

from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev(native_lib):
    from structure_from_motion_amd import device

    device.require_gpu()
    return device


def scene(n, seed=6, outliers=0.3):
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(n, seed=seed, outlier_fraction=outliers)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    return pa, pb, K, corr


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


# ------------------------------------------------------------------------------------------------------
# kernel-level parity
# ------------------------------------------------------------------------------------------------------
def test_normalize_bit_exact(dev):
    pa, pb, K, corr = scene(1000)
    got = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).cpu().numpy()
    np.testing.assert_array_equal(got, corr)


@pytest.mark.parametrize("n,h,h_begin", [(8, 100, 0), (37, 1000, 12345), (50000, 4096, 2**33 + 5)])
def test_philox_sampler_bit_exact(dev, n, h, h_begin):
    got = dev.sample_philox(5, h_begin, h, n).cpu().numpy()[0]
    np.testing.assert_array_equal(got, orc.philox_sample_table(5, h_begin, h, n))


def test_philox_batch_seeds(dev):
    got = dev.sample_philox(6, 0, 64, 500, batch=3, seed_stride=1).cpu().numpy()
    for b in range(3):
        np.testing.assert_array_equal(got[b], orc.philox_sample_table(6 + b, 0, 64, 500))


def test_fused_sample_and_fit_equals_separate_calls(dev):
    """sfm_sample_fit_philox == sfm_sample_philox (or _dev) followed by sfm_fit_eight_point: same S, E bit for bit."""
    B, n, h = 3, 700, 333
    corr = torch.stack([dev.to_device(scene(n, seed=20 + b)[3]) for b in range(B)])
    for seed_arg, seed, h_begin, stride in ((41, 41, 0, 1), (2**63 + 5, 2**63 + 5, 1000, 7), ("dev", 9, 12, 2)):
        S_ref = dev.sample_philox(seed, h_begin, h, n, batch=B, seed_stride=stride)
        E_ref, f_ref = dev.fit_eight_point(corr, S_ref)
        S = torch.zeros_like(S_ref)
        E = torch.zeros_like(E_ref)
        flags = torch.full_like(f_ref, -1)
        arg = torch.tensor([seed], dtype=torch.int64, device=corr.device) if seed_arg == "dev" else seed_arg
        dev.sample_fit_philox(corr, arg, h_begin, S, E, flags, seed_stride=stride)
        np.testing.assert_array_equal(S.cpu().numpy(), S_ref.cpu().numpy())
        np.testing.assert_array_equal(E.cpu().numpy(), E_ref.cpu().numpy())
        np.testing.assert_array_equal(flags.cpu().numpy(), f_ref.cpu().numpy())


def test_sed_values_bit_exact(dev, golden):
    rng = np.random.default_rng(0)
    _, _, _, corr = scene(3000)
    for _ in range(4):
        E = rng.normal(size=(3, 3))
        E[2, 2] = 1.0
        got = dev.sed_values(dev.to_device(corr), dev.to_device(E.reshape(9))).cpu().numpy()
        np.testing.assert_array_equal(got, orc.sed_values(E, corr))
    d = golden("g4_sed")  # the real reference's values (BLAS order): 1e-13 relative
    c = orc.pack_correspondences(d["norm_a"], d["norm_b"])
    for k in range(len(d["E"])):
        got = dev.sed_values(dev.to_device(c), dev.to_device(d["E"][k].reshape(9))).cpu().numpy()
        assert np.max(np.abs(got - d["sed"][k]) / d["sed"][k]) <= 1e-13


def test_sed_scale_invariance_exact(dev):
    """SED is homogeneous of degree 0 in E; scaling E by a power of two is exact in fp64."""
    _, _, _, corr = scene(500)
    E = np.random.default_rng(1).normal(size=9)
    a = dev.sed_values(dev.to_device(corr), dev.to_device(E)).cpu().numpy()
    b = dev.sed_values(dev.to_device(corr), dev.to_device(E * 4.0)).cpu().numpy()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,h", [(200, 50), (1000, 257), (64, 64)])
def test_fit_matches_oracle(dev, n, h):
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(7, 0, h, n)
    E_ref, deg_ref, lam_ref = orc.fit_hypotheses(corr, S)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    S_d = dev.to_device(S, torch.int32).reshape(1, h, 8)
    lam = torch.empty((1, h), dtype=torch.float64, device=corr_d.device)
    E, flags = dev.fit_eight_point(corr_d, S_d, lambda2=lam)
    E = E.cpu().numpy().reshape(h, 3, 3)
    np.testing.assert_array_equal(flags.cpu().numpy()[0] != 0, deg_ref)
    np.testing.assert_array_equal(E[:, 2, 2], 1.0)
    # Jacobi (device) vs LAPACK dgeev/dgesdd (oracle): two double-precision routes to an ill-conditioned
    # null vector.  Hard bar (north_star): 1e-6 relative on every hypothesis; typical agreement 1e-13.
    err = np.max(np.abs(E - E_ref), axis=(1, 2)) / np.max(np.abs(E_ref), axis=(1, 2))
    assert err.max() <= 1e-6 and np.median(err) <= 1e-12, (err.max(), np.median(err))
    # Which of the two is closer to the truth?  Tie-break with a 40-digit evaluation of the same
    # algorithm (oracle/fit_mp.py) on a subset: the device must be as accurate as LAPACK.
    from oracle.fit_mp import fit_eight_point_mp

    pick = np.unique(np.concatenate([np.argsort(err)[-12:], np.arange(0, h, max(1, h // 40))]))
    dev_err, lap_err = [], []
    for i in pick:
        pts = corr[S[i]]
        truth, _ = fit_eight_point_mp(pts[:, 0:2], pts[:, 2:4])
        scale = np.max(np.abs(truth))
        dev_err.append(np.max(np.abs(E[i] - truth)) / scale)
        lap_err.append(np.max(np.abs(E_ref[i] - truth)) / scale)
    dev_err, lap_err = np.array(dev_err), np.array(lap_err)
    assert np.median(dev_err) <= 3.0 * np.median(lap_err) + 1e-14, (np.median(dev_err), np.median(lap_err))
    assert dev_err.max() <= 10.0 * lap_err.max() + 1e-13, (dev_err.max(), lap_err.max())
    lam_got = lam.cpu().numpy()[0]
    assert np.max(np.abs(lam_got - lam_ref)) <= 1e-12 * 20.0


@pytest.mark.parametrize("noise_px", [0.5, 0.0])
def test_fit_every_hypothesis_of_20000(dev, noise_px):
    """20 000 Philox hypotheses on 5 000 correspondences, with and without pixel noise: EVERY fit is within 1e-6 of the
    oracle (the reference's route: LAPACK eig of YtY, svd) — or, for the handful of near-degenerate samples on which the two
    double-precision routes disagree by more (up to 7e-5 noise-free: the eigen-solve of the squared matrix loses what the
    QR of Y keeps), the device is within 1e-9 of a 40-digit evaluation of the same algorithm and closer to it than the
    oracle.  Degeneracy flags equal on all of them."""
    n, h = 5000, 20000
    pa, pb, K, *_ = orc.synthetic_two_view(n, seed=6, noise_px=noise_px)
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(11, 0, h, n)
    E_ref, deg_ref, _ = orc.fit_hypotheses(corr, S)
    E, flags = dev.fit_eight_point(dev.to_device(corr).reshape(1, n, 4), dev.to_device(S, torch.int32).reshape(1, h, 8))
    np.testing.assert_array_equal(flags.cpu().numpy()[0] != 0, deg_ref)
    err = assert_fits_agree(corr, S, E.cpu().numpy()[0], E_ref, ok=~deg_ref, median=1e-12, mp_budget=40)
    assert np.quantile(err, 0.99) <= 1e-9


def test_fit_golden_reference(dev, golden):
    d = golden("g3_per_hypothesis")
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(d["pix_a"], d["K"]),
                                    orc.to_normalized_image_coords(d["pix_b"], d["K"]))
    E, flags = dev.fit_eight_point(dev.to_device(corr).reshape(1, -1, 4),
                                   dev.to_device(d["S"], torch.int32).reshape(1, -1, 8))
    E = E.cpu().numpy().reshape(-1, 3, 3)
    err = np.max(np.abs(E - d["Eall"]), axis=(1, 2)) / np.max(np.abs(d["Eall"]), axis=(1, 2))
    assert err.max() <= 1e-9, err.max()
    assert not flags.cpu().numpy().any()


@pytest.mark.parametrize("n,h", [(200, 50), (1000, 130), (64, 7), (4099, 33)])
def test_score_matches_oracle_with_identical_E(dev, n, h):
    """Same E on both sides -> counts bit-exact; sums differ only by summation order."""
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(11, 0, h, n)
    E_ref, _, _ = orc.fit_hypotheses(corr, S)
    thr = 1.5e-6
    cnt_ref, s1_ref, s2_ref = orc.score_hypotheses(corr, E_ref, S, thr)
    cnt, s1, s2 = dev.score_sed(dev.to_device(corr).reshape(1, n, 4), dev.to_device(E_ref.reshape(1, h, 9)),
                                dev.to_device(S, torch.int32).reshape(1, h, 8), thr)
    np.testing.assert_array_equal(cnt.cpu().numpy()[0], cnt_ref)
    np.testing.assert_allclose(s1.cpu().numpy()[0], s1_ref, rtol=1e-13, atol=0)
    np.testing.assert_allclose(s2.cpu().numpy()[0], s2_ref, rtol=1e-13, atol=0)


def test_score_nan_and_inf_models_never_win(dev):
    n, h = 300, 8
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(3, 0, h, n)
    E_ref, _, _ = orc.fit_hypotheses(corr, S)
    E_bad = E_ref.copy()
    E_bad[2] = np.nan
    E_bad[5] = np.inf
    E_bad[6] = 0.0  # 1/0 -> inf, inf * 0 -> NaN
    ws = dev.RansacWorkspace(1, n, h)
    ws.S.copy_(dev.to_device(S, torch.int32).reshape(1, h, 8))
    ws.E.copy_(dev.to_device(E_bad.reshape(1, h, 9)))
    ws.flags.zero_()
    dev.score_sed(dev.to_device(corr).reshape(1, n, 4), ws.E, ws.S, 1.5e-6, ws.cnt, ws.s1, ws.s2)
    for agg, name in enumerate([orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS]):
        res = dev.read_select(dev.select_best(ws.cnt, ws.s1, ws.s2, ws.flags, 0, agg))[0]
        cnt_o, s1_o, s2_o = orc.score_hypotheses(corr, E_bad, S, 1.5e-6)
        best_o, err_o = orc.select_best(orc.aggregate(cnt_o, s1_o, s2_o, name), cnt_o, 0)
        assert res.best_h == best_o and res.best_h not in (2, 5, 6)


@pytest.mark.parametrize("h", [5000, 70000])  # one-launch block kernel (<= 32768) / multi-block atomics path
@pytest.mark.parametrize("method", [orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS])
def test_select_matches_oracle(dev, method, h):
    rng = np.random.default_rng(5)
    cnt = rng.integers(0, 40, h).astype(np.int32)
    s1 = rng.random(h)
    s2 = rng.random(h)
    s1[100] = s1[4000] = 1e-9  # exact tie -> earliest wins
    s2[100] = s2[4000] = 1e-9
    cnt[100] = cnt[4000] = 39
    s1[7] = np.nan
    s2[9] = np.inf
    flags = np.zeros(h, dtype=np.int32)
    code = [orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS].index(method)
    for min_extra in (0, 10, 37.5, 1000):
        res = dev.read_select(dev.select_best(
            dev.to_device(cnt, torch.int32).reshape(1, h), dev.to_device(s1).reshape(1, h),
            dev.to_device(s2).reshape(1, h), dev.to_device(flags, torch.int32).reshape(1, h),
            min_extra, code, h_offset=1_000_000))[0]
        best, err = orc.select_best(orc.aggregate(cnt, s1, s2, method), cnt, min_extra)
        if best < 0:
            assert res.best_h == -1 and res.best_err == np.inf
        else:
            assert res.best_h == best + 1_000_000
            assert res.best_err == err
            assert res.best_cnt == cnt[best]
    flags[[17, 3000]] = 1
    res = dev.read_select(dev.select_best(
        dev.to_device(cnt, torch.int32).reshape(1, h), dev.to_device(s1).reshape(1, h),
        dev.to_device(s2).reshape(1, h), dev.to_device(flags, torch.int32).reshape(1, h), 0, code))[0]
    assert res.n_flagged == 2 and res.first_flagged == 17


@pytest.mark.parametrize("h", [1, 63, 1025, 33000])
def test_select_batched_matches_oracle(dev, h):
    """Several independent pairs in one launch (both select paths), including a pair with nothing gated, a pair
    whose minimum is shared by the first and the last hypothesis, and flags."""
    rng = np.random.default_rng(h)
    B = 4
    cnt = rng.integers(0, 30, (B, h)).astype(np.int32)
    s1, s2 = rng.random((B, h)), rng.random((B, h))
    cnt[1] = 0                      # pair 1: gate (>= 5) never met
    s2[2, 0] = s2[2, -1] = 1e-12    # pair 2: tie between first and last -> first
    cnt[2, 0] = cnt[2, -1] = 29
    flags = np.zeros((B, h), dtype=np.int32)
    flags[3, h // 2] = 1            # pair 3: one flagged hypothesis never competes
    s2[3, h // 2], cnt[3, h // 2] = 0.0, 29
    res = dev.read_select(dev.select_best(dev.to_device(cnt, torch.int32), dev.to_device(s1), dev.to_device(s2),
                                          dev.to_device(flags, torch.int32), 5, 3, h_offset=7))
    for b in range(B):
        err = orc.aggregate(cnt[b], s1[b], s2[b], orc.RMS)
        err[flags[b] != 0] = np.inf
        best, best_err = orc.select_best(err, cnt[b], 5)
        if best < 0:
            assert res[b].best_h == -1 and res[b].best_err == np.inf and res[b].best_cnt == 0
        else:
            assert (res[b].best_h, res[b].best_err, res[b].best_cnt) == (best + 7, best_err, cnt[b, best])
        assert res[b].n_flagged == int(flags[b].sum())
        if flags[b].any():
            assert res[b].first_flagged == int(np.nonzero(flags[b])[0][0]) + 7


# ------------------------------------------------------------------------------------------------------
# pipeline-level parity (sample -> fit -> score -> select -> mask)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,h,seed", [(300, 200, 5), (2000, 1000, 9), (5000, 10000, 5)])
def test_ransac_pipeline_matches_oracle(dev, n, h, seed):
    """C2-sized at the top end (5k x 10k): winner, inlier index set and E against the oracle."""
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(n)
    thr, min_extra = 1.5e-6, 10
    S = orc.philox_sample_table(seed, 0, h, n)
    ref = orc.ransac_essential(corr, S, thr, min_extra, orc.RMS)
    S_d = dev.sample_philox(seed, 0, h, n)
    got = dev.ransac_essential(dev.to_device(corr), S_d[0], thr, min_extra, AGG_RMS)
    assert got.best_h == ref["best"]
    assert got.n_flagged == int(ref["degenerate"].sum())
    np.testing.assert_array_equal(got.sample, S[ref["best"]])
    assert rel(got.E, ref["E"]) <= 1e-6
    np.testing.assert_array_equal(np.nonzero(got.mask == 1)[0], ref["inliers"][8:])  # bit-exact index set
    np.testing.assert_array_equal(np.sort(np.nonzero(got.mask == 2)[0]), np.sort(S[ref["best"]]))
    assert abs(got.error - ref["err"]) <= 1e-8 * ref["err"]  # error is a function of E (agrees to ~1e-10)
    assert got.extra_inliers == ref["cnt"][ref["best"]]
    # decision margin of the winner: how far the nearest SED is from the threshold
    sed = orc.sed_values(ref["E"], corr)
    assert np.min(np.abs(sed - thr) / thr) > 1e-9


def test_golden_explicit_table_all_methods(dev, golden):
    """The real reference driven by the same sample table (tests/golden g9)."""
    d = golden("g9_explicit_table")
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(d["pix_a"], d["K"]),
                                    orc.to_normalized_image_coords(d["pix_b"], d["K"]))
    for code, method in enumerate([orc.SUM, orc.SQUARE, orc.MEAN, orc.RMS]):
        got = dev.ransac_essential(dev.to_device(corr), d["S"], float(d["thr"]), int(d["min_extra"]), code)
        assert got.best_h == int(d["best_" + method])
        assert rel(got.E, d["E_" + method]) <= 1e-6
        order = list(got.sample) + list(np.nonzero(got.mask == 1)[0])
        np.testing.assert_array_equal(order, d["inliers_" + method])


def test_batched_pairs_equal_single_runs(dev):
    from structure_from_motion_amd._native import AGG_RMS

    B, n, h = 3, 700, 300
    corr_all = np.stack([scene(n, seed=20 + b)[3] for b in range(B)])
    ws = dev.RansacWorkspace(B, n, h)
    dev.sample_philox(40, 0, h, n, batch=B, seed_stride=1, out=ws.S)
    ws.run(dev.to_device(corr_all), 1.5e-6, 10, AGG_RMS)
    for b in range(B):
        single = dev.ransac_essential(dev.to_device(corr_all[b]), dev.sample_philox(40 + b, 0, h, n)[0],
                                      1.5e-6, 10, AGG_RMS)
        batched = ws.outcome(b)
        assert batched.best_h == single.best_h
        np.testing.assert_array_equal(batched.E, single.E)
        np.testing.assert_array_equal(batched.mask, single.mask)


def test_run_to_run_determinism(dev):
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(3000)
    S = dev.sample_philox(1, 0, 2000, 3000)
    outs = []
    for _ in range(2):
        ws = dev.RansacWorkspace(1, 3000, 2000)
        ws.S.copy_(S)
        ws.run(dev.to_device(corr).reshape(1, 3000, 4), 1.5e-6, 10, AGG_RMS)
        outs.append([t.cpu().numpy().tobytes() for t in (ws.E, ws.cnt, ws.s1, ws.s2, ws.result, ws.mask)])
    assert outs[0] == outs[1]


def assert_fits_agree(corr, S, E_dev, E_ref, ok=None, hard=1e-6, median=1e-11, mp_budget=24):
    """Every hypothesis: max|dE|/max|E| <= 1e-6 against the oracle (the reference's LAPACK route: eig of YtY, svd) —
    or, where the two double-precision routes disagree by more (near-degenerate samples the reference itself resolves
    to only ~1e-5: DESIGN.md section 4), the DEVICE is within 1e-9 of a 40-digit evaluation of the same algorithm and at
    least as close to it as the oracle.  Returns the relative differences."""
    E_dev = E_dev.reshape(-1, 3, 3)
    E_ref = E_ref.reshape(-1, 3, 3)
    finite = np.isfinite(E_ref).all(axis=(1, 2)) & np.isfinite(E_dev).all(axis=(1, 2))
    if ok is not None:
        finite &= ok
    err = np.full(len(E_dev), 0.0)
    err[finite] = (np.max(np.abs(E_dev[finite] - E_ref[finite]), axis=(1, 2))
                   / np.max(np.abs(E_ref[finite]), axis=(1, 2)))
    assert np.median(err[finite]) <= median, np.median(err[finite])
    loose = np.nonzero(err > hard)[0]
    assert len(loose) <= mp_budget, (len(loose), len(err))   # a handful per 100 000, not a population
    if len(loose):
        from oracle.fit_mp import fit_eight_point_mp

        for i in loose:
            pts = corr[S[i]]
            truth, _ = fit_eight_point_mp(pts[:, 0:2], pts[:, 2:4])
            scale = np.max(np.abs(truth))
            d_dev = np.max(np.abs(E_dev[i] - truth)) / scale
            d_ref = np.max(np.abs(E_ref[i] - truth)) / scale
            assert d_dev <= 1e-9 and d_dev <= d_ref, (int(i), err[i], d_dev, d_ref)
    return err


def assert_counts_explained(corr, S, thr, E_dev, E_ref, cnt_dev, cnt_ref, fit_err, min_agree=0.999):
    """Counts scored with the device's own fits vs counts scored with the oracle's fits: equal for (almost) every
    hypothesis, and every disagreement is explained point by point — each flipped point's SED lies within the fit
    difference of the threshold.  A relative perturbation eps of E moves r = b^T E a of a point AT the threshold
    (|r| = sqrt(thr d / 2), d ~ |E|^2) by up to eps |E| |a| |b|, i.e. its SED by a relative 2 eps / sqrt(thr / 2) ~ 2 300 eps
    at thr = 1.5e-6: the bar is |sed - thr| / thr <= 1e4 max(rel(E_dev, E_ref), 1e-13)."""
    differ = np.nonzero(cnt_dev != cnt_ref)[0]
    assert len(differ) <= (1.0 - min_agree) * len(cnt_dev), (len(differ), len(cnt_dev))
    for i in differ:
        sed_d = orc.sed_values(E_dev[i].reshape(3, 3), corr)
        sed_r = orc.sed_values(E_ref[i].reshape(3, 3), corr)
        with np.errstate(invalid="ignore"):
            flipped = np.nonzero((sed_d <= thr) != (sed_r <= thr))[0]
        flipped = flipped[~np.isin(flipped, S[i])]
        assert len(flipped) >= abs(int(cnt_dev[i]) - int(cnt_ref[i]))
        margin = 1e4 * max(fit_err[i], 1e-13)
        for j in flipped:
            assert min(abs(sed_d[j] - thr), abs(sed_r[j] - thr)) / thr <= margin, (int(i), int(j), sed_d[j], sed_r[j], margin)
    return len(differ)


def test_full_size_properties(dev, c_oracle_lib):
    """BASELINE configs[2] (50 000 x 100 000) against the oracle on EVERY hypothesis (the H x N loop of reference
    ransac.py:66-86; the C/OpenMP oracle scores the whole configuration in about a second on the box's cores):
      * sample table == the oracle's Philox table, all 100 000 rows;
      * scoring: the oracle scores the device's own E -> counts bit-equal for all 100 000 hypotheses, both sums to
        summation order (rtol 1e-12); winner == select_best over the ORACLE's aggregates; the winner's inlier index
        set bit-equal to the oracle's;
      * fit: every E against the numpy oracle (LAPACK route) <= 1e-6 or settled by 40-digit arithmetic; flags equal;
      * the whole reference pipeline on the CPU (oracle fit -> oracle score -> select) picks the same winner with the
        same ordered inlier list, and per-hypothesis counts agree up to points within the fit accuracy of the threshold."""
    from structure_from_motion_amd._native import AGG_RMS

    n, h = 50_000, 100_000
    _, _, _, corr = scene(n)
    thr, min_extra = 1.5e-6, 10
    ws = dev.RansacWorkspace(1, n, h)
    dev.sample_philox(5, 0, h, n, out=ws.S)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    ws.run(corr_d, thr, min_extra, AGG_RMS)
    out = ws.outcome(0)
    cnt = ws.cnt.cpu().numpy()[0]
    s1 = ws.s1.cpu().numpy()[0]
    s2 = ws.s2.cpu().numpy()[0]
    S = ws.S.cpu().numpy()[0]
    E = ws.E.cpu().numpy()[0]
    flags = ws.flags.cpu().numpy()[0]
    np.testing.assert_array_equal(S, orc.philox_sample_table(5, 0, h, n))
    # scoring of every hypothesis, same E on both sides
    cnt_o, s1_o, s2_o = c_oracle_lib.score(corr, E, S, thr)
    np.testing.assert_array_equal(cnt, cnt_o)
    np.testing.assert_allclose(s1, s1_o, rtol=1e-12, atol=0)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-12, atol=0)
    best, best_err = orc.select_best(orc.aggregate(cnt_o, s1_o, s2_o, orc.RMS), cnt_o, min_extra)
    assert out.best_h == best and abs(out.error - best_err) <= 1e-12 * best_err
    want = orc.inlier_indices(corr, E[best].reshape(3, 3), S[best], thr)
    np.testing.assert_array_equal(np.nonzero(out.mask == 1)[0], want[8:])
    np.testing.assert_array_equal(np.sort(np.nonzero(out.mask == 2)[0]), np.sort(S[best]))
    assert int((out.mask == 1).sum()) == cnt[best]
    # the fit of every hypothesis, and the reference pipeline end to end on the CPU
    with orc.FitPool(corr, c_oracle_lib.threads) as pool:
        E_o, deg_o, _ = pool.fit(S)
    np.testing.assert_array_equal(flags != 0, deg_o)
    fit_err = assert_fits_agree(corr, S, E, E_o, ok=~deg_o)
    cnt_r, s1_r, s2_r = c_oracle_lib.score(corr, E_o, S, thr)
    best_r, err_r = orc.select_best(orc.aggregate(cnt_r, s1_r, s2_r, orc.RMS), cnt_r, min_extra)
    assert best_r == out.best_h and abs(out.error - err_r) <= 1e-8 * err_r
    np.testing.assert_array_equal(orc.inlier_indices(corr, E_o[best_r], S[best_r], thr), want)
    assert rel(out.E, E_o[best_r]) <= 1e-6
    assert_counts_explained(corr, S, thr, E.reshape(-1, 3, 3), E_o, cnt, cnt_r, fit_err)


# ------------------------------------------------------------------------------------------------------
# pose recovery and triangulation
# ------------------------------------------------------------------------------------------------------
def pose_sets_equal(poses, R1, R2, t1, atol):
    """The device's 4 candidates equal {R1,R2} x {t,-t} as a set."""
    got = [(p[:9].reshape(3, 3), p[9:]) for p in poses]
    want = [(R1, t1), (R1, -t1), (R2, t1), (R2, -t1)]
    for Rw, tw in want:
        if not any(np.allclose(Rg, Rw, atol=atol) and np.allclose(tg, tw, atol=atol) for Rg, tg in got):
            return False
    return True


def test_decompose_essential(dev, golden):
    for name in ("g1_eight_point", "g5_cheirality"):
        d = golden(name)
        poses, status = dev.decompose_essential(dev.to_device(d["E"].reshape(1, 9)))
        assert int(status.cpu()[0]) == 0
        poses = poses.cpu().numpy()[0]
        assert pose_sets_equal(poses, d["R1"], d["R2"], d["t1"], 1e-10)
        for p in poses:
            R = p[:9].reshape(3, 3)
            np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-13)
            assert abs(np.linalg.det(R) - 1.0) <= 1e-13 and abs(np.linalg.norm(p[9:]) - 1.0) <= 1e-13
    bad = np.eye(3).reshape(1, 9)  # full rank: smallest singular value is not ~0
    _, status = dev.decompose_essential(dev.to_device(bad))
    assert int(status.cpu()[0]) == 1


def test_cheirality_matches_reference_golden(dev, golden):
    d = golden("g5_cheirality")
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(d["pix_a"], d["K"]),
                                    orc.to_normalized_image_coords(d["pix_b"], d["K"]))
    import itertools
    poses = np.array([np.concatenate([R.reshape(9), t]) for R, t in
                      itertools.product([d["R1"], d["R2"]], [d["t1"], -d["t1"]])])
    got = dev.cheirality(dev.to_device(corr), dev.to_device(poses), 50.0).cpu().numpy()
    np.testing.assert_array_equal(got, d["passes"])


def test_cheirality_matches_oracle_large(dev):
    n = 20000
    pa, pb, K, corr = scene(n, seed=31, outliers=0.25)
    _, _, _, R, t, _ = orc.synthetic_two_view(n, seed=31, outlier_fraction=0.25)
    tn = t / np.linalg.norm(t)
    Tx = np.array([[0, -tn[2], tn[1]], [tn[2], 0, -tn[0]], [-tn[1], tn[0], 0]])
    E = Tx @ R
    R1, R2, t1 = orc.recover_all_r_t(E / E[2, 2])
    import itertools
    poses = np.array([np.concatenate([Rc.reshape(9), tc]) for Rc, tc in itertools.product([R1, R2], [t1, -t1])])
    got = dev.cheirality(dev.to_device(corr), dev.to_device(poses), 50.0).cpu().numpy()
    for c, (Rc, tc) in enumerate(itertools.product([R1, R2], [t1, -t1])):
        want = orc.cheirality_pass(corr, Rc, tc)
        mism = np.nonzero(got[c].astype(bool) != want)[0]
        assert len(mism) == 0, (c, mism[:10])


def test_triangulate_matches_oracle_and_golden(dev, golden):
    d = golden("g6_triangulate")
    K_ext = np.hstack((d["K"], np.zeros((3, 1))))
    P1 = K_ext @ np.eye(4)
    P2 = K_ext @ d["cam2_T_cam1"]
    corr = orc.pack_correspondences(d["pix_a"], d["pix_b"])
    X = dev.triangulate(dev.to_device(corr), dev.to_device(P1.reshape(12)), dev.to_device(P2.reshape(12))).cpu().numpy()
    near = slice(0, 60)
    assert np.max(np.abs(X[near] - d["X"][near]) / np.abs(d["X"][near])) <= 1e-6   # north_star bar
    assert np.max(np.abs(X[near] - d["X"][near]) / np.linalg.norm(d["X"][near], axis=1, keepdims=True)) <= 1e-10
    # low-parallax points (z 200..2000 baselines): ill-conditioned, compare at the conditioning-scaled bar
    far = slice(60, 66)
    assert np.max(np.abs(X[far] - d["X"][far]) / np.linalg.norm(d["X"][far], axis=1, keepdims=True)) <= 1e-6
    Xk = dev.triangulate(dev.to_device(orc.pack_correspondences(d["known_a"][None], d["known_b"][None])),
                         dev.to_device(d["known_P1"].reshape(12)), dev.to_device(d["known_P2"].reshape(12))).cpu().numpy()[0]
    np.testing.assert_allclose(Xk, [0.0, 0.0, 10.0], atol=1e-10)  # reference test_triangulate bar


def test_triangulate_large_vs_oracle(dev):
    n = 30000
    pa, pb, K, _ = scene(n, seed=77, outliers=0.0)
    _, _, _, R, t, _ = orc.synthetic_two_view(n, seed=77, outlier_fraction=0.0)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    want = orc.triangulate_points(pa, pb, K, T)
    K_ext = np.hstack((K, np.zeros((3, 1))))
    got = dev.triangulate(dev.to_device(orc.pack_correspondences(pa, pb)), dev.to_device((K_ext @ np.eye(4)).reshape(12)),
                          dev.to_device((K_ext @ T).reshape(12))).cpu().numpy()
    assert np.max(np.abs(got - want) / np.linalg.norm(want, axis=1, keepdims=True)) <= 1e-9


def test_triangulate_mixed_waves_with_outliers(dev):
    """30 % gross outliers: waves mix lanes whose inverse iteration converges in three steps with lanes that never
    converge, so the QR route and the per-wave Jacobi fallback both run and their results are merged per lane.
    Inliers must match the oracle's SVD point for point; for the outliers (sigma_3 ~ sigma_4: the point itself is
    ill-determined) the returned vector must still be a minimiser: |A x| / |x| == sigma_4."""
    n = 20000
    pa, pb, K, R, t, is_out = orc.synthetic_two_view(n, seed=31, outlier_fraction=0.3)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    K_ext = np.hstack((K, np.zeros((3, 1))))
    P1, P2 = K_ext @ np.eye(4), K_ext @ T
    corr = orc.pack_correspondences(pa, pb)
    got = dev.triangulate(dev.to_device(corr), dev.to_device(P1.reshape(12)), dev.to_device(P2.reshape(12))).cpu().numpy()
    want = orc.triangulate_points(pa, pb, K, T)
    inl = ~is_out
    assert np.max(np.abs(got[inl] - want[inl]) / np.linalg.norm(want[inl], axis=1, keepdims=True)) <= 1e-9
    A = orc.dlt_matrix(corr, P1[:3], P2[:3])                       # (n, 4, 4)
    x = np.concatenate([got, np.ones((n, 1))], axis=1)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    residual = np.linalg.norm(np.einsum("nij,nj->ni", A, x), axis=1)
    sv = np.linalg.svd(A, compute_uv=False)
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(residual - sv[:, 3]) / sv[:, 0]) <= 1e-9


def test_triangulate_random_geometries(dev):
    """40 random camera pairs (baselines from 1 % to 200 % of the scene depth, rotations up to 60 degrees, focal
    lengths 300..3000 px, noise 0..2 px): every returned point is a minimiser of |A x| (== sigma_4 of the DLT matrix,
    LAPACK) and well-conditioned points agree with the SVD solution to 1e-9."""
    rng = np.random.default_rng(12)
    n = 1500
    for trial in range(40):
        f = rng.uniform(300, 3000)
        K = np.array([[f, 0, 320.0], [0, f * rng.uniform(0.9, 1.1), 240.0], [0, 0, 1.0]])
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, np.pi / 3)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        depth = rng.uniform(2, 20)
        t = rng.normal(size=3)
        t *= depth * 10.0 ** rng.uniform(-2, 0.3) / np.linalg.norm(t)
        X = np.column_stack([rng.uniform(-1, 1, n) * depth / 3, rng.uniform(-1, 1, n) * depth / 3,
                             rng.uniform(0.7, 1.3, n) * depth])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        K_ext = np.hstack((K, np.zeros((3, 1))))
        P1, P2 = K_ext @ np.eye(4), K_ext @ T
        Xh = np.column_stack([X, np.ones(n)])
        pa = (P1 @ Xh.T).T
        pb = (P2 @ Xh.T).T
        noise = rng.uniform(0, 2)
        pa = pa[:, :2] / pa[:, 2:3] + rng.normal(scale=noise + 1e-12, size=(n, 2))
        pb = pb[:, :2] / pb[:, 2:3] + rng.normal(scale=noise + 1e-12, size=(n, 2))
        corr = orc.pack_correspondences(pa, pb)
        got = dev.triangulate(dev.to_device(corr), dev.to_device(P1.reshape(12)), dev.to_device(P2.reshape(12))).cpu().numpy()
        A = orc.dlt_matrix(corr, P1[:3], P2[:3])
        u, sv, vt = np.linalg.svd(A)
        x = np.column_stack([got, np.ones(n)])
        finite = np.all(np.isfinite(x), axis=1)
        assert finite.mean() > 0.999
        xn = x[finite] / np.linalg.norm(x[finite], axis=1, keepdims=True)
        residual = np.linalg.norm(np.einsum("nij,nj->ni", A[finite], xn), axis=1)
        assert np.max(np.abs(residual - sv[finite, 3]) / sv[finite, 0]) <= 1e-9, trial
        want = vt[:, 3, :3] / vt[:, 3, 3:4]
        well = finite & (sv[:, 3] < 1e-3 * sv[:, 2]) & (np.abs(vt[:, 3, 3]) > 1e-6)
        if well.any():
            assert np.max(np.abs(got[well] - want[well]) / np.linalg.norm(want[well], axis=1, keepdims=True)) <= 1e-9, trial


# ------------------------------------------------------------------------------------------------------
# sharding: G virtual shards on one GPU through the sharded engine == one run over all hypotheses
# ------------------------------------------------------------------------------------------------------
def run_virtual_shards(dev, corr_d, total, world, seed, thr=1.5e-6, min_extra=10, spoil=None):
    """``world`` virtual ranks on one GPU: each runs ShardedRansac.step_local over its shard_range of ``total``
    hypotheses; their records are stacked exactly as the all-gather would deliver them and every rank finishes
    from that.  ``spoil(rank, engine)`` may tamper with a rank's buffers between the fit and the selection.
    Returns the engines."""
    from structure_from_motion_amd import distributed
    from structure_from_motion_amd._native import AGG_RMS

    engines = [distributed.ShardedRansac(corr_d, None, thr, min_extra, AGG_RMS, rank=r, world=world,
                                         total_hypotheses=total) for r in range(world)]
    for r, eng in enumerate(engines):
        eng.step_local(seed)
        if spoil is not None and spoil(r, eng):
            # redo the selection over the tampered buffers (same call the local pass makes)
            dev.select_best(eng.ws.cnt, eng.ws.s1, eng.ws.s2, eng.ws.flags, min_extra, AGG_RMS, eng.h_begin,
                            eng.ws.result)
    gathered = torch.stack([eng.ws.result for eng in engines])   # [world, 1, 5], rank-major
    for eng in engines:
        eng.finish(seed, gathered)
    return engines


def test_virtual_shards_equal_single_run(dev):
    from structure_from_motion_amd import distributed
    from structure_from_motion_amd._native import AGG_RMS

    n, total, G, seed = 1500, 1601, 4, 21   # 1601: the last shard is shorter than the others
    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr)
    whole = distributed.ShardedRansac(corr_d, total, 1.5e-6, 10, AGG_RMS, rank=0, world=1)
    whole.step(seed)
    want = whole.outcome()
    engines = run_virtual_shards(dev, corr_d, total, G, seed)
    assert [e.h for e in engines] == [401, 401, 401, 398]
    for eng in engines:   # every rank arrives at the same winner, E, sample and mask — those of the single run
        got = eng.outcome()
        assert got.best_h == want.best_h and got.error == want.error and got.n_flagged == 0
        np.testing.assert_array_equal(got.E, want.E)
        np.testing.assert_array_equal(got.sample, want.sample)
        np.testing.assert_array_equal(got.mask, want.mask)
    ref = orc.ransac_essential(corr, orc.philox_sample_table(seed, 0, total, n), 1.5e-6, 10, orc.RMS)
    assert ref["best"] == want.best_h
    np.testing.assert_array_equal(np.nonzero(want.mask)[0], np.sort(ref["inliers"]))
    # more ranks than hypotheses: empty shards publish "no model" records and the fold ignores them
    tiny = run_virtual_shards(dev, corr_d, 5, 8, seed, min_extra=0)
    assert [e.h for e in tiny] == [1, 1, 1, 1, 1, 0, 0, 0]
    ref = orc.ransac_essential(corr, orc.philox_sample_table(seed, 0, 5, n), 1.5e-6, 0, orc.RMS)
    assert all(e.outcome().best_h == ref["best"] for e in tiny)


def test_fold_kernel_equals_host_fold(dev):
    """sfm_fold_select_records (device kernel) == sfm_fold_select_records_host on random gathered records with ties
    across ranks, no-model ranks and flag statistics: the CPU process-group tests exercise the same routine the GPUs run."""
    from structure_from_motion_amd import distributed
    from structure_from_motion_amd._native import INT64_MAX

    rng = np.random.default_rng(17)
    world, batch = 7, 96
    gathered = np.zeros((world, batch, 5), dtype=np.int64)
    errs = rng.choice(np.array([1e-9, 2e-9, 3.5e-7, 0.0, 1.0]), size=(world, batch))   # few distinct values: many ties
    none = rng.random((world, batch)) < 0.3
    for r in range(world):
        for b in range(batch):
            if none[r, b]:
                key, best, err, cnt = INT64_MAX, -1, np.inf, 0
            else:
                err = errs[r, b]
                key, best, cnt = int(np.float64(err).view(np.int64)), int(rng.integers(0, 10**7)) + r * 10**7, int(rng.integers(10, 500))
            flagged = int(rng.integers(0, 3))
            first = int(rng.integers(0, 10**7)) + r * 10**7 if flagged else INT64_MAX
            words = np.frombuffer(np.array([key], dtype=np.uint64).tobytes() + np.array([best], dtype=np.int64).tobytes()
                                  + np.array([err], dtype=np.float64).tobytes() + np.array([first], dtype=np.int64).tobytes()
                                  + np.array([flagged, cnt], dtype=np.int32).tobytes(), dtype=np.int64)
            gathered[r, b] = words
    host = distributed.fold_records(torch.from_numpy(gathered.copy()))
    device_out = distributed.fold_records(dev.to_device(gathered, torch.int64))
    for a, b in zip(host, device_out):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())
    # and the rule itself, record by record
    recs = distributed.read_records(host[0])
    for b, rec in enumerate(recs):
        cand = [(gathered[r, b, 0], gathered[r, b, 1]) for r in range(world) if gathered[r, b, 1] >= 0]
        want = min(cand) if cand else (INT64_MAX, -1)
        assert (rec.key, rec.best_h) == (want[0], want[1])
        assert rec.n_flagged == int(sum(gathered[r, b, 4] & 0xFFFFFFFF for r in range(world)))


def test_virtual_shards_degenerate_sample_reaches_every_rank(dev):
    """A degenerate sample on ONE rank (eight_point.py:415-421) must abort the call on EVERY rank, as it does
    in the reference (ransac.py:65) and in the single-GPU drop-in path; SFM_DEGENERATE=skip drops the hypothesis
    everywhere instead — also when it would have been the winner."""
    from structure_from_motion_amd.epipolar.eight_point import EightPointCalculationError

    n, total, G, seed = 1200, 800, 4, 33
    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr)
    clean = run_virtual_shards(dev, corr_d, total, G, seed)
    winner = clean[0].outcome()
    owner = winner.best_h // 200
    victim = winner.best_h - owner * 200

    def spoil(rank, eng):   # flag the would-be winner as degenerate on its own rank
        if rank != owner:
            return False
        eng.ws.flags[0, victim] = 1
        return True

    spoiled = run_virtual_shards(dev, corr_d, total, G, seed, spoil=spoil)
    for eng in spoiled:
        with pytest.raises(EightPointCalculationError, match=f"hypothesis {winner.best_h}, 1 in total"):
            eng.outcome()
        with pytest.raises(EightPointCalculationError):
            eng.outcome(policy="raise")
    # skip: the flagged hypothesis does not compete; the runner-up of the sequential rule wins on every rank
    S = orc.philox_sample_table(seed, 0, total, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    cnt, s1, s2 = orc.score_hypotheses(corr, E, S, 1.5e-6)
    err = orc.aggregate(cnt, s1, s2, orc.RMS)
    err[winner.best_h] = np.inf
    runner_up, _ = orc.select_best(err, cnt, 10)
    outs = [eng.outcome(policy="skip") for eng in spoiled]
    assert all(o.best_h == runner_up and o.n_flagged == 1 and o.first_flagged == winner.best_h for o in outs)
    for o in outs[1:]:
        np.testing.assert_array_equal(o.E, outs[0].E)
        np.testing.assert_array_equal(o.mask, outs[0].mask)


def test_c4_virtual_shards_full_size(dev, c_oracle_lib):
    """BASELINE config 4 at its real shape — 1 000 000 hypotheses over 8 ranks (125 000 each) x 50 000
    correspondences — as 8 virtual ranks on one GPU, against the oracle on EVERY hypothesis (5 * 10^10 evaluations of
    the C/OpenMP oracle, ~10 s on the box's cores; the loop of reference ransac.py:61-86 cut into 8 contiguous blocks):
      * per shard: sample table == the global Philox stream at the shard's offset (all rows); the oracle scores the
        shard's own E -> counts bit-equal for all 125 000, sums to summation order; the shard's record == select_best
        over the oracle's aggregates (global index); every fit against the numpy oracle (<= 1e-6 or 40-digit tie-break),
        degeneracy flags equal;
      * the fold == the sequential rule over all 1 000 000 oracle aggregates (lowest error, earliest index), and the
        whole reference pipeline on the CPU (oracle fit -> oracle score -> select over 1 000 000) picks the same
        winner with the same inlier index set;
      * the winner re-derived from (seed, h*) on every rank equals the owning shard's own E / S bit for bit, with
        global indices beyond 2^19 in play."""
    from structure_from_motion_amd import distributed
    from structure_from_motion_amd._native import AGG_RMS, INT64_MAX

    n, total, G, seed = 50_000, 1_000_000, 8, 5
    thr, min_extra = 1.5e-6, 10
    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr)
    engines = run_virtual_shards(dev, corr_d, total, G, seed)
    assert [(e.h_begin, e.h) for e in engines] == [(r * 125_000, 125_000) for r in range(G)]
    records = []
    err_dev_all, err_ref_all, cnt_ref_all = [], [], []
    with orc.FitPool(corr, c_oracle_lib.threads) as pool:
        for r, eng in enumerate(engines):
            rec = distributed.read_records(eng.ws.result)[0]
            records.append(rec)
            cnt = eng.ws.cnt.cpu().numpy()[0]
            s1 = eng.ws.s1.cpu().numpy()[0]
            s2 = eng.ws.s2.cpu().numpy()[0]
            S = eng.ws.S.cpu().numpy()[0]
            E = eng.ws.E.cpu().numpy()[0]
            np.testing.assert_array_equal(S, orc.philox_sample_table(seed, eng.h_begin, eng.h, n))
            cnt_o, s1_o, s2_o = c_oracle_lib.score(corr, E, S, thr)
            np.testing.assert_array_equal(cnt, cnt_o)
            np.testing.assert_allclose(s1, s1_o, rtol=1e-12, atol=0)
            np.testing.assert_allclose(s2, s2_o, rtol=1e-12, atol=0)
            err_o = orc.aggregate(cnt_o, s1_o, s2_o, orc.RMS)
            best, best_err = orc.select_best(err_o, cnt_o, min_extra)
            assert rec.best_h == best + eng.h_begin and rec.best_cnt == cnt[best]
            assert abs(rec.best_err - best_err) <= 1e-12 * best_err
            assert rec.n_flagged == 0 and rec.first_flagged == INT64_MAX
            err_dev_all.append(np.where(cnt_o >= min_extra, err_o, np.inf))
            # the reference route for the same 125 000 samples
            E_o, deg_o, _ = pool.fit(S)
            np.testing.assert_array_equal(eng.ws.flags.cpu().numpy()[0] != 0, deg_o)
            fit_err = assert_fits_agree(corr, S, E, E_o, ok=~deg_o)
            cnt_r, s1_r, s2_r = c_oracle_lib.score(corr, E_o, S, thr)
            assert_counts_explained(corr, S, thr, E.reshape(-1, 3, 3), E_o, cnt, cnt_r, fit_err)
            err_ref_all.append(orc.aggregate(cnt_r, s1_r, s2_r, orc.RMS))
            cnt_ref_all.append(cnt_r)
            del E, E_o
    # fold == the sequential rule over all 1 000 000 hypotheses: lowest error, then lowest global index
    err_dev_all = np.concatenate(err_dev_all)
    want_h = int(np.argmin(err_dev_all))
    assert max(rec.best_h for rec in records) >= 2**19   # global indices beyond 2^19 went through the reducer
    owner = engines[want_h // 125_000]
    local = want_h - owner.h_begin
    E_own = owner.ws.E[0, local].cpu().numpy().reshape(3, 3)
    S_own = owner.ws.S[0, local].cpu().numpy().astype(np.int64)
    outs = [eng.outcome() for eng in engines]
    for o in outs:
        assert o.best_h == want_h and abs(o.error - err_dev_all[want_h]) <= 1e-12 * o.error
        np.testing.assert_array_equal(o.E, E_own)        # re-derived winner == the owning shard's own fit, bit for bit
        np.testing.assert_array_equal(o.sample, S_own)
        np.testing.assert_array_equal(o.mask, outs[0].mask)
    assert int((outs[0].mask == 2).sum()) == 8
    assert int((outs[0].mask == 1).sum()) == int(owner.ws.cnt[0, local].cpu())
    np.testing.assert_array_equal(np.nonzero(outs[0].mask == 2)[0], np.sort(S_own))
    # the reference pipeline end to end on the CPU picks the same winner and the same inlier index set
    best_ref, _ = orc.select_best(np.concatenate(err_ref_all), np.concatenate(cnt_ref_all), min_extra)
    assert best_ref == want_h
    E_ref_best = orc.fit_hypotheses(corr, S_own[None, :])[0][0]
    assert rel(outs[0].E, E_ref_best) <= 1e-6
    np.testing.assert_array_equal(np.nonzero(outs[0].mask == 1)[0], orc.inlier_indices(corr, E_ref_best, S_own, thr)[8:])
    # a re-derivation far beyond 2^19 (the last hypothesis of the stream) equals the oracle's sample
    last = torch.tensor([total - 1], dtype=torch.int64, device=corr_d.device)
    np.testing.assert_array_equal(dev.sample_philox_at(seed, last, n).cpu().numpy().reshape(8),
                                  orc.philox_sample_table(seed, total - 1, 1, n)[0])


@pytest.mark.parametrize("n,h,philox", [(8, 1, True), (130, 3, False), (300, 2000, True), (777, 5, True),
                                        (1000, 64, False), (5000, 10000, True), (8192, 32768, True), (4099, 1300, False),
                                        (600, 4097, True), (2500, 5121, False),
                                        # two hypotheses per wave (>= 10240 hypotheses) with odd step counts and tail chunks
                                        (1001, 11000, True), (450, 10241, False),
                                        # one-hypothesis waves with the block barrier in their loop (>= 4096 points) and a last
                                        # block whose surplus waves have ended before the first barrier
                                        (4100, 4099, True), (6000, 7001, False), (8191, 9, True)])
def test_fused_small_pass_equals_separate_calls(dev, monkeypatch, n, h, philox):
    """sfm_ransac_pass_small (workspace preparation inside the fit launch, scoring from per-block partial maxima,
    selection spread over up to 32 blocks folded by the last arriver) against the five separate calls on the same
    inputs: samples, E, flags, counts, sums, winner record and mask identical; and against the oracle where it
    finishes in seconds."""
    from structure_from_motion_amd._native import AGG_RMS, AGG_SUM

    _, _, _, corr = scene(n, seed=40 + n % 7)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    thr, min_extra = 1.5e-6, 10 if n >= 300 else 0
    table = orc.philox_sample_table(9, 100, h, n)
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SFM_SMALL_PASS", fused)
        assert dev.small_pass_eligible(1, n, h) == (fused == "1")
        ws = dev.RansacWorkspace(1, n, h)
        if philox:
            ws.run(corr_d, thr, min_extra, AGG_RMS, philox=(9, 100, 1))
        else:
            ws.S.copy_(dev.to_device(table, torch.int32).reshape(1, h, 8))
            ws.run(corr_d, thr, min_extra, AGG_SUM)
        outs.append({k: getattr(ws, k).cpu().numpy().copy() for k in ("S", "E", "flags", "cnt", "s1", "s2", "result", "mask")})
    fused, plain = outs
    for key in ("S", "E", "flags", "cnt", "mask"):
        np.testing.assert_array_equal(fused[key], plain[key], err_msg=key)
    np.testing.assert_allclose(fused["s1"], plain["s1"], rtol=1e-13, atol=0, equal_nan=True)
    np.testing.assert_allclose(fused["s2"], plain["s2"], rtol=1e-13, atol=0, equal_nan=True)
    np.testing.assert_array_equal(fused["S"][0], table)
    rec_f, rec_p = fused["result"][0], plain["result"][0]
    # winner, flag statistics and count identical; the error (words 0 and 2: key, best_err) to summation order
    assert rec_f[1] == rec_p[1] and rec_f[3] == rec_p[3] and rec_f[4] == rec_p[4]
    if rec_p[1] >= 0:
        assert abs(rec_f[2:3].view(np.float64)[0] / rec_p[2:3].view(np.float64)[0] - 1.0) <= 1e-13
    if n * h <= 13_000_000:
        ref = orc.ransac_essential(corr, table, thr, min_extra, orc.RMS if philox else orc.SUM)
        assert ref["best"] == rec_f[1]
        np.testing.assert_array_equal(fused["cnt"][0], ref["cnt"])
        if ref["best"] >= 0:
            np.testing.assert_array_equal(np.nonzero(fused["mask"][0])[0], np.sort(ref["inliers"]))


@pytest.mark.parametrize("n,h,philox,h_offset", [(10_000, 50_000, True, 0), (9_000, 70_001, False, 0), (50_000, 20_000, True, 0),
                                                 (8_192, 65_000, True, 1_000_000), (16_384, 33_000, False, 0)])
def test_fused_large_pass_equals_separate_calls(dev, monkeypatch, n, h, philox, h_offset):
    """sfm_ransac_pass_large — seven launches: partial maxima + zeroing, the fits whose lanes also write the hypotheses' operand
    rows and sample corrections (+ blocks that write the point table), cost pre-pass, class histogram, scan + scatter, the
    matrix-pipe scoring kernel, fold of the point ranges + selection over up to 256 blocks + mask — against the separate calls
    (SFM_LARGE_PASS=0: eighteen launches) on the same inputs: samples, E, flags, counts, both sums, the winner record and the
    mask identical bit for bit (the same fit code, the same table routines, the ranges added in the same order)."""
    from structure_from_motion_amd._native import AGG_RMS, AGG_SUM

    _, _, _, corr = scene(n, seed=40 + n % 7)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    thr, min_extra = 1.5e-6, 10
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SFM_LARGE_PASS", fused)
        assert dev.large_pass_eligible(1, n, h) == (fused == "1")
        ws = dev.RansacWorkspace(1, n, h)
        ws.mask.fill_(7)
        if philox:
            ws.run(corr_d, thr, min_extra, AGG_RMS, philox=(9, 100, 1), h_offset=h_offset, with_mask=h_offset == 0)
        else:
            ws.S.copy_(dev.sample_philox(11, 5, h, n))
            ws.run(corr_d, thr, min_extra, AGG_SUM, h_offset=h_offset, with_mask=h_offset == 0)
        outs.append({k: getattr(ws, k).cpu().numpy().copy() for k in ("S", "E", "flags", "cnt", "s1", "s2", "result", "mask")})
    fused, plain = outs
    for key in ("S", "E", "flags", "cnt", "result") + (("mask",) if h_offset == 0 else ()):
        np.testing.assert_array_equal(fused[key], plain[key], err_msg=key)
    for key in ("s1", "s2"):
        np.testing.assert_array_equal(fused[key].view(np.int64), plain[key].view(np.int64), err_msg=key)
    assert fused["result"][0][1] >= h_offset                                  # a model was found, global index
    if h_offset == 0:
        assert set(np.unique(fused["mask"]).tolist()) <= {0, 1, 2} and (fused["mask"] == 2).sum() == 8
    # ... and the counts are the all-fp64 kernel's
    exact = dev.score_sed(corr_d, dev.to_device(fused["E"]), dev.to_device(fused["S"], torch.int32), thr, exact_only=True)
    np.testing.assert_array_equal(exact[0].cpu().numpy(), fused["cnt"])


@pytest.mark.parametrize("batch,n,h,kernel,philox", [(64, 10_000, 2_000, None, True), (17, 2_100, 40, "matrix", True),
                                                     (9, 8_300, 1_100, "matrix", False), (5, 700, 70, None, True),
                                                     (3, 4_099, 300, "filtered", False), (40, 9_000, 2_064, "matrix", True)])
def test_fused_batch_pass_equals_separate_calls(dev, monkeypatch, batch, n, h, kernel, philox):
    """sfm_ransac_pass_batch — partial maxima + zeroing, the pairs' point tables, the fits whose lanes also write the hypotheses'
    operand rows and sample corrections, pre-pass, sort, scoring, one block per pair for fold + selection + mask — against the
    separate calls (SFM_LARGE_PASS=0) on the same inputs: samples, E, flags, counts, both sums, records and masks bit for bit,
    with the matrix-pipe kernel (by size or forced) and with the VALU-filter kernel; counts equal to the all-fp64 kernel's."""
    from structure_from_motion_amd._native import AGG_RMS, AGG_SUM

    corr = np.stack([scene(n, seed=50 + b)[3] for b in range(min(batch, 6))])
    corr = np.ascontiguousarray(corr[np.arange(batch) % len(corr)])
    corr_d = dev.to_device(corr)
    thr, min_extra = 1.5e-6, 10
    saved = dev.default_score_options()
    outs = []
    try:
        if kernel is not None:
            dev.set_default_score_options(_options(kernel=kernel))
        from structure_from_motion_amd import _native
        want = {None: None, "matrix": 2, "filtered": 1}[kernel]
        if want is not None:
            assert _native.load().sfm_score_kernel_choice(n, h, batch) == want
        for fused in ("1", "0"):
            monkeypatch.setenv("SFM_LARGE_PASS", fused)
            assert dev.batch_pass_eligible(batch) == (fused == "1")
            ws = dev.RansacWorkspace(batch, n, h)
            ws.mask.fill_(7)
            if philox:
                ws.run(corr_d, thr, min_extra, AGG_RMS, philox=(9, 100, 3))
            else:
                ws.S.copy_(dev.sample_philox(11, 5, h, n, batch=batch, seed_stride=2))
                ws.run(corr_d, thr, min_extra, AGG_SUM)
            torch.cuda.synchronize()
            outs.append({k: getattr(ws, k).cpu().numpy().copy() for k in ("S", "E", "flags", "cnt", "s1", "s2", "result", "mask")})
    finally:
        dev.set_default_score_options(saved)
    fused, plain = outs
    for key in ("S", "E", "flags", "cnt", "result", "mask"):
        np.testing.assert_array_equal(fused[key], plain[key], err_msg=key)
    for key in ("s1", "s2"):
        np.testing.assert_array_equal(fused[key].view(np.int64), plain[key].view(np.int64), err_msg=key)
    assert set(np.unique(fused["mask"]).tolist()) <= {0, 1, 2}
    found = fused["result"][:, 1] >= 0
    assert found.sum() >= batch // 2 and np.all((fused["mask"][found] == 2).sum(axis=1) == 8)
    exact = dev.score_sed(corr_d, dev.to_device(fused["E"]), dev.to_device(fused["S"], torch.int32), thr, exact_only=True)
    np.testing.assert_array_equal(exact[0].cpu().numpy(), fused["cnt"])


def test_fused_batch_and_large_pass_random_sizes(dev, monkeypatch):
    """The fused passes against the separate calls on random small shapes — ragged last steps, fewer points than one setup
    block, one hypothesis group, a single hypothesis, batches of one, no mask — with the matrix-pipe kernel forced and with the
    VALU filter: every output bit for bit."""
    from structure_from_motion_amd._native import AGG_MEAN, AGG_RMS

    # (a soak: SFM_SOAK_SHAPES=300 SFM_SOAK_N_MAX=40000 SFM_SOAK_H_MAX=3000 SFM_SOAK_SEED=7 — profiles/r05/soak_fused_passes.txt)
    rng = np.random.default_rng(int(os.environ.get("SFM_SOAK_SEED", 2025)))
    n_max, h_max = int(os.environ.get("SFM_SOAK_N_MAX", 3000)), int(os.environ.get("SFM_SOAK_H_MAX", 200))
    shapes = [(1, 8, 1), (1, 40, 33), (3, 31, 1), (2, 257, 64), (7, 1000, 65), (4, 2049, 31), (1, 4100, 97), (5, 640, 130)]
    shapes += [(int(rng.integers(1, 9)), int(rng.integers(8, n_max)), int(rng.integers(1, h_max)))
               for _ in range(int(os.environ.get("SFM_SOAK_SHAPES", 8)))]
    saved = dev.default_score_options()
    try:
        for kernel in ("matrix", "filtered"):
            dev.set_default_score_options(_options(kernel=kernel))
            for batch, n, h in shapes:
                corr = np.stack([scene(max(n, 8), seed=70 + b)[3][:n] for b in range(batch)])
                corr_d = dev.to_device(np.ascontiguousarray(corr))
                outs = []
                for fused in ("1", "0"):
                    monkeypatch.setenv("SFM_LARGE_PASS", fused)
                    monkeypatch.setenv("SFM_SMALL_PASS", fused)    # (one pair of this size would otherwise take the lean small pass)
                    ws = dev.RansacWorkspace(batch, n, h)
                    ws.mask.fill_(9)
                    with_mask = (n + h) % 3 != 0
                    if batch == 1 and fused == "1":
                        dev.ransac_pass_large(corr_d, ws.S, ws.E, ws.flags, ws.cnt, ws.s1, ws.s2, ws.result, ws.mask if with_mask else None,
                                              ws.score_ws, 1.5e-6, 3, AGG_MEAN, philox=(21, 7))
                    else:
                        ws.run(corr_d, 1.5e-6, 3, AGG_MEAN if batch == 1 else AGG_RMS, with_mask=with_mask,
                               philox=(21, 7, 1 if batch == 1 else 5))
                    torch.cuda.synchronize()
                    keys = ("S", "E", "flags", "cnt", "s1", "s2", "result") + (("mask",) if with_mask else ())
                    outs.append({k: getattr(ws, k).cpu().numpy().copy() for k in keys})
                for key in outs[0]:
                    a, b = outs[0][key], outs[1][key]
                    if key == "result" and batch > 1:
                        pass
                    if a.dtype == np.float64:
                        a, b = a.view(np.int64), b.view(np.int64)
                    np.testing.assert_array_equal(a, b, err_msg=f"{kernel} batch {batch} n {n} h {h}: {key}")
    finally:
        dev.set_default_score_options(saved)


@pytest.mark.parametrize("n,h", [(4_000, 87_500), (4_100, 90_000), (7_000, 50_001), (171_000, 2_048), (90_000, 4_000)])
def test_large_pass_at_the_floors_of_the_size_rule(dev, n, h):
    """The size rule of the matrix-pipe kernel (4000 points, 2048 hypotheses, 3.5 x 10^8 evaluations since round 5) at its floors:
    the pass the default route takes there — fused, matrix-pipe kernel, wide waves where the hypotheses are many — against the
    separate calls with the all-fp64 scoring kernel: same samples and E, same counts, same winner and mask, sums to summation order."""
    from structure_from_motion_amd import _native
    from structure_from_motion_amd._native import AGG_RMS

    assert _native.load().sfm_score_kernel_choice(n, h, 1) == 2 and dev.large_pass_eligible(1, n, h)
    _, _, _, corr = scene(n, seed=14)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    thr, min_extra = 1.5e-6, 10
    a, b = dev.RansacWorkspace(1, n, h), dev.RansacWorkspace(1, n, h)
    a.run(corr_d, thr, min_extra, AGG_RMS, philox=(9, 100, 1))
    dev.sample_fit_philox(corr_d, 9, 100, b.S, b.E, b.flags, 1)
    b.cnt, b.s1, b.s2 = dev.score_sed(corr_d, b.E, b.S, thr, exact_only=True)
    dev.select_best(b.cnt, b.s1, b.s2, b.flags, min_extra, AGG_RMS, 0, b.result)
    dev.inlier_mask(corr_d, b.E, b.S, b.result, thr, b.mask)
    for key in ("S", "E", "flags", "cnt", "mask"):
        np.testing.assert_array_equal(getattr(a, key).cpu().numpy(), getattr(b, key).cpu().numpy(), err_msg=key)
    ra, rb = dev.read_select(a.result)[0], dev.read_select(b.result)[0]
    assert (ra.best_h, ra.best_cnt, ra.n_flagged) == (rb.best_h, rb.best_cnt, rb.n_flagged) and ra.best_h >= 0
    for key in ("s1", "s2"):
        x, y = getattr(a, key).cpu().numpy(), getattr(b, key).cpu().numpy()
        both_nan = np.isnan(x) & np.isnan(y)
        np.testing.assert_allclose(x[~both_nan], y[~both_nan], rtol=1e-13, atol=0)


@pytest.mark.parametrize("n,h", [(9_000, 20_000), (20_000, 3_000), (600, 900)])
def test_large_pass_entry_on_sizes_of_the_other_kernels(dev, n, h):
    """sfm_ransac_pass_large takes any size: where sfm_score_sed would not pick the matrix-pipe kernel its own launches run,
    followed by the pass's one selection + mask launch (no ranges to fold) — or, below a few thousand hypotheses, by the
    stand-alone selection and mask.  Outputs equal to the separate calls'."""
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(n, seed=12)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    thr, min_extra = 1.5e-6, 10
    a, b = dev.RansacWorkspace(1, n, h), dev.RansacWorkspace(1, n, h)
    dev.ransac_pass_large(corr_d, a.S, a.E, a.flags, a.cnt, a.s1, a.s2, a.result, a.mask, a.score_ws, thr, min_extra, AGG_RMS,
                          philox=(9, 100))
    dev.sample_fit_philox(corr_d, 9, 100, b.S, b.E, b.flags, 1)
    dev.score_sed(corr_d, b.E, b.S, thr, b.cnt, b.s1, b.s2, workspace=b.score_ws)
    dev.select_best(b.cnt, b.s1, b.s2, b.flags, min_extra, AGG_RMS, 0, b.result)
    dev.inlier_mask(corr_d, b.E, b.S, b.result, thr, b.mask)
    for key in ("S", "E", "flags", "cnt", "result", "mask"):
        np.testing.assert_array_equal(getattr(a, key).cpu().numpy(), getattr(b, key).cpu().numpy(), err_msg=key)
    for key in ("s1", "s2"):
        np.testing.assert_array_equal(getattr(a, key).cpu().numpy().view(np.int64), getattr(b, key).cpu().numpy().view(np.int64))


@pytest.mark.parametrize("n,h", [(20_000, 1_000), (9_000, 2_000), (33_000, 700)])
def test_large_pass_forced_matrix_kernel_without_room_for_the_selection_state(dev, n, h):
    """Round 4's advisor: below ~2064 hypotheses the selection state of the fused pass does not fit its place in the workspace,
    sfm_ransac_pass_large falls back to the stand-alone selection and mask — and with the matrix-pipe kernel forced
    (process-wide options) and the points cut into ranges it still deferred the fold of the ranges to a selection launch that
    never came: the winner was picked from the pre-pass' cost estimates.  Outputs equal to the separate calls', and a model
    is found."""
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(n, seed=12)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    thr, min_extra = 1.5e-6, 10
    saved = dev.default_score_options()
    try:
        dev.set_default_score_options(_options(kernel="matrix"))
        a, b = dev.RansacWorkspace(1, n, h), dev.RansacWorkspace(1, n, h)
        dev.ransac_pass_large(corr_d, a.S, a.E, a.flags, a.cnt, a.s1, a.s2, a.result, a.mask, a.score_ws, thr, min_extra,
                              AGG_RMS, philox=(9, 100))
        dev.sample_fit_philox(corr_d, 9, 100, b.S, b.E, b.flags, 1)
        dev.score_sed(corr_d, b.E, b.S, thr, b.cnt, b.s1, b.s2, workspace=b.score_ws)
        dev.select_best(b.cnt, b.s1, b.s2, b.flags, min_extra, AGG_RMS, 0, b.result)
        dev.inlier_mask(corr_d, b.E, b.S, b.result, thr, b.mask)
        torch.cuda.synchronize()
    finally:
        dev.set_default_score_options(saved)
    for key in ("S", "E", "flags", "cnt", "result", "mask"):
        np.testing.assert_array_equal(getattr(a, key).cpu().numpy(), getattr(b, key).cpu().numpy(), err_msg=key)
    for key in ("s1", "s2"):
        np.testing.assert_array_equal(getattr(a, key).cpu().numpy().view(np.int64), getattr(b, key).cpu().numpy().view(np.int64))
    exact = dev.score_sed(corr_d, a.E, a.S, thr, exact_only=True)
    np.testing.assert_array_equal(exact[0].cpu().numpy(), a.cnt.cpu().numpy())   # counts, not cost estimates
    assert a.outcome(0).best_h >= 0


def test_fused_small_pass_random_sizes(dev, monkeypatch):
    """The lean small pass against the separate calls on 40 random (points, hypotheses) sizes over its whole range — every
    hypotheses-per-wave choice, loop remainder, partial last block, with and without the block barrier: counts, flags, masks
    and the winner identical, sums to summation order."""
    from structure_from_motion_amd._native import AGG_RMS

    rng = np.random.default_rng(2024)
    sizes = [(int(rng.integers(8, 8193)), int(rng.integers(1, 32769))) for _ in range(28)]
    sizes += [(int(rng.integers(4096, 8193)), int(rng.integers(4097, 10239))) for _ in range(12)]   # barrier + one per wave
    for n, h in sizes:
        _, _, _, corr = scene(n, seed=n % 11)
        corr_d = dev.to_device(corr).reshape(1, n, 4)
        outs = []
        for fused in ("1", "0"):
            monkeypatch.setenv("SFM_SMALL_PASS", fused)
            ws = dev.RansacWorkspace(1, n, h)
            ws.run(corr_d, 1.5e-6, 10 if n >= 300 else 0, AGG_RMS, philox=(n + h, 0, 1))
            outs.append({k: getattr(ws, k).cpu().numpy().copy() for k in ("flags", "cnt", "s1", "s2", "result", "mask")})
        fused, plain = outs
        for key in ("flags", "cnt", "mask"):
            np.testing.assert_array_equal(fused[key], plain[key], err_msg=f"{key} at {n} x {h}")
        np.testing.assert_allclose(fused["s1"], plain["s1"], rtol=1e-13, atol=0, equal_nan=True)
        np.testing.assert_allclose(fused["s2"], plain["s2"], rtol=1e-13, atol=0, equal_nan=True)
        assert fused["result"][0][1] == plain["result"][0][1], (n, h)


def test_fused_small_pass_repeated_and_offsets(dev):
    """The arrival counter of the sharded selection is re-armed by every pass (50 passes in a row on one workspace give
    50 correct winners), and h_offset / no-mask (the form a multi-GPU shard uses) behaves like sfm_select_best's."""
    from structure_from_motion_amd._native import AGG_RMS

    n, h = 2000, 7000
    _, _, _, corr = scene(n, seed=3)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    ws = dev.RansacWorkspace(1, n, h)
    got = []
    for seed in range(50):
        ws.run(corr_d, 1.5e-6, 10, AGG_RMS, philox=(seed, 0, 1))
        got.append(int(ws.result[0, 1].cpu()))
    for seed in (0, 17, 49):
        ref = orc.ransac_essential(corr, orc.philox_sample_table(seed, 0, h, n), 1.5e-6, 10, orc.RMS)
        assert got[seed] == ref["best"]
    ws.run(corr_d, 1.5e-6, 10, AGG_RMS, h_offset=123_456, with_mask=False, philox=(49, 0, 1))
    rec = dev.read_select(ws.result)[0]
    assert rec.best_h == got[49] + 123_456 and rec.n_flagged == 0


@pytest.mark.parametrize("n,h", [(300, 2000), (5000, 10000), (4096, 4097), (8192, 1000)])
def test_small_pass_mask_with_hypothesis_offset(dev, n, h):
    """sfm_ransac_pass_small through the C ABI with mask != NULL and h_offset > 0, on both selection paths (one block
    that selects and writes the mask: h, n <= 4096; sharded selection + waiting mask blocks beyond): the record carries
    global indices, the mask is the LOCAL winner's — identical to the pass at offset 0."""
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(n, seed=11)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    ws = dev.RansacWorkspace(1, n, h)
    assert dev.small_pass_eligible(1, n, h)
    out = {}
    for off in (0, 7_000_000_000):
        ws.mask.fill_(0x55)
        dev.ransac_pass_small(corr_d, ws.S, ws.E, ws.flags, ws.cnt, ws.s1, ws.s2, ws.result, ws.mask, ws.score_ws,
                              1.5e-6, 10, AGG_RMS, h_offset=off, philox=(13, 0))
        out[off] = (dev.read_select(ws.result)[0], ws.mask.cpu().numpy()[0].copy())
    (r0, m0), (r1, m1) = out[0], out[7_000_000_000]
    assert r0.best_h >= 0 and r1.best_h == r0.best_h + 7_000_000_000 and r1.best_err == r0.best_err
    np.testing.assert_array_equal(m1, m0)
    assert int((m0 == 2).sum()) == 8 and int((m0 == 1).sum()) == r0.best_cnt
    ref = orc.ransac_essential(corr, orc.philox_sample_table(13, 0, h, n), 1.5e-6, 10, orc.RMS)
    assert ref["best"] == r0.best_h
    np.testing.assert_array_equal(np.nonzero(m0)[0], np.sort(ref["inliers"]))


@pytest.mark.parametrize("n,h", [(300, 2000), (9000, 12000)])
def test_graph_replay_equals_eager(dev, n, h):
    """A captured HIP graph of the whole pass, replayed with the seed rewritten in device memory, gives the
    same winner / E / sample / mask as the eager launch sequence — and as the oracle (both launch paths of
    the scoring kernel: plain order at 300 x 2000, longest-first ordering at 9000 x 12000)."""
    from structure_from_motion_amd import distributed
    from structure_from_motion_amd._native import AGG_RMS

    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr)
    eager = distributed.ShardedRansac(corr_d, h, 1.5e-6, 10, AGG_RMS)
    graphed = distributed.ShardedRansac(corr_d, h, 1.5e-6, 10, AGG_RMS)
    graphed.capture()
    for seed in (5, 6, 2**63 + 11, 7):
        eager.step(seed)
        graphed.step(seed)
        want, got = eager.outcome(), graphed.outcome()
        assert got[0] == want[0] and got[1] == want[1]
        for a, b in zip(got[2:], want[2:]):
            np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(dev.sample_philox_dev(graphed.seed_dev, 3, 50, n).cpu().numpy(),
                                  dev.sample_philox(7, 3, 50, n).cpu().numpy())
    if h <= 2000:
        ref = orc.ransac_essential(corr, orc.philox_sample_table(7, 0, h, n), 1.5e-6, 10, orc.RMS)
        assert ref["best"] == got[0]
        np.testing.assert_array_equal(np.nonzero(got[4])[0], np.sort(ref["inliers"]))


# ------------------------------------------------------------------------------------------------------
# two-tier scoring kernel (fp32 pre-filter + exact fp64) against the all-fp64 kernel
# ------------------------------------------------------------------------------------------------------
def _score_both(dev, corr, E, S, thr, options=None):
    """(all-fp64 kernel, two-tier kernel launched with `options` — sfm_score_sed_ex; None = the library's own rules)."""
    n, h = corr.shape[0], E.shape[0]
    args = (dev.to_device(corr).reshape(1, n, 4), dev.to_device(E.reshape(1, h, 9)),
            dev.to_device(S, torch.int32).reshape(1, h, 8), thr)
    exact = [t.cpu().numpy()[0] for t in dev.score_sed(*args, exact_only=True)]
    filt = [t.cpu().numpy()[0] for t in dev.score_sed(*args, options=options)]
    return exact, filt


def _options(**fields):
    from structure_from_motion_amd._native import ScoreOptions

    return ScoreOptions(**fields)


@pytest.fixture(params=["filtered", "matrix"])
def score_kernel(request):
    """Both two-tier scoring kernels: the VALU filter (score_sed_filtered_kernel) and the matrix-pipe one
    (score_sed_matrix_kernel, csrc/sfm_score_matrix.h), whatever the library's size rule would pick — as launch options
    of the call (sfm_score_options.kernel), not through the process environment."""
    return _options(kernel=request.param)


def _assert_same_scores(exact, filt):
    np.testing.assert_array_equal(filt[0], exact[0])                      # counts: bit-exact
    for a, b in ((filt[1], exact[1]), (filt[2], exact[2])):                # sums: summation order only
        both_nan = np.isnan(a) & np.isnan(b)
        np.testing.assert_allclose(a[~both_nan], b[~both_nan], rtol=1e-13, atol=0)


@pytest.mark.parametrize("n,h", [(300, 64), (4099, 130), (20000, 515)])
@pytest.mark.parametrize("thr", [1.5e-6, 1e-3, 0.0, 1e-12, 1e30, -1.0, float("nan"), float("inf"),
                                 # around the range in which the threshold is folded into the prepared coordinates
                                 1e-31, 9e-31, 1.1e-30, 1e-20, 9e29, 1.1e30, 1e38, 1e300, 5e-324])
def test_filtered_score_equals_exact(dev, score_kernel, n, h, thr):
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(13, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    exact, filt = _score_both(dev, corr, E, S, thr, score_kernel)
    _assert_same_scores(exact, filt)
    if thr == 1.5e-6:
        cnt_o, s1_o, s2_o = orc.score_hypotheses(corr, E, S, thr)
        np.testing.assert_array_equal(filt[0], cnt_o)
        np.testing.assert_allclose(filt[2], s2_o, rtol=1e-13)


@pytest.mark.parametrize("hpw", [1, 2, 4, "matrix"])
def test_filtered_score_every_loop_remainder(dev, hpw):
    """Every hypotheses-per-wave variant of the two-tier kernel (forced with options.hyps_per_wave: small launches would always
    pick one per wave) over point counts that hit each exit of the staged point-load loop — 0, 1, 2, ... full steps of
    128 (the loop is unrolled over 2 or 3 stages), with 0, 1 or 2 tail chunks, full and partial — and hypothesis counts
    that leave slots of the last wave empty.  Counts equal to the all-fp64 kernel's, sums to summation order; a high
    threshold so that the exact tier runs every few steps, a low one so that it almost never does."""
    # "matrix": the matrix-pipe kernel — steps of 32 points, the last one masked; 32 hypothesis slots per wave
    options = _options(kernel="matrix") if hpw == "matrix" else _options(kernel="filtered", hyps_per_wave=hpw)
    for n in (8, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 320, 383, 384, 385, 449, 512, 640, 767, 1000):
        _, _, _, corr = scene(n, seed=n)
        for h in (1, 3, 4, 5, 9):
            S = orc.philox_sample_table(n + h, 0, h, n)
            E, _, _ = orc.fit_hypotheses(corr, S)
            for thr in (1.5e-6, 1e-2):
                exact, filt = _score_both(dev, corr, E, S, thr, options)
                _assert_same_scores(exact, filt)


@pytest.mark.parametrize("scale", [1e-30, 1e-20, 1e-3, 1e3, 1e20, 1e36, 1e40, 1e150, 1e-150, 1e-300])
def test_filtered_score_extreme_matrix_scales(dev, score_kernel, scale):
    """SED is scale-free in E, but the fp32 tier over/underflows: every such pair must fall through to
    the exact tier (same counts as the exact kernel at every scale)."""
    n, h = 2000, 96
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(17, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    exact, filt = _score_both(dev, corr, E * scale, S, 1.5e-6, score_kernel)
    _assert_same_scores(exact, filt)


def test_filtered_score_pixel_units_and_bad_matrices(dev, score_kernel):
    """Un-normalised (pixel) coordinates -> large coordinate maxima; plus NaN / inf / zero matrices."""
    n, h = 3000, 64
    pa, pb, K, _ = scene(n)
    corr = orc.pack_correspondences(pa, pb)
    S = orc.philox_sample_table(19, 0, h, n)
    F, _, _ = orc.fit_hypotheses(corr, S)
    for thr in (1.0, 4.0, 1e-2):
        _assert_same_scores(*_score_both(dev, corr, F, S, thr, score_kernel))
    bad = F.copy()
    bad[3] = np.nan
    bad[4, 1, 1] = np.nan
    bad[5] = np.inf
    bad[6] = 0.0
    bad[7, 0, 0] = np.inf
    bad[8] = 1e-200
    _assert_same_scores(*_score_both(dev, corr, bad, S, 1.0, score_kernel))
    corr_nan = corr.copy()
    corr_nan[17, 2] = np.nan
    corr_nan[99] = np.inf
    corr_nan[500] = 0.0
    _assert_same_scores(*_score_both(dev, corr_nan, F, S, 1.0, score_kernel))


def test_filtered_score_threshold_ties(dev, score_kernel):
    """Thresholds placed exactly on a point's SED (<= is inclusive) and one ulp below."""
    n, h = 1500, 32
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(23, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    sed = orc.sed_values(E, corr)
    rng = np.random.default_rng(0)
    for _ in range(12):
        k, i = rng.integers(0, h), rng.integers(0, n)
        for thr in (sed[k, i], np.nextafter(sed[k, i], 0.0), np.nextafter(sed[k, i], np.inf)):
            exact, filt = _score_both(dev, corr, E, S, float(thr), score_kernel)
            _assert_same_scores(exact, filt)
            cnt_o, _, _ = orc.score_hypotheses(corr, E, S, float(thr))
            np.testing.assert_array_equal(filt[0], cnt_o)


def test_filtered_score_full_size_equals_exact(dev, score_kernel):
    """50k x 20k: the two kernels agree on every count."""
    n, h = 50_000, 20_000
    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    S = dev.sample_philox(5, 0, h, n)
    E, _ = dev.fit_eight_point(corr_d, S)
    exact = dev.score_sed(corr_d, E, S, 1.5e-6, exact_only=True)
    filt = dev.score_sed(corr_d, E, S, 1.5e-6, options=score_kernel)
    assert torch.equal(exact[0], filt[0])
    torch.testing.assert_close(filt[1], exact[1], rtol=1e-13, atol=0, equal_nan=True)
    torch.testing.assert_close(filt[2], exact[2], rtol=1e-13, atol=0, equal_nan=True)


# ---- the matrix-pipe filter's error bound, measured per (point, hypothesis) -------------------------------------------
def _matrix_filter_dump(dev, corr, E, thr):
    """sfm_debug_matrix_filter: the raw tier-1 accumulators r'' [h, n] and d'' [h, n] (upper bound of (dA + dB) / 4 plus the
    slack, scaled) as the three 16-bit matrix instructions produced them, and the per-hypothesis bound record [h, 8]."""
    import ctypes

    from structure_from_motion_amd import _native

    n, h = corr.shape[0], E.shape[0]
    n_pad = (n + 31) // 32 * 32
    corr_d = dev.to_device(corr)
    E_d = dev.to_device(E.reshape(h, 9))
    ws = dev.score_workspace(n, h, 1, corr_d.device, _options(kernel="matrix", split=0))   # the matrix-pipe kernel's tables, no ranges
    r = torch.full((h, n_pad), float("nan"), dtype=torch.float32, device=corr_d.device)
    d = torch.full((h, n_pad), float("nan"), dtype=torch.float32, device=corr_d.device)
    bound = torch.zeros((h, 8), dtype=torch.float32, device=corr_d.device)
    lib = _native.load()
    _native.check(lib.sfm_debug_matrix_filter(corr_d.data_ptr(), n, E_d.data_ptr(), h, float(thr), ws.data_ptr(), ws.numel(),
                                              r.data_ptr(), d.data_ptr(), bound.data_ptr(),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "sfm_debug_matrix_filter")
    torch.cuda.synchronize()
    return r.cpu().numpy()[:, :n].astype(np.float64), d.cpu().numpy()[:, :n].astype(np.float64), bound.cpu().numpy().astype(np.float64)


def _matrix_filter_margin(dev, corr, E, thr):
    """What the bound of csrc/sfm_score_matrix.h promises, checked per evaluation against fp64 recomputation from the true
    operands: (a) |r''_mfma - r''| <= delta'' — returns the worst ratio; (b) the accumulated denominator minus its slack is an
    upper bound of the exact (dA + dB) / 4 (scaled) — returns the smallest margin in units of the rounding allowance; (c) no
    pair with sed <= thr in fp64 is rejected.  Disarmed hypotheses (all-zero operands, infinite slack) reject nothing."""
    r_mfma, d_mfma, bound = _matrix_filter_dump(dev, corr, E, thr)
    delta, slack, sh, armed, sp, rounding = (bound[:, k] for k in range(6))
    kappa = 1.0 / 32.0
    T = thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5)
    c = (1.0 - 1e-6) / np.sqrt(T * (1.0 + kappa))
    xa, ya, xb, yb = (corr[:, k].astype(np.longdouble) for k in range(4))
    one = np.ones_like(xa)
    cl = np.longdouble(c)
    m = np.stack([xb * (xa * cl), xb * (ya * cl), xb * cl, yb * (xa * cl), yb * (ya * cl), yb * cl, xa * cl, ya * cl, one * cl])  # [9, n]
    Ef = E.reshape(len(E), 9).astype(np.longdouble)
    on = armed > 0
    assert on.any()
    r_true = (Ef @ m).astype(np.longdouble) * (sp[:, None] * sh[:, None])
    err = np.abs(r_mfma - r_true.astype(np.float64))[on]
    ratio = float(np.max(err / delta[on, None]))
    # (b) denominators
    e = E.astype(np.longdouble)
    la0 = e[:, 0, 0, None] * xa + e[:, 0, 1, None] * ya + e[:, 0, 2, None]
    la1 = e[:, 1, 0, None] * xa + e[:, 1, 1, None] * ya + e[:, 1, 2, None]
    lb0 = e[:, 0, 0, None] * xb + e[:, 1, 0, None] * yb + e[:, 2, 0, None]
    lb1 = e[:, 0, 1, None] * xb + e[:, 1, 1, None] * yb + e[:, 2, 1, None]
    quarter = ((la0 * la0 + la1 * la1 + lb0 * lb0 + lb1 * lb1) / 4).astype(np.float64) * (sp[:, None] * sh[:, None]) ** 2
    allowance = (rounding * sp * sp)[:, None]   # what the constant slot adds for the bf16 roundings, scaled like d
    margin = ((d_mfma - slack[:, None]) - quarter)[on] / allowance[on]
    # (c) decisions: the kernel rejects iff fma(-r, r, d) < 0 (one rounding: the sign is that of d - r^2)
    rejected = (d_mfma - r_mfma * r_mfma) < 0.0
    sed = orc.sed_values(E, corr)
    with np.errstate(invalid="ignore"):
        inlier = sed <= thr
    lost = int(np.count_nonzero(rejected & inlier))
    assert not rejected[~on].any()
    return ratio, float(np.min(margin)), lost, float(np.mean(rejected[on])), int(np.count_nonzero(inlier))


def _epipolar_scene_wide(n, h, coord, thr, seed):
    """Pixel-like coordinates up to +-coord, h fundamental matrices F = K^-T [t]x R K^-1 of random poses, and the points of
    hypothesis i % h moved off their epipolar line until sed = (0.97 .. 1.03) thr: the advisor's failing case for the
    fp16 split (small terms of a scaled pair land on fp16's subnormal grid)."""
    rng = np.random.default_rng(seed)
    f = coord * 1.2
    K = np.array([[f, 0.0, 0.0], [0.0, f, 0.0], [0.0, 0.0, 1.0]])
    Ki = np.linalg.inv(K)
    F = np.empty((h, 3, 3))
    for k in range(h):
        w = rng.normal(size=3) * 0.15
        th = np.linalg.norm(w)
        W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Fk = Ki.T @ tx @ R @ Ki
        F[k] = Fk / Fk[2, 2] if abs(Fk[2, 2]) > 1e-12 * np.abs(Fk).max() else Fk / np.abs(Fk).max()
    a = rng.uniform(-coord, coord, (n, 2))
    b = rng.uniform(-coord, coord, (n, 2))
    for i in range(n):
        Fk = F[i % h]
        la = Fk @ np.array([a[i, 0], a[i, 1], 1.0])             # line of a in image b
        nrm = np.hypot(la[0], la[1])
        b0 = b[i] - (la[0] * b[i, 0] + la[1] * b[i, 1] + la[2]) / nrm ** 2 * la[:2]   # foot of b on the line
        target = thr * rng.uniform(0.97, 1.03)
        dist = 0.0
        for _ in range(4):
            bb = b0 + dist * la[:2] / nrm
            lb = Fk.T @ np.array([bb[0], bb[1], 1.0])
            dist = np.sqrt(target / (1.0 + nrm ** 2 / (lb[0] ** 2 + lb[1] ** 2)))
        b[i] = b0 + dist * la[:2] / nrm * (1 if i % 2 else -1)
    corr = np.ascontiguousarray(np.column_stack([a, b]))
    return corr, F


def test_matrix_filter_error_bound_margin(dev):
    """not only the outcome of the matrix-pipe filter but the MARGIN of its error bound, per (point,
    hypothesis), on (i) 10^6 pairs of the bench scene, (ii) pixel-unit coordinates, (iii) wide pixel coordinates with points
    within 3 % of the threshold (round 3's advisor: the fp16 split's subnormal tail), (iv) crafted cancellation cases.  On
    every set: |r''_mfma - r''| / delta'' stays under a CEILING of its coordinate regime — about 1.3 x what round 4 measured
    (profiles/r04/filter_margin_report.txt: bench scene 0.036, pixel units 0.29, +-1000 0.32, +-4000 0.47, +-10 000 0.57,
    +-30 000 0.55, crafted 0.06), never above 0.7 — so that a slow drift of the margin (the bound's accumulation term prices
    probed, not specified, truncation behaviour of the matrix unit) fails loudly instead of only being printed —, the
    accumulated denominator is an upper bound of the exact one, and no pair with sed <= thr is rejected.  The worst ratios are
    printed (pytest -s) and recorded in DESIGN.md."""
    report, failures = [], []

    def check(name, corr, E, thr, max_ratio=0.5):
        ratio, margin, lost, rejected, inliers = _matrix_filter_margin(dev, corr, E, thr)
        report.append(f"{name}: worst |error| / delta'' = {ratio:.3f}, least denominator margin = {margin:.3f} allowances, "
                      f"{rejected:.3f} of the evaluations rejected, {inliers} true inliers, {lost} of them rejected")
        if lost != 0 or not ratio <= max_ratio or not margin >= 0.0:
            failures.append(report[-1])

    # (i) the bench scene: 1000 points x 1024 fitted hypotheses
    n, h = 1000, 1024
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(5, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    check("bench scene 1000 x 1024, thr 1.5e-6", corr, E, 1.5e-6, max_ratio=0.06)
    check("bench scene 1000 x 1024, thr 1e-3", corr, E, 1e-3, max_ratio=0.06)
    # (ii) pixel units
    pa, pb, K, _ = scene(3000)
    pix = orc.pack_correspondences(pa, pb)
    Sp = orc.philox_sample_table(19, 0, 256, 3000)
    F, _, _ = orc.fit_hypotheses(pix, Sp)
    for thr in (1.0, 1e-2, 100.0):
        check(f"pixel units 3000 x 256, thr {thr:g}", pix, F, thr, max_ratio=0.4)
    # (iii) wide coordinates, points within 3 % of the threshold
    for coord in (1000.0, 4000.0, 10000.0, 30000.0):
        for thr in (1e-2, 1.0, 100.0):
            cw, Fw = _epipolar_scene_wide(4096, 64, coord, thr, seed=int(coord) + int(thr * 100))
            # (here the absolute term of the subnormal split dominates delta''; the advisor's emulation saw 0.38 .. 0.58 of it used)
            check(f"coordinates +-{coord:g}, 4096 x 64, thr {thr:g}", cw, Fw, thr,
                  max_ratio={1000.0: 0.45, 4000.0: 0.6}.get(coord, 0.7))
    # (iv) crafted: products of alternating sign at the top of the fp16 range; one large + many small addends; mantissas of
    # all ones (worst case of the hi / mid split); entries 2^-20 .. 1 apart (the subnormal tail)
    rng = np.random.default_rng(77)
    n, h = 512, 256
    ones = 2.0 - 2.0 ** -52
    pts = np.empty((n, 4))
    pts[:, :] = rng.choice([1.0, -1.0, ones / 2, -ones / 2, 1.0 - 2.0 ** -12, 1.0 + 2.0 ** -11], size=(n, 4))
    pts[n // 2:] *= 2.0 ** rng.integers(-10, 1, size=(n - n // 2, 4))
    Ec = np.empty((h, 3, 3))
    sign = np.where(np.arange(9) % 2 == 0, 1.0, -1.0).reshape(3, 3)
    Ec[: h // 4] = sign * (1.0 - 2.0 ** -11 - 2.0 ** -22) * (1.0 + 2.0 ** -30 * rng.integers(0, 8, size=(h // 4, 3, 3)))
    Ec[h // 4: h // 2] = sign * 2.0 ** rng.integers(-20, 1, size=(h // 4, 3, 3)).astype(np.float64) * ones / 2
    Ec[h // 2: 3 * h // 4] = rng.choice([1.0, -1.0], size=(h // 4, 3, 3)) * 2.0 ** -12 * rng.uniform(0.5, 1.0, size=(h // 4, 3, 3))
    Ec[h // 2: 3 * h // 4, 0, 0] = 1.0                                   # one large + eight small
    Ec[3 * h // 4:] = rng.normal(size=(h - 3 * h // 4, 3, 3)) * 10.0 ** rng.integers(-6, 1, size=(h - 3 * h // 4, 3, 3))
    for thr in (1e-4, 1.0):
        check(f"crafted cancellation 512 x 256, thr {thr:g}", pts, Ec, thr, max_ratio=0.12)
    print("\n" + "\n".join(report))
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("coord", [4000.0, 10000.0, 30000.0])
@pytest.mark.parametrize("thr", [1e-2, 1.0, 100.0])
def test_matrix_score_wide_coordinates_near_threshold(dev, coord, thr):
    """Un-normalised coordinates of +-4000 .. 30 000 with ~94 points per hypothesis placed within 3 % of the threshold: the
    matrix-pipe kernel's counts equal the all-fp64 kernel's (round 3's advisor found the fp16 split's absolute rounding on
    the subnormal grid missing from the bound: at +-10 000 an emulation lost 4 % of such inliers)."""
    corr, F = _epipolar_scene_wide(6016, 64, coord, thr, seed=int(coord) + 7)
    S = orc.philox_sample_table(3, 0, 64, 6016)
    exact, filt = _score_both(dev, corr, F, S, thr, _options(kernel="matrix"))
    _assert_same_scores(exact, filt)
    assert exact[0].sum() > 1000    # the near-threshold points are there: about half of 6016 lie below the threshold


@pytest.mark.parametrize("split", [0, 2, 3, 4])
@pytest.mark.parametrize("order", [0, 1])
def test_matrix_score_ranges_and_order(dev, split, order):
    """The matrix-pipe kernel with its points cut into 1..4 ranges (partials published per range, added in range order by the
    range that arrives last) and with / without the heaviest-first order: counts equal to the all-fp64 kernel's, the sums to
    summation order — and the same bits when the launch is repeated (a lane's queue is first-in first-out, so the order in
    which a hypothesis' errors are added does not depend on the hypotheses it shares a wave with)."""
    options = _options(kernel="matrix", split=split, order=order)
    n, h = 9000, 2500
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(31, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    for thr in (1.5e-6, 1e-3):
        exact, filt = _score_both(dev, corr, E, S, thr, options)
        _assert_same_scores(exact, filt)
        _, again = _score_both(dev, corr, E, S, thr, options)
        for a, b in zip(filt, again):
            np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                          b.view(np.int64) if b.dtype == np.float64 else b)


@pytest.mark.parametrize("n,h", [(8200, 8191), (8200, 8192), (8200, 8193), (9000, 12345), (8192, 20001), (16000, 40000)])
@pytest.mark.parametrize("split,persistent", [(-1, 0), (3, 0), (8, 0), (8, 1), (16, 1)])
def test_matrix_score_wide_waves(dev, n, h, split, persistent):
    """One pair of 8192 hypotheses and more, in cost order: the entries of the order behind the heaviest classes go in waves of 64
    hypotheses (two operand groups per step, the reject words exchanged between the lane halves; the boundary is the first class
    boundary at or behind entry 4096, written by the sort) — around the threshold, with hypothesis counts that leave ragged last
    waves of both kinds, cut into ranges, with persistent waves: counts equal to the all-fp64 kernel's, sums to summation order,
    the same bits when repeated, and the same counts as the launch without the order (waves of 32 throughout)."""
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(41, 3, h, n)
    E = dev.fit_eight_point(dev.to_device(corr.reshape(1, n, 4)), dev.to_device(S.reshape(1, h, 8), torch.int32))[0].cpu().numpy().reshape(h, 3, 3)
    options = _options(kernel="matrix", split=split, persistent=persistent)
    for thr in (1.5e-6, 2e-4):
        exact, filt = _score_both(dev, corr, E, S, thr, options)
        _assert_same_scores(exact, filt)
        _, again = _score_both(dev, corr, E, S, thr, options)
        for a, b in zip(filt, again):
            np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                          b.view(np.int64) if b.dtype == np.float64 else b)
        _, plain = _score_both(dev, corr, E, S, thr, _options(kernel="matrix", split=split, persistent=persistent, order=0))
        np.testing.assert_array_equal(filt[0], plain[0])


def test_matrix_score_wide_waves_class_boundaries(dev):
    """Where the waves of 64 begin is a class boundary of the cost order: all hypotheses in ONE class (no boundary in
    [4096, 16384]: every wave wide from entry 0), a heavy class of exactly 4096 (boundary at the lower end), one of 16 384 (at the
    upper end) and one of 16 385 (just beyond: all wide), NaN hypotheses (the class without survivors, last in the order) — counts
    equal to the all-fp64 kernel's, sums to summation order, repeated launches bit-identical."""
    n = 8300
    _, _, _, corr = scene(n)
    S_all = orc.philox_sample_table(43, 0, 64, n)
    E_all = dev.fit_eight_point(dev.to_device(corr.reshape(1, n, 4)), dev.to_device(S_all.reshape(1, 64, 8), torch.int32))[0].cpu().numpy().reshape(64, 3, 3)
    sed = np.stack([orc.sed_values(E_all[k], corr) for k in range(64)])
    dense = int(np.argmax((sed <= 1.5e-6).sum(axis=1)))          # a model of the scene: thousands of inliers
    sparse = int(np.argmin((sed <= 1.5e-6).sum(axis=1)))         # a stray one: a handful
    options = _options(kernel="matrix")

    def run(picks):
        E = E_all[picks]
        S = S_all[picks]
        exact, filt = _score_both(dev, corr, E, S, 1.5e-6, options)
        _assert_same_scores(exact, filt)
        _, again = _score_both(dev, corr, E, S, 1.5e-6, options)
        for a, b in zip(filt, again):
            np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a, b.view(np.int64) if b.dtype == np.float64 else b)

    run(np.full(9000, dense))                                                         # one class
    for heavy in (4096, 16384, 16385):
        run(np.concatenate([np.full(heavy, dense), np.full(20000 - heavy, sparse)]))  # two classes, the boundary at `heavy`
    mixed = np.concatenate([np.full(3000, dense), np.arange(64).repeat(150), np.full(2500, sparse)])
    E = E_all[mixed].copy()
    E[::7] = np.nan                                                                   # hypotheses that never count anything
    exact, filt = _score_both(dev, corr, E, S_all[mixed], 1.5e-6, options)
    _assert_same_scores(exact, filt)


@pytest.mark.parametrize("split", [-1, 1, 3, 8, 16])
def test_matrix_score_persistent_waves(dev, split):
    """sfm_score_options.persistent = 1: the grid is what the chip holds and every wave takes (group of 32 hypotheses, range)
    items from a counter — per XCD when the ranges are a multiple of eight, one counter otherwise — until they run out.  Same
    counts as the all-fp64 kernel, and the same BITS as the default launch (one block per four items) with the same ranges: which
    wave scores an item changes nothing about the item."""
    n, h = 9000, 5000
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(31, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    for thr in (1.5e-6, 1e-3):
        exact, persistent = _score_both(dev, corr, E, S, thr, _options(kernel="matrix", split=split, persistent=1))
        _, plain = _score_both(dev, corr, E, S, thr, _options(kernel="matrix", split=split, persistent=0))
        _assert_same_scores(exact, persistent)
        for a, b in zip(persistent, plain):
            np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                          b.view(np.int64) if b.dtype == np.float64 else b)


@pytest.mark.parametrize("n,h", [(32768, 2100), (40011, 4100), (33000, 2050)])
@pytest.mark.parametrize("persistent", [0, 1])
def test_matrix_score_replays_the_cost_prepass(dev, n, h, persistent):
    """Eight ranges over at least 32 768 points: the cost pre-pass scans the first 16 steps of each range and leaves its reject
    words; the scoring waves replay them instead of running tier 1 on those steps again (csrc/sfm_score_matrix.h, MatrixPair::record).
    Forced here at sizes the size rule would give other range counts (the full-size tests run the shape the bench uses): counts
    equal to the all-fp64 kernel's, sums to summation order, and the SAME BITS as the launch with the order switched off, which
    has no pre-pass and computes every step itself."""
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(37, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    for thr in (1.5e-6, 2e-4):
        exact, replayed = _score_both(dev, corr, E, S, thr, _options(kernel="matrix", split=8, order=1, persistent=persistent))
        _assert_same_scores(exact, replayed)
        _, computed = _score_both(dev, corr, E, S, thr, _options(kernel="matrix", split=8, order=0, persistent=persistent))
        for a, b in zip(replayed, computed):
            np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                          b.view(np.int64) if b.dtype == np.float64 else b)


@pytest.mark.parametrize("batch,n,h", [(5, 700, 70), (9, 300, 33), (17, 2100, 40), (3, 4099, 300)])
def test_matrix_score_batches(dev, batch, n, h):
    """The matrix-pipe kernel on a batch of pairs (blocks of a pair share one L2: groups of eight pairs, the last one
    padded; per-pair maxima, operand tables, cost order): every pair's counts equal the all-fp64 kernel's."""
    corr = np.stack([scene(n, seed=40 + b)[3] for b in range(batch)])
    S = np.stack([orc.philox_sample_table(50 + b, 0, h, n) for b in range(batch)])
    E = np.stack([orc.fit_hypotheses(corr[b], S[b])[0] for b in range(batch)])
    args = (dev.to_device(corr), dev.to_device(E.reshape(batch, h, 9)), dev.to_device(S, torch.int32))
    for thr in (1.5e-6, 1e-3):
        exact = [t.cpu().numpy() for t in dev.score_sed(*args, thr, exact_only=True)]
        filt = [t.cpu().numpy() for t in dev.score_sed(*args, thr, options=_options(kernel="matrix"))]
        for b in range(batch):
            _assert_same_scores([x[b] for x in exact], [x[b] for x in filt])


@pytest.mark.parametrize("n,h", [(66_000, 96), (200_000, 64), (65_537, 33)])
def test_matrix_kernel_beyond_65536_points(dev, n, h):
    """Until round 4 a queue entry kept the absolute step of its 32 points in 16 bits and pairs of more than 65 536 points fell
    back to the VALU-filter kernel; the entry now keeps the step relative to its range (ranges of at most 2^16 steps, up to 4 M
    points per pair).  The matrix-pipe kernel at 66 000 and 200 000 points: counts equal to the all-fp64 kernel's, the sums to
    summation order, the same bits when repeated — and it IS the matrix kernel (its sums differ in the last bits from the
    VALU-filter kernel's, which adds in another order)."""
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(3, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    exact, asked = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix"))
    _, again = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix"))
    _, valu = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="filtered"))
    _assert_same_scores(exact, asked)
    _assert_same_scores(exact, valu)
    for a, b in zip(asked, again):
        np.testing.assert_array_equal(a.view(np.int64) if a.dtype == np.float64 else a, b.view(np.int64) if b.dtype == np.float64 else b)
    assert not np.array_equal(asked[1].view(np.int64), valu[1].view(np.int64))
    for split in (2, 16):   # ranges with rebased step indices
        _, ranged = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix", split=split))
        _assert_same_scores(exact, ranged)


def test_matrix_kernel_beyond_two_million_points(dev):
    """More than 2^16 steps of 32 points (2 097 152 points) cannot be one range — a queue entry keeps its step relative to the
    range in 16 bits — so the launcher must cut the points into ranges of at most 65 536 steps whatever was asked for
    (split = 0 included): a relative step of 65 535 and 32-bit byte offsets up to 2^26 are executed here.  Counts equal to the
    all-fp64 kernel's, sums to summation order."""
    n, h = 2_097_152 + 48, 64
    _, _, _, corr = scene(n)
    S = orc.philox_sample_table(3, 0, h, n)
    E, _, _ = orc.fit_hypotheses(corr, S)
    exact, auto = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix"))
    _assert_same_scores(exact, auto)
    _, unsplit = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix", split=0))
    _assert_same_scores(exact, unsplit)
    _, persistent = _score_both(dev, corr, E, S, 1.5e-6, _options(kernel="matrix", split=0, persistent=1))
    _assert_same_scores(exact, persistent)
    assert exact[0].max() > 500_000       # the fitted scene: the best hypotheses keep a large share of the 70 % inliers


def test_score_kernel_size_rule_picks_the_matrix_kernel(dev):
    """Left to itself (options.kernel = auto) a single-pair launch of at least 4000 points, 2048 hypotheses and 3.5 x 10^8
    evaluations runs the matrix-pipe kernel: same counts as the all-fp64 kernel, and as the VALU-filter kernel forced by
    its option."""
    n, h = 8200, 62_000
    _, _, _, corr = scene(n)
    corr_d = dev.to_device(corr).reshape(1, n, 4)
    S = dev.sample_philox(7, 0, h, n)
    E, _ = dev.fit_eight_point(corr_d, S)
    exact = dev.score_sed(corr_d, E, S, 1.5e-6, exact_only=True)
    default = dev.score_sed(corr_d, E, S, 1.5e-6, options=_options())
    valu = dev.score_sed(corr_d, E, S, 1.5e-6, options=_options(kernel="filtered"))
    assert torch.equal(exact[0], default[0]) and torch.equal(exact[0], valu[0])
    torch.testing.assert_close(default[1], exact[1], rtol=1e-13, atol=0, equal_nan=True)
    torch.testing.assert_close(default[2], exact[2], rtol=1e-13, atol=0, equal_nan=True)
    # the two filtered kernels add in different orders: if the default launch had run the VALU kernel, these would be equal bits
    assert not torch.equal(default[1].view(torch.int64), valu[1].view(torch.int64))


# ------------------------------------------------------------------------------------------------------
# batched device-resident pipeline (config C5 shape) against the oracle's composition of the three calls
# ------------------------------------------------------------------------------------------------------
def _oracle_pair(pa, pb, K, seed, h, thr, min_extra):
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
    S = orc.philox_sample_table(seed, 0, h, len(pa))
    ref = orc.ransac_essential(corr, S, thr, min_extra, orc.RMS)
    order = ref["inliers"]
    R, t, mask, votes = orc.recover_r_t(corr[order], ref["E"])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    pts = orc.triangulate_points(pa[order][mask], pb[order][mask], K, T)
    return ref, order, R, t, mask, votes, pts


def test_batched_two_view_pipeline(dev):
    from structure_from_motion_amd import batched
    from structure_from_motion_amd._native import AGG_RMS

    B, n, h, thr, min_extra = 5, 1200, 400, 1.5e-6, 10
    scenes = [orc.synthetic_two_view(n, seed=50 + b, outlier_fraction=0.25) for b in range(B)]
    K = scenes[0][2]
    pix_a = dev.to_device(np.stack([s[0] for s in scenes]))
    pix_b = dev.to_device(np.stack([s[1] for s in scenes]))
    pipe = batched.TwoViewBatch(B, n, h)
    pipe.run(pix_a, pix_b, K, seed=70, thr=thr, min_extra=min_extra, aggregation=AGG_RMS)
    results = pipe.results()
    for b, res in enumerate(results):
        ref, order, R, t, mask, votes, pts = _oracle_pair(scenes[b][0], scenes[b][1], K, 70 + b, h, thr, min_extra)
        assert res.status == batched.OK
        assert res.best_h == ref["best"]
        assert rel(res.E, ref["E"]) <= 1e-6
        np.testing.assert_array_equal(res.inlier_order, order)            # bit-exact index list
        assert sorted(res.votes.tolist()) == sorted(votes)                  # candidate order may differ
        np.testing.assert_allclose(res.R, R, atol=1e-6)
        np.testing.assert_allclose(res.t, t, atol=1e-6)
        np.testing.assert_array_equal(res.pose_mask, mask)
        assert np.max(np.abs(res.points - pts) / np.linalg.norm(pts, axis=1, keepdims=True)) <= 1e-6


def _c5_scenes(B, n):
    scenes = [orc.synthetic_two_view(n, seed=300 + b, outlier_fraction=0.25) for b in range(B)]
    return scenes, scenes[0][2]


def test_c5_batched_pipeline_full_size(dev, tmp_path, c_oracle_lib):
    """BASELINE config 5 at its real shape: 256 pairs x 10 000 correspondences x 2 000 hypotheses, E-estimation +
    cheirality + triangulation on the device in one enqueue.  Exercises the batch-flattened grids and the XCD-aware
    block -> (pair, block) map of the scoring kernel with all 32 groups of eight pairs.
      * every pair: count of the winner == population of its mask, winner == host argmin of the device errors;
      * one pair out of every group of eight, cycling through the residues mod 8, and pair 255: the full oracle comparison of
        test_batched_two_view_pipeline — winner, ordered inlier list, votes, pose mask bit-exact; E, R, t, points
        <= 1e-6;
      * the same batch scored in a child process with SFM_SCORE_XCD=0 (plain (block, pair) grid): byte-identical
        cnt / s1 / s2."""
    import hashlib
    import os
    import subprocess
    import sys

    from structure_from_motion_amd import batched
    from structure_from_motion_amd._native import AGG_RMS

    B, n, h, thr, min_extra, seed = 256, 10_000, 2_000, 1.5e-6, 10, 70
    scenes, K = _c5_scenes(B, n)
    pix_a = dev.to_device(np.stack([s[0] for s in scenes]))
    pix_b = dev.to_device(np.stack([s[1] for s in scenes]))
    pipe = batched.TwoViewBatch(B, n, h)
    pipe.run(pix_a, pix_b, K, seed=seed, thr=thr, min_extra=min_extra, aggregation=AGG_RMS)
    results = pipe.results()
    cnt = pipe.ws.cnt.cpu().numpy()
    s1 = pipe.ws.s1.cpu().numpy()
    s2 = pipe.ws.s2.cpu().numpy()
    mask = pipe.ws.mask.cpu().numpy()
    assert all(r.status == batched.OK for r in results)
    for b, res in enumerate(results):
        err = orc.aggregate(cnt[b], s1[b], s2[b], orc.RMS)
        best, _ = orc.select_best(err, cnt[b], min_extra)
        assert res.best_h == best, b
        assert int((mask[b] == 1).sum()) == cnt[b, best] and int((mask[b] == 2).sum()) == 8, b
        assert len(res.inlier_order) == cnt[b, best] + 8
        assert len(res.points) == len(res.pose_mask) and np.all(np.isfinite(res.points))
    # every hypothesis of every pair: the oracle scores the device's own E -> counts bit-equal (256 x 2 000), sums to
    # summation order; the winner is select_best over the ORACLE's aggregates
    E_all = pipe.ws.E.cpu().numpy()
    S_all = pipe.ws.S.cpu().numpy()
    corr_all = pipe.corr.cpu().numpy()
    for b in range(B):
        cnt_o, s1_o, s2_o = c_oracle_lib.score(corr_all[b], E_all[b], S_all[b], thr)
        np.testing.assert_array_equal(cnt[b], cnt_o, err_msg=str(b))
        np.testing.assert_allclose(s1[b], s1_o, rtol=1e-12, atol=0)
        np.testing.assert_allclose(s2[b], s2_o, rtol=1e-12, atol=0)
        best_o, _ = orc.select_best(orc.aggregate(cnt_o, s1_o, s2_o, orc.RMS), cnt_o, min_extra)
        assert results[b].best_h == best_o, b
    checked = sorted({8 * g + (3 * g) % 8 for g in range(32)} | {B - 1})   # one pair of every group of eight
    assert {b % 8 for b in checked} == set(range(8)) and {b // 8 for b in checked} == set(range(32))
    ties = 0
    for b in checked:
        ref, order, R, t, pmask, votes, pts = _oracle_pair(scenes[b][0], scenes[b][1], K, seed + b, h, thr, min_extra)
        res = results[b]
        assert res.best_h == ref["best"], b
        assert rel(res.E, ref["E"]) <= 1e-6
        np.testing.assert_array_equal(res.inlier_order, order)
        assert sorted(res.votes.tolist()) == sorted(votes)
        np.testing.assert_array_equal(corr_all[b], orc.pack_correspondences(
            orc.to_normalized_image_coords(scenes[b][0], K), orc.to_normalized_image_coords(scenes[b][1], K)))
        if sorted(votes)[-1] == sorted(votes)[-2]:
            # two candidate poses tie for the most votes (the winner of the lowest-error rule can be a model with
            # a dozen inliers): the reference then takes the first in ITS candidate order, which depends on LAPACK's
            # sign choices for the singular vectors of E — not a property of the data.  The device's
            # pose must then be ONE of the oracle's tied candidates, and mask / points are compared for that candidate
            ties += 1
            R1, R2, t1 = orc.recover_all_r_t(ref["E"])
            cands = [(Rc, tc) for Rc in (R1, R2) for tc in (t1, -t1)]
            tied = [c for c, v in zip(cands, votes) if v == max(votes)]
            match = [c for c in tied if np.allclose(res.R, c[0], atol=1e-6) and np.allclose(res.t, c[1], atol=1e-6)]
            assert len(match) == 1, (b, votes)
            R, t = match[0]
            pmask = np.nonzero(orc.cheirality_pass(corr_all[b][order], R, t, None))[0]
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R, t
            pts = orc.triangulate_points(scenes[b][0][order][pmask], scenes[b][1][order][pmask], K, T)
        np.testing.assert_allclose(res.R, R, atol=1e-6)
        np.testing.assert_allclose(res.t, t, atol=1e-6)
        np.testing.assert_array_equal(res.pose_mask, pmask)
        assert np.max(np.abs(res.points - pts) / np.linalg.norm(pts, axis=1, keepdims=True)) <= 1e-6
        # every hypothesis of the pair against the reference route (oracle fit + oracle score).  The device fits E
        # itself (<= 1e-6 from the oracle's, typically 1e-13): a point within that margin of the threshold may be decided
        # differently for an ill-conditioned sample — every such disagreement is checked point by point
        ok = ~ref["degenerate"]
        fit_err = assert_fits_agree(corr_all[b], S_all[b], E_all[b], ref["Eall"], ok=ok, mp_budget=4)
        assert_counts_explained(corr_all[b], S_all[b], thr, E_all[b].reshape(-1, 3, 3), ref["Eall"], cnt[b], ref["cnt"],
                                fit_err)
        same = cnt[b] == ref["cnt"]
        assert np.median(np.abs(s2[b][same] - ref["s2"][same]) / np.maximum(ref["s2"][same], 1e-300)) <= 1e-9
    assert ties <= len(checked) // 4
    digest = hashlib.sha256(cnt.tobytes() + s1.tobytes() + s2.tobytes()).hexdigest()
    del pipe
    torch.cuda.empty_cache()
    out_file = tmp_path / "digest.txt"
    code = (
        "import hashlib, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import sfm_oracle as orc\n"
        "from structure_from_motion_amd import batched, device as dev\n"
        "from structure_from_motion_amd._native import AGG_RMS\n"
        "B, n, h = 256, 10000, 2000\n"
        "scenes = [orc.synthetic_two_view(n, seed=300 + b, outlier_fraction=0.25) for b in range(B)]\n"
        "pipe = batched.TwoViewBatch(B, n, h)\n"
        "pipe.run(dev.to_device(np.stack([s[0] for s in scenes])), dev.to_device(np.stack([s[1] for s in scenes])),\n"
        "         scenes[0][2], seed=70, thr=1.5e-6, min_extra=10, aggregation=AGG_RMS)\n"
        "cnt, s1, s2 = (t.cpu().numpy() for t in (pipe.ws.cnt, pipe.ws.s1, pipe.ws.s2))\n"
        "open(%r, 'w').write(hashlib.sha256(cnt.tobytes() + s1.tobytes() + s2.tobytes()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(out_file))
    env = dict(os.environ, SFM_SCORE_XCD="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out_file.read_text() == digest   # the block -> pair map only moves work between XCDs


def test_batched_pipeline_with_local_optimisation(dev):
    """Extension: winners refitted on all inliers (two rounds) before pose recovery — per pair identical to the
    oracle's RANSAC -> local optimisation -> pose -> triangulation, inlier lists in index order."""
    from structure_from_motion_amd import batched
    from structure_from_motion_amd._native import AGG_RMS

    B, n, h, thr, min_extra = 4, 1500, 300, 1.5e-6, 150
    scenes = [orc.synthetic_two_view(n, seed=90 + b, outlier_fraction=0.25) for b in range(B)]
    K = scenes[0][2]
    pipe = batched.TwoViewBatch(B, n, h)
    pipe.run(dev.to_device(np.stack([s[0] for s in scenes])), dev.to_device(np.stack([s[1] for s in scenes])), K,
             seed=70, thr=thr, min_extra=min_extra, aggregation=AGG_RMS, local_optimisation=2)
    gained = 0
    for b, res in enumerate(pipe.results()):
        pa, pb = scenes[b][0], scenes[b][1]
        corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
        ref = orc.ransac_essential(corr, orc.philox_sample_table(70 + b, 0, h, n), thr, min_extra, orc.RMS)
        assert res.status == batched.OK and res.best_h == ref["best"]
        mask0 = np.zeros(n, dtype=bool)
        mask0[ref["inliers"]] = True
        E_o, m_o, cnt_o, _, acc_o = orc.local_optimisation(corr, ref["E"], mask0, ref["err"], thr, orc.RMS, 2)
        gained += cnt_o - mask0.sum()
        order = np.nonzero(m_o)[0]
        np.testing.assert_array_equal(res.inlier_order, order)
        assert rel(res.E, E_o) <= 1e-9
        R, t, mask, votes = orc.recover_r_t(corr[order], E_o)
        assert sorted(res.votes.tolist()) == sorted(votes)
        np.testing.assert_allclose(res.R, R, atol=1e-6)
        np.testing.assert_allclose(res.t, t, atol=1e-6)
        np.testing.assert_array_equal(res.pose_mask, mask)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        pts = orc.triangulate_points(pa[order][mask], pb[order][mask], K, T)
        assert np.max(np.abs(res.points - pts) / np.linalg.norm(pts, axis=1, keepdims=True)) <= 1e-6
    assert gained > 0


def test_batched_pipeline_status_codes(dev):
    from structure_from_motion_amd import batched
    from structure_from_motion_amd._native import AGG_RMS

    B, n, h = 3, 400, 100
    scenes = [orc.synthetic_two_view(n, seed=80 + b) for b in range(B)]
    K = scenes[0][2]
    pa = np.stack([s[0] for s in scenes])
    pb = np.stack([s[1] for s in scenes])
    pb[1] = np.random.default_rng(0).uniform(0, 500, (n, 2))  # pair 1: pure noise -> no model at min_extra=150
    pipe = batched.TwoViewBatch(B, n, h)
    pipe.run(dev.to_device(pa), dev.to_device(pb), K, seed=3, thr=1.5e-6, min_extra=150, aggregation=AGG_RMS)
    res = pipe.results()
    assert res[1].status == batched.NO_MODEL
    assert res[0].status in (batched.OK, batched.NO_MODEL)


def test_filtered_score_randomized_sweep(dev, score_kernel):
    """200 random (matrix family, scale, threshold, size) combinations: the two-tier kernel's counts equal the
    all-fp64 kernel's bit for bit, sums to summation-order accuracy."""
    rng = np.random.default_rng(2024)
    _, _, _, corr_full = scene(6000)
    total_checked = 0
    for trial in range(200):
        n = int(rng.integers(8, 6000))
        h = int(rng.integers(1, 70))
        corr = corr_full[rng.permutation(len(corr_full))[:n]].copy()
        family = trial % 5
        if family == 0:      # fitted hypotheses (the realistic case)
            S = orc.philox_sample_table(trial, 0, h, n)
            E, _, _ = orc.fit_hypotheses(corr, S)
        else:
            S = orc.philox_sample_table(trial, 0, h, n)
            E = rng.normal(size=(h, 3, 3))
            if family == 1:  # random dense matrices
                pass
            elif family == 2:  # rank-1-ish / tiny third column: loose bounds, heavy cancellation
                E[:, :, 2] *= 1e-6
            elif family == 3:  # huge dynamic range between entries
                E *= 10.0 ** rng.integers(-8, 9, size=(h, 3, 3))
            else:            # nearly singular rows
                E[:, 1] = E[:, 0] * (1.0 + 1e-9 * rng.normal(size=(h, 1)))
            E[:, 2, 2] = 1.0
        E = E * 10.0 ** float(rng.integers(-40, 41))
        if trial % 7 == 0:
            corr[:, :] *= 50.0  # far outside the normalised range (larger coordinate maxima)
        thr = float(10.0 ** rng.uniform(-12, 2)) if trial % 11 else 0.0
        exact, filt = _score_both(dev, corr, E, S, thr, score_kernel)
        _assert_same_scores(exact, filt)
        total_checked += n * h
    assert total_checked > 5e6


def test_two_sided_filter_variant(dev):
    """options.one_sided = 0 selects the two-sided tier-1 test of the VALU filter everywhere: same counts as the all-fp64
    kernel (an option of the call since round 4; SFM_SCORE_ONE_SIDED=0 sets it as the process default)."""
    from structure_from_motion_amd import synthetic

    n, h = 9000, 700
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
    S = dev.sample_philox(5, 0, h, n)
    E, _ = dev.fit_eight_point(corr, S)
    for thr in (1.5e-6, 1e-9, 0.0, 3e-4):
        a = dev.score_sed(corr, E, S, thr, exact_only=True)
        b = dev.score_sed(corr, E, S, thr, options=_options(kernel="filtered", one_sided=0))
        assert torch.equal(a[0], b[0]), thr
        assert torch.allclose(a[1], b[1], rtol=1e-12, atol=0, equal_nan=True)


@pytest.mark.parametrize("order", [1, 0])
def test_score_hypothesis_ordering_does_not_change_results(dev, order):
    """The longest-first processing order (forced on / off) only affects speed: identical outputs on odd sizes,
    tiny hypothesis counts and batches."""
    options = _options(order=order)
    rng = np.random.default_rng(3)
    for n, h in [(64, 1), (100, 3), (999, 5), (2000, 4), (4099, 130), (300, 257)]:
        _, _, _, corr = scene(n)
        S = orc.philox_sample_table(31, 0, h, n)
        E, _, _ = orc.fit_hypotheses(corr, S)
        exact, filt = _score_both(dev, corr, E, S, 1.5e-6, options)
        _assert_same_scores(exact, filt)
        cnt_o, _, s2_o = orc.score_hypotheses(corr, E, S, 1.5e-6)
        np.testing.assert_array_equal(filt[0], cnt_o)
    # batched: every pair gets its own order
    B, n, h = 3, 1500, 37
    corr_all = np.stack([scene(n, seed=60 + b)[3] for b in range(B)])
    S_all = np.stack([orc.philox_sample_table(70 + b, 0, h, n) for b in range(B)])
    E_all = np.stack([orc.fit_hypotheses(corr_all[b], S_all[b])[0] for b in range(B)])
    cnt, s1, s2 = dev.score_sed(dev.to_device(corr_all), dev.to_device(E_all.reshape(B, h, 9)),
                                dev.to_device(S_all, torch.int32), 1.5e-6, options=options)
    for b in range(B):
        cnt_o, s1_o, s2_o = orc.score_hypotheses(corr_all[b], E_all[b], S_all[b], 1.5e-6)
        np.testing.assert_array_equal(cnt.cpu().numpy()[b], cnt_o)
        np.testing.assert_allclose(s2.cpu().numpy()[b], s2_o, rtol=1e-13)


# ------------------------------------------------------------------------------------------------------
# local optimisation (extension): refit on all inliers
# ------------------------------------------------------------------------------------------------------
def _refine(dev, corr, E, mask, err, thr, agg, rounds):
    E_d = dev.to_device(np.asarray(E, dtype=np.float64).reshape(-1, 9))
    m_d = dev.to_device(np.asarray(mask).astype(np.uint8).reshape(E_d.shape[0], -1), torch.uint8)
    e_d = dev.to_device(np.asarray(err, dtype=np.float64).reshape(-1))
    c_d = dev.to_device(corr).reshape(E_d.shape[0], -1, 4)
    E_out, m_out, info = dev.refine_inliers(c_d, E_d, m_d, e_d, thr, agg, rounds)
    return E_out.cpu().numpy().reshape(-1, 3, 3), m_out.cpu().numpy(), dev.read_refine_info(info)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_refit_golden_reference_helpers(dev, golden, case):
    """One forced refit (threshold = inf accepts it) on the inlier subset of G13 == the N-point fit composed
    from the reference's own helpers."""
    d = golden("g13_refit")
    K = d[f"{case}_K"]
    corr = orc.pack_correspondences(orc.to_normalized_image_coords(d[f"{case}_pix_a"], K),
                                    orc.to_normalized_image_coords(d[f"{case}_pix_b"], K))
    n = corr.shape[0]
    mask = np.zeros(n, dtype=np.uint8)
    mask[d[f"{case}_idx"]] = 2  # any non-zero value marks an inlier
    E, m, info = _refine(dev, corr, np.eye(3), mask, [np.inf], np.inf, 0, 1)
    want = d[f"{case}_E"]
    assert np.max(np.abs(E[0] - want)) / np.max(np.abs(want)) <= 1e-9
    assert info[0][1] == n and info[0][2] == 1 and m.sum() == n


@pytest.mark.parametrize("agg,name", [(0, orc.SUM), (1, orc.SQUARE), (2, orc.MEAN), (3, orc.RMS)])
def test_local_optimisation_matches_oracle(dev, agg, name):
    n, thr = 3000, 1.5e-6
    pa, pb, K, corr = scene(n, seed=9)
    S = orc.philox_sample_table(5, 0, 400, n)
    ref = orc.ransac_essential(corr, S, thr, 300, name)
    assert ref["best"] >= 0
    mask = np.zeros(n, dtype=np.uint8)
    mask[ref["inliers"]] = 1
    mask[S[ref["best"]]] = 2
    for rounds in (0, 1, 2, 6):
        E_o, m_o, cnt_o, err_o, acc_o = orc.local_optimisation(corr, ref["E"], mask, ref["err"], thr, name, rounds)
        E, m, info = _refine(dev, corr, ref["E"], mask, [ref["err"]], thr, agg, rounds)
        np.testing.assert_array_equal(m[0] != 0, m_o)  # inlier index set: bit-exact
        assert info[0][1] == cnt_o and info[0][2] == acc_o
        assert abs(info[0][0] - err_o) <= 1e-12 * abs(err_o)
        assert np.max(np.abs(E[0] - E_o)) / np.max(np.abs(E_o)) <= 1e-9
    assert acc_o >= 1  # the scene is one where refitting helps


def test_local_optimisation_random_scenes(dev):
    """16 scenes (noise 0 .. 1.5 px, 10 .. 60 % outliers, 600 .. 6000 points): the refit's inverse-iteration eigenvector
    + inertia-count predicate against the oracle's LAPACK eig — inlier sets bit-exact, E to 1e-9, through 3 rounds."""
    rng = np.random.default_rng(77)
    thr = 1.5e-6
    improved = 0
    for trial in range(16):
        n = int(rng.integers(600, 6000))
        noise = float(rng.choice([0.0, 0.1, 0.5, 1.0, 1.5]))
        pa, pb, K, R, t, is_out = orc.synthetic_two_view(n, seed=100 + trial, outlier_fraction=float(rng.uniform(0.1, 0.6)),
                                                        noise_px=noise)
        corr = orc.pack_correspondences(orc.to_normalized_image_coords(pa, K), orc.to_normalized_image_coords(pb, K))
        S = orc.philox_sample_table(trial, 0, 300, n)
        ref = orc.ransac_essential(corr, S, thr, max(20, n // 20), orc.RMS)
        if ref["best"] < 0:
            continue
        mask = np.zeros(n, dtype=np.uint8)
        mask[ref["inliers"]] = 1
        E_o, m_o, cnt_o, err_o, acc_o = orc.local_optimisation(corr, ref["E"], mask, ref["err"], thr, orc.RMS, 3)
        E, m, info = _refine(dev, corr, ref["E"], mask, [ref["err"]], thr, 3, 3)
        np.testing.assert_array_equal(m[0] != 0, m_o, err_msg=f"trial {trial}")
        assert info[0][1] == cnt_o and info[0][2] == acc_o
        assert np.max(np.abs(E[0] - E_o)) / np.max(np.abs(E_o)) <= 1e-9, trial
        improved += acc_o > 0
    assert improved >= 8


def test_local_optimisation_batch_and_edge_cases(dev):
    """Three pairs in one launch: a normal one, one whose RANSAC found nothing (empty mask), one with a
    degenerate inlier set (identical points) — each handled independently, untouched where nothing applies."""
    n, thr = 800, 1.5e-6
    _, _, _, c0 = scene(n, seed=3)
    S = orc.philox_sample_table(5, 0, 300, n)
    ref = orc.ransac_essential(c0, S, thr, 100, orc.RMS)
    m0 = np.zeros(n, dtype=np.uint8)
    m0[ref["inliers"]] = 1
    flat = np.repeat(c0[:1], n, axis=0)
    corr = np.stack([c0, c0, flat])
    E_in = np.stack([ref["E"], np.full((3, 3), 7.0), ref["E"]])
    masks = np.stack([m0, np.zeros(n, dtype=np.uint8), np.ones(n, dtype=np.uint8)])
    errs = [ref["err"], np.inf, 1.0]
    E, m, info = _refine(dev, corr, E_in, masks, errs, thr, 3, 4)
    E_o, m_o, cnt_o, err_o, acc_o = orc.local_optimisation(c0, ref["E"], m0, ref["err"], thr, orc.RMS, 4)
    np.testing.assert_array_equal(m[0] != 0, m_o)
    assert info[0][1:] == (cnt_o, acc_o)
    np.testing.assert_array_equal(E[1], E_in[1])
    assert info[1] == (np.inf, 0, 0) and not m[1].any()
    np.testing.assert_array_equal(E[2], E_in[2])
    assert info[2][1:] == (n, 0) and m[2].all()
