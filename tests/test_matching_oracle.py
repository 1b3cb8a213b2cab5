"""CPU tests: the matcher oracle (oracle/match_oracle.py) against the real reference's outputs
(tests/golden/g11_matching.npz) and the reference's own unit vectors (test_ncc.py, test_ssd.py,
test_util.py, test_matching.py)."""
import numpy as np
import pytest

from oracle import match_oracle as mo

COMBOS = {"none": None, "ratio": {mo.RATIO_TEST}, "cross": {mo.CROSSCHECK}, "both": {mo.RATIO_TEST, mo.CROSSCHECK}}
METRICS = {"ncc": (mo.ncc_scores, 9), "ncc5": (mo.ncc_scores, 5), "ssd": (mo.ssd_scores, 5)}


@pytest.mark.parametrize("metric", list(METRICS))
def test_scores_match_reference(golden, metric):
    d = golden("g11_matching")
    fn, ws = METRICS[metric]
    ia, ib = (d["image_a"].astype(np.float64), d["image_b"].astype(np.float64)) if metric == "ssd" else (d["image_a"], d["image_b"])
    got = fn(ia, ib, d["feats_a"], d["feats_b"], ws)
    ref = d[f"scores_{metric}"]
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-13, atol=2e-15)
    assert (got == 2.0).sum() == (ref == 2.0).sum() if metric != "ssd" else True


@pytest.mark.parametrize("metric", list(METRICS))
def test_heap_closed_form_equals_heapq(golden, metric):
    ref = golden("g11_matching")[f"scores_{metric}"]
    best, arg, second = mo.row_summary(ref)
    literal = np.array([mo.heap_top_two(r) for r in ref])
    np.testing.assert_array_equal(literal[:, 0], best)
    np.testing.assert_array_equal(literal[:, 1], arg)
    np.testing.assert_array_equal(literal[:, 2], second)
    # heap[1] is NOT always the second smallest: the quirk must be visible on this fixture
    true_second = np.sort(ref, axis=1)[:, 1]
    assert np.any(second != true_second)


def test_heap_closed_form_random_rows():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 33, 200):
        rows = rng.integers(0, 12, size=(50, n)).astype(np.float64)  # many ties
        best, arg, second = mo.row_summary(rows)
        for r in range(len(rows)):
            b, a, s = mo.heap_top_two(rows[r])
            assert (b, a) == (best[r], arg[r])
            assert (np.isnan(s) and np.isnan(second[r])) or s == second[r]


@pytest.mark.parametrize("metric", list(METRICS))
@pytest.mark.parametrize("combo", list(COMBOS))
@pytest.mark.parametrize("thr", [0.7, 0.95])
def test_match_lists_equal_reference(golden, metric, combo, thr):
    d = golden("g11_matching")
    fn, ws = METRICS[metric]
    ia, ib = (d["image_a"].astype(np.float64), d["image_b"].astype(np.float64)) if metric == "ssd" else (d["image_a"], d["image_b"])
    want = d[f"matches_{metric}_{combo}_{thr}"]
    on_ref = np.array(mo.match_brute_force(d[f"scores_{metric}"], COMBOS[combo], thr), dtype=np.float64).reshape(-1, 3)
    np.testing.assert_array_equal(on_ref, want)
    own = np.array(mo.match_brute_force(fn(ia, ib, d["feats_a"], d["feats_b"], ws), COMBOS[combo], thr),
                   dtype=np.float64).reshape(-1, 3)
    np.testing.assert_array_equal(own[:, :2], want[:, :2])
    np.testing.assert_allclose(own[:, 2], want[:, 2], rtol=1e-12, atol=2e-15)


def test_reference_unit_vectors(golden):
    d = golden("g11_matching")
    img = np.array([[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [9, 8, 7, 6, 5], [4, 3, 2, 1, 0], [1, 2, 3, 4, 5]])
    c = np.array([[2.0, 2.0]])
    np.testing.assert_allclose(mo.ncc_scores(img, img.copy(), c, c, 5)[0, 0], 0.0, atol=1e-10)   # test_ncc.py
    np.testing.assert_allclose(mo.ncc_scores(img, -img, c, c, 5)[0, 0], 2.0)
    for ia, ib, s in zip(d["unit_ncc_random_a"], d["unit_ncc_random_b"], d["unit_ncc_random_scores"]):
        np.testing.assert_allclose(mo.ncc_scores(ia, ib, c, c, 5)[0, 0], s, rtol=1e-9, atol=1e-12)
    a = np.zeros((6, 6)); a[:3, :3] = np.arange(1, 10).reshape(3, 3)
    b = np.zeros((6, 6)); b[3:, 3:] = np.arange(9, 0, -1).reshape(3, 3)
    sq = lambda x: x ** 2
    expect = (sq(8) + sq(6) + sq(4) + sq(2) + sq(0) + sq(2) + sq(4) + sq(6) + sq(8)) / 9           # test_ssd.py
    assert mo.ssd_scores(a, b, np.array([[1.0, 1.0]]), np.array([[4.0, 4.0]]), 3)[0, 0] == expect
    assert mo.ssd_scores(a, b, np.array([[0.0, 0.0]]), np.array([[4.0, 4.0]]), 3)[0, 0] == np.inf
    assert mo.ssd_scores(a, b, np.array([[1.0, 1.0]]), np.array([[5.0, 5.0]]), 3)[0, 0] == np.inf
    f = np.array([[2, 2], [1, 2], [2, 1], [197, 97], [197, 98], [198, 97]], dtype=float)           # test_util.py (x, y)
    f[1] = [2, 1]; f[2] = [1, 2]
    np.testing.assert_array_equal(mo.within_bounds(f, (100, 200), 5), [True, False, False, True, False, False])
    with pytest.raises(ValueError):
        mo.ncc_scores(np.zeros((4, 4)), np.zeros((4, 5)), c, c, 3)


def test_reference_matching_unit_vectors():
    """test_matching.py: mock score tables."""
    table = np.array([[10, 20, 30, 7], [30, 9, 20, 15], [20, 30, 8, 31]], dtype=float)
    assert mo.match_brute_force(table) == [(0, 3, 7.0), (1, 1, 9.0), (2, 2, 8.0)]
    ratio = np.array([[10, 5, 20], [10, 6, 7]], dtype=float)
    assert mo.match_brute_force(ratio, {mo.RATIO_TEST}, 0.5) == [(0, 1, 5.0)]
    with pytest.raises(IndexError):
        mo.match_brute_force(np.zeros((2, 0)))
    assert mo.match_brute_force(np.zeros((2, 0)), {mo.RATIO_TEST}) == []
    single = np.array([[3.0], [4.0]])
    assert mo.match_brute_force(single, {mo.RATIO_TEST}, 0.1) == [(0, 0, 3.0), (1, 0, 4.0)]
    assert mo.match_brute_force(single, {mo.CROSSCHECK}) == [(0, 0, 3.0)]


INT_DTYPES = ("uint8", "int8", "uint16", "int16", "uint32", "int32", "uint64", "int64")


@pytest.mark.parametrize("name", INT_DTYPES)
def test_ssd_integer_dtypes_equal_reference_bit_for_bit(golden, name):
    """ssd.py:31-36 in the image dtype (G14: the real reference on images over each dtype's full range): both restatements —
    exact Python integers with explicit reductions, and NumPy's own fixed-width arithmetic — reproduce every score exactly."""
    d = golden("g14_ssd_integer")
    g = d["grid_feats"]
    for ws in (3, 5):
        ref = d[f"{name}_scores_w{ws}"]
        np.testing.assert_array_equal(mo.ssd_scores(d[f"{name}_a"], d[f"{name}_b"], g, g[::3], ws), ref)
        np.testing.assert_array_equal(mo.ssd_scores_in_dtype(d[f"{name}_a"], d[f"{name}_b"], g, g[::3], ws), ref)
    assert np.isinf(d[f"{name}_scores_w5"]).any() and np.isfinite(d[f"{name}_scores_w5"]).any()


def test_ssd_uint8_scores_and_match_lists_equal_reference(golden):
    d = golden("g14_ssd_integer")
    g11 = golden("g11_matching")
    for ws in (5, 9):
        got = mo.ssd_scores(d["image_a"], d["image_b"], d["feats_a"], d["feats_b"], ws)
        np.testing.assert_array_equal(got, d[f"scores_u8_w{ws}"])
        np.testing.assert_array_equal(mo.ssd_scores_in_dtype(d["image_a"], d["image_b"], d["feats_a"], d["feats_b"], ws), got)
        for combo, strat in COMBOS.items():
            for thr in (0.7, 0.95):
                own = np.array(mo.match_brute_force(got, strat, thr), dtype=np.float64).reshape(-1, 3)
                np.testing.assert_array_equal(own, d[f"matches_u8_w{ws}_{combo}_{thr}"])
    # the wrap is visible on this fixture: the uint8 scores are NOT the float-image scores of G11
    assert not np.array_equal(d["scores_u8_w5"], g11["scores_ssd"])
    assert np.nanmax(d["scores_u8_w5"][np.isfinite(d["scores_u8_w5"])]) <= 255.0


def test_ssd_mixed_dtypes_and_bool(golden):
    d = golden("g14_ssd_integer")
    g = d["grid_feats"]
    for na, nb in (("uint8", "int16"), ("uint8", "int8"), ("uint32", "int32")):   # NumPy promotes to int16, int16, int64
        np.testing.assert_array_equal(mo.ssd_scores(d[f"{na}_a"], d[f"{nb}_b"], g, g[::3], 3), d[f"mixed_{na}_{nb}_scores"])
    assert mo.ssd_result_kind(np.uint8, np.int16) == ("int", 16, True)
    assert mo.ssd_result_kind(np.uint8, np.int8) == ("int", 16, True)
    assert mo.ssd_result_kind(np.uint32, np.int32) == ("int", 64, True)
    assert mo.ssd_result_kind(np.uint64, np.int64) == ("float",)     # NumPy goes to float64 here
    for na, nb, key_b in (("uint64", "int64", "int64_b"), ("uint8", "float64", "mixed_uint8_float64_b")):
        got = mo.ssd_scores(d[f"{na}_a"], d[key_b], g, g[::3], 3)
        ref = d[f"mixed_{na}_{nb}_scores"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-14)
    assert int(d["bool_raises"]) == 1
    with pytest.raises(TypeError):
        mo.ssd_scores(d["uint8_a"] > 4, d["uint8_b"] > 5, g, g, 3)
    img = d["unit_int_image"]
    c = np.array([[2.0, 2.0]])
    assert mo.ssd_scores(img, img.copy(), c, c, 5)[0, 0] == float(d["unit_int_same"]) == 0.0
    assert mo.ssd_scores(img, -img, c, c, 5)[0, 0] == float(d["unit_int_negated"])
