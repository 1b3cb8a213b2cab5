"""CPU tests of the matcher's host side: the generic (arbitrary callable) route against the reference's own
test_matching.py cases, and recognition of image-pair score functions."""
import functools

import numpy as np
import pytest

from lib.common.feature import Feature
from lib.feature_matching import matching, util
from structure_from_motion_amd.feature_matching import ncc, ssd
from structure_from_motion_amd.feature_matching.matching import ImagePairScore, _device_score_spec


def _table_function(features_a, features_b, table):
    def score(feature_a, feature_b):
        return table[features_a.index(feature_a)][features_b.index(feature_b)]
    return score


def test_match_features_no_validation():
    """reference test_matching.py:8-35."""
    fa = [Feature(1, 1), Feature(2, 2), Feature(3, 3)]
    fb = [Feature(4, 4), Feature(5, 5), Feature(6, 6), Feature(7, 7)]
    table = {0: {0: 10, 1: 20, 2: 30, 3: 7}, 1: {0: 30, 1: 9, 2: 20, 3: 15}, 2: {0: 20, 1: 30, 2: 8, 3: 31}}
    matches = matching.match_brute_force(fa, fb, _table_function(fa, fb, table))
    assert matches == [matching.Match(0, 3, 7), matching.Match(1, 1, 9), matching.Match(2, 2, 8)]


def test_match_features_ratio_test():
    """reference test_matching.py:37-58: a ratio of exactly the threshold passes."""
    fa = [Feature(1, 1), Feature(2, 2)]
    fb = [Feature(3, 3), Feature(4, 4), Feature(5, 5)]
    table = {0: {0: 10, 1: 5, 2: 20}, 1: {0: 10, 1: 6, 2: 7}}
    matches = matching.match_brute_force(
        fa, fb, _table_function(fa, fb, table),
        validation_strategies=matching.ValidationStrategy.RATIO_TEST, ratio_test_threshold=0.5)
    assert matches == [matching.Match(0, 1, 5)]


def test_generic_route_equals_oracle_on_random_tables():
    from oracle import match_oracle as mo

    rng = np.random.default_rng(3)
    for nA, nB in [(7, 1), (5, 2), (9, 13), (30, 40)]:
        table = rng.integers(1, 15, size=(nA, nB)).astype(float)  # ties on purpose
        fa = [Feature(float(i), 0.0) for i in range(nA)]
        fb = [Feature(float(j), 1.0) for j in range(nB)]
        fn = lambda a, b: table[int(a.x), int(b.x)]
        combos = {None: None, "r": {mo.RATIO_TEST}, "c": {mo.CROSSCHECK}, "rc": {mo.RATIO_TEST, mo.CROSSCHECK}}
        strat = {None: None, "r": matching.ValidationStrategy.RATIO_TEST,
                 "c": {matching.ValidationStrategy.CROSSCHECK},
                 "rc": {matching.ValidationStrategy.RATIO_TEST, matching.ValidationStrategy.CROSSCHECK}}
        for key in combos:
            got = matching.match_brute_force(fa, fb, fn, validation_strategies=strat[key], ratio_test_threshold=0.8)
            want = mo.match_brute_force(table, combos[key], 0.8)
            assert [(m.a_index, m.b_index, m.match_score) for m in got] == want


def test_empty_inputs():
    fa = [Feature(1, 1)]
    with pytest.raises(IndexError):
        matching.match_brute_force(fa, [], lambda a, b: 0.0)
    assert matching.match_brute_force(fa, [], lambda a, b: 0.0,
                                      validation_strategies=matching.ValidationStrategy.RATIO_TEST) == []
    assert matching.match_brute_force([], [Feature(0, 0)], lambda a, b: 0.0) == []
    # the reference's order of operations with every heap empty (matching.py:55-81): no features at all -> [];
    # the ratio filter drops empty heaps before anything is indexed; the cross-check alone indexes heap[0]
    cross = matching.ValidationStrategy.CROSSCHECK
    ratio = matching.ValidationStrategy.RATIO_TEST
    assert matching.match_brute_force([], [], lambda a, b: 0.0) == []
    assert matching.match_brute_force([], [], lambda a, b: 0.0, validation_strategies=cross) == []
    with pytest.raises(IndexError):
        matching.match_brute_force(fa, [], lambda a, b: 0.0, validation_strategies=cross)
    assert matching.match_brute_force(fa, [], lambda a, b: 0.0, validation_strategies={ratio, cross}) == []


def test_crosscheck_with_nan_scores():
    """A NaN best score: the reference's cross-check compares `Match` objects with `==` (matching.py:113-117), and a
    claimant compared with itself is equal through the identity shortcut of tuple comparison even when its score is
    NaN — the row stays; a NaN claimant is never displaced (`nan > x` is false, :105-110)."""
    import math

    fa = [Feature(0.0, 0.0), Feature(1.0, 0.0), Feature(2.0, 0.0)]
    fb = [Feature(0.0, 1.0), Feature(1.0, 1.0)]
    nan = float("nan")
    table = {0: {0: nan, 1: 5.0}, 1: {0: 1.0, 1: 7.0}, 2: {0: 9.0, 1: 2.0}}
    got = matching.match_brute_force(fa, fb, _table_function(fa, fb, table),
                                     validation_strategies=matching.ValidationStrategy.CROSSCHECK)
    # literal restatement of the reference's two steps on heaps of Match objects
    import heapq
    heaps = []
    for a in range(3):
        heap = []
        for b in range(2):
            heapq.heappush(heap, matching.Match(a_index=a, b_index=b, match_score=table[a][b]))
        heaps.append(heap)
    best_for_b = {}
    for heap in heaps:
        m = heap[0]
        if m.b_index not in best_for_b or best_for_b[m.b_index].match_score > m.match_score:
            best_for_b[m.b_index] = m
    want = [heap[0] for heap in heaps if heap[0] == best_for_b[heap[0].b_index]]
    key = lambda m: (m.a_index, m.b_index, "nan" if math.isnan(m.match_score) else m.match_score)  # noqa: E731
    assert [key(m) for m in got] == [key(m) for m in want]
    assert any(math.isnan(m.match_score) for m in want)   # the case is exercised


def test_score_function_recognition():
    ia, ib = np.zeros((20, 30)), np.ones((20, 30))
    spec = _device_score_spec(ImagePairScore(ia, ib, ncc.calculate_ncc, 9))
    assert spec[0] == 0 and spec[1] is ia and spec[2] is ib and spec[3] == 9
    assert _device_score_spec(ImagePairScore(ia, ib, ssd.calculate_ssd))[3] == 5         # ssd default window
    assert _device_score_spec(functools.partial(ncc.calculate_ncc, ia, ib, window_size=7))[3] == 7
    assert _device_score_spec(functools.partial(ncc.calculate_ncc, ia, ib))[3] == 3       # ncc default window

    def _create_score_function(image_a, image_b, full_score_function):                    # apps/sfm.py:247-257
        def ssd_score(feature_a, feature_b):
            return full_score_function(image_a, image_b, feature_a, feature_b)
        return ssd_score

    spec = _device_score_spec(_create_score_function(ia, ib, functools.partial(ncc.calculate_ncc, window_size=9)))
    assert spec[0] == 0 and spec[1] is ia and spec[2] is ib and spec[3] == 9
    assert _device_score_spec(_create_score_function(ia, ib, ssd.calculate_ssd)) == (1, ia, ib, 5)
    assert _device_score_spec(lambda a, b: 0.0) is None
    assert _device_score_spec(_create_score_function(ia, ib, lambda A, B, a, b: 1.0)) is None
    with pytest.raises(TypeError):
        ImagePairScore(ia, ib, lambda *a: 0.0)


def test_util_helpers():
    """reference test_util.py."""
    within = functools.partial(util.is_within_bounds, image_shape=(100, 200), window_size=5)
    assert within(Feature(2, 2)) and not within(Feature(1, 2)) and not within(Feature(2, 1))
    assert within(Feature(y=97, x=197)) and not within(Feature(y=98, x=197)) and not within(Feature(y=97, x=198))
    img = np.arange(100).reshape(10, 10)
    np.testing.assert_array_equal(util.select_window(img, Feature(x=4.7, y=5.2), 3), img[4:7, 3:6])


def test_match_dataclass():
    m = matching.Match()
    assert (m.a_index, m.b_index, m.match_score) == (-1, -1, np.inf)
    assert matching.Match(0, 0, 1.0) < matching.Match(5, 5, 2.0)
    assert {s.name for s in matching.ValidationStrategy} == {"CROSSCHECK", "RATIO_TEST"}
    assert matching.ValidationStrategy.CROSSCHECK.value == 1 and matching.ValidationStrategy.RATIO_TEST.value == 2
