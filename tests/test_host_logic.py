"""CPU tests of the host side: C-ABI exports, the exact random.shuffle replay, the generic RANSAC driver,
the boundary value types, and the "fail loudly without a GPU" contract.  No kernel is launched."""
import ctypes as C
import os
import random
import re
from functools import partial
from math import isclose, sqrt

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(native_lib):
    """Every function include/sfm_hip.h declares is exported by libsfm_hip.so and bound in _native."""
    from structure_from_motion_amd import _native

    header = open(os.path.join(REPO, "include", "sfm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 12
    bound = set(_native.SIGNATURES) | set(_native.OTHER_SYMBOLS)
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(native_lib, name) is not None
    assert native_lib.sfm_abi_version() == _native.ABI_VERSION
    assert C.sizeof(_native.SelectResult) == 40


def test_score_workspace_holds_every_region(native_lib):
    """sfm_score_workspace_bytes(_ex) (csrc/sfm_score_ws.h), sized by what the call will launch (round 5; round 3's advisor):
    always fp32 points, class counters and the scoring order; the partials of the k ranges the launcher picks for (n, h, batch,
    options) — 20 k bytes per hypothesis —; the matrix-pipe kernel's tables (96 bytes per point, 96 + 20 per hypothesis) only
    where that kernel is chosen; the recorded pre-pass (512 bytes per hypothesis) only for a single pair of eight ranges."""
    from structure_from_motion_amd import _native

    plain = native_lib.sfm_score_workspace_bytes

    def size(n, h, b, **options):
        return native_lib.sfm_score_workspace_bytes_ex(n, h, b, C.byref(_native.ScoreOptions(**options)) if options else None)

    n, h = 50_000, 100_000
    one = size(n, h, 1)
    assert one == plain(n, h, 1)                                     # NULL options = the process-wide defaults
    tables = 96 * ((n + 31) // 32 * 32) + (96 + 20) * h
    assert 16 * n + 4 * h + 4 * h + 8 * 20 * h + tables + 512 * h <= one < 16 * n + 8 * h + 8 * 20 * h + tables + 512 * h + (1 << 20)   # 86 MB
    assert size(n + 32, h, 1) > one and size(n, h + 4, 1) > one
    valu = size(n, h, 1, kernel="filtered")                          # the VALU filter: 2 ranges of this size, no tables, no record
    assert 16 * n + 8 * h + 2 * 20 * h <= valu < 16 * n + 8 * h + 2 * 20 * h + (1 << 20) and valu < one / 8
    assert size(n, h, 1, kernel="filtered", split=0) < valu          # no ranges: no partials
    assert size(n, h, 1, split=16) > size(n, h, 1, split=4)          # more ranges asked for: more partials (and no recorded pre-pass)
    big = size(5_000_000, h, 1)
    assert big < size(4_194_304, h, 1)          # no operand tables for a pair the matrix-pipe kernel cannot take (> 4 M points)
    assert big >= 16 * 5_000_000 + 8 * h
    assert size(70_000, h, 1) > size(65_536, h, 1) >= 96 * 65_536    # (the cap of rounds 1-3 is gone)
    # BASELINE configs[4]: 256 pairs x 10 000 x 2 000 — matrix-pipe kernel by size (9 ranges), 47 MB with the VALU filter
    c5 = size(10_000, 2_000, 256)
    assert native_lib.sfm_score_kernel_choice(10_000, 2_000, 256) == 2
    assert 256 * (16 * 10_000 + 4 * 2_000 + 9 * 20 * 2_000 + 96 * 10_016 + 116 * 2_000) <= c5 < 460_000_000
    c5_valu = size(10_000, 2_000, 256, kernel="filtered")
    assert 256 * (16 * 10_000 + 4 * 2_000) <= c5_valu < 100_000_000     # under 100 MB when the VALU kernel is chosen
    assert size(10_000, 2_000, 257) > c5
    assert size(10_000, 2_000, 16) < 16 * (16 * 10_000 + 8 * 2_000) + (1 << 20)   # a batch too small for the matrix-pipe kernel: no tables
    assert size(300, 64, 1, kernel="matrix") > size(300, 64, 1)      # forced on: the tables are reserved
    assert plain(-1, 5, 1) == -1 and plain(5, -1, 1) == -1 and plain(5, 5, -1) == -1
    assert size(300, 64, 1, kernel=7) == -1                          # bad options


def test_score_kernel_choice_rule(native_lib):
    """sfm_score_kernel_choice_ex: 2 = the matrix-pipe kernel — single pairs from 4000 points x 2048 hypotheses x 3.5e8 evaluations,
    batches from 8192 x 1024 per pair with 6144 waves and 5e8 evaluations over the batch, never beyond 4 M points per pair;
    options.kernel forces it on (where it applies) / off."""
    from structure_from_motion_amd import _native

    auto, valu, matrix = (C.byref(_native.ScoreOptions(kernel=k)) for k in ("auto", "filtered", "matrix"))

    def choice(n, h, b, options=auto):
        return native_lib.sfm_score_kernel_choice_ex(n, h, b, options)

    assert choice(50_000, 100_000, 1) == 2 and choice(50_000, 125_000, 1) == 2 and choice(50_000, 10_000, 1) == 2
    assert choice(20_000, 20_000, 1) == 2 and choice(25_000, 12_000, 1) == 1 and choice(8_192, 42_725, 1) == 2 and choice(8_192, 42_724, 1) == 1
    assert choice(5_000, 10_000, 1) == 1 and choice(16_000, 16_000, 1) == 1 and choice(3_999, 1_000_000, 1) == 1
    assert choice(4_000, 87_500, 1) == 2 and choice(4_000, 87_499, 1) == 1 and choice(200_000, 2_048, 1) == 2 and choice(200_000, 2_047, 1) == 1
    assert choice(70_000, 100_000, 1) == 2 and choice(5_000_000, 100_000, 1) == 1
    assert choice(10_000, 2_000, 256) == 2 and choice(10_000, 2_000, 16) == 1 and choice(4_000, 2_000, 256) == 1
    assert choice(50_000, 100_000, 1, valu) == 1
    assert choice(300, 64, 1, matrix) == 2 and choice(70_000, 100_000, 1, matrix) == 2 and choice(5_000_000, 64, 1, matrix) == 1
    assert choice(-1, 5, 1) == -1
    bad = _native.ScoreOptions(kernel=7)
    assert choice(300, 64, 1, C.byref(bad)) == -1


def test_score_options_are_set_by_the_caller_not_read_from_the_environment(native_lib, monkeypatch):
    """The library never calls getenv for its launch options: the package translates SFM_SCORE_* once into
    a ScoreOptions (`_native.score_options_from_env`) and hands it over with sfm_score_set_default_options; calls that pass
    NULL options use that set, calls that pass options ignore it; changing the environment afterwards changes nothing."""
    from structure_from_motion_amd import _native

    from_env = _native.score_options_from_env
    fields = [f[0] for f in _native.ScoreOptions._fields_ if not f[0].startswith("timing_")]

    def values(o):
        return [getattr(o, f) for f in fields]

    assert values(from_env({})) == [0, 0, -1, -1, -1, -1, -1, -1] == values(_native.ScoreOptions())
    env = {"SFM_SCORE_MATRIX": "1", "SFM_SCORE_HPW": "2", "SFM_SCORE_SPLIT": "3", "SFM_SCORE_ORDER": "0",
           "SFM_SCORE_ONE_SIDED": "0", "SFM_SCORE_XCD": "0", "SFM_SCORE_SYNC": "4", "SFM_SCORE_PERSISTENT": "0"}
    assert values(from_env(env)) == [2, 2, 3, 0, 0, 0, 4, 0]
    assert values(from_env({"SFM_SCORE_MATRIX": "0", "SFM_SCORE_HPW": "3", "SFM_SCORE_SPLIT": "x"})) == [1, 0, -1, -1, -1, -1, -1, -1]
    assert C.sizeof(_native.ScoreOptions) == 48   # eight int32 + the call's two timing events (ABI 11)
    # the timing events belong to ONE call: they are never part of the process-wide defaults
    timed = _native.ScoreOptions(kernel="filtered", timing=(0x1234, 0x5678))
    assert (timed.timing_before, timed.timing_after) == (0x1234, 0x5678)
    assert timed.with_timing(None, None).timing_before is None and timed.with_timing(None, None).kernel == 1
    before = _native.ScoreOptions()
    assert native_lib.sfm_score_get_default_options(C.byref(before)) == 0
    try:
        monkeypatch.setenv("SFM_SCORE_MATRIX", "0")                      # ignored: the library does not look
        if before.kernel == 0:
            assert native_lib.sfm_score_kernel_choice(50_000, 100_000, 1) == 2
        forced = _native.ScoreOptions(kernel="filtered", timing=(0x1234, 0x5678))
        assert native_lib.sfm_score_set_default_options(C.byref(forced)) == 0
        assert native_lib.sfm_score_kernel_choice(50_000, 100_000, 1) == 1
        stored = _native.ScoreOptions()
        native_lib.sfm_score_get_default_options(C.byref(stored))
        assert stored.kernel == 1 and stored.timing_before is None and stored.timing_after is None   # cleared on the way in
        assert native_lib.sfm_score_kernel_choice_ex(50_000, 100_000, 1, C.byref(_native.ScoreOptions())) == 2   # per call wins
        bad = _native.ScoreOptions(split=-5)
        assert native_lib.sfm_score_set_default_options(C.byref(bad)) == -1
        assert native_lib.sfm_score_kernel_choice(50_000, 100_000, 1) == 1   # a refused set leaves the old one in place
        assert native_lib.sfm_score_set_default_options(None) == 0           # NULL: the built-in defaults
        now = _native.ScoreOptions(kernel=2)
        native_lib.sfm_score_get_default_options(C.byref(now))
        assert values(now) == [0, 0, -1, -1, -1, -1, -1, -1]
    finally:
        native_lib.sfm_score_set_default_options(C.byref(before))
    # no getenv left in the scoring translation unit
    src = open(os.path.join(REPO, "structure_from_motion_amd", "csrc", "sfm_score.hip")).read()
    assert "getenv" not in src


def test_argument_validation_without_gpu(native_lib):
    """Bad sizes are rejected before any HIP call (safe on a CPU-only box)."""
    assert native_lib.sfm_sample_philox(1, 1, 0, 10, 7, 1, None, None) == -1
    assert b"n" in native_lib.sfm_last_error()
    assert native_lib.sfm_fit_eight_point(None, 5, None, 4, 1, None, None, None, None) == -1
    assert native_lib.sfm_score_sed(None, -1, None, None, 4, 1, 0.0, None, None, None, None, 0, None) == -1
    assert native_lib.sfm_score_workspace_bytes(1000, 50, 2) >= 2 * 16 + 2 * 1000 * 16 + 2 * 50 * 4
    assert native_lib.sfm_cheirality_batched(None, 0, 3, None, None, 50.0, None, None) == 0
    assert native_lib.sfm_pose_vote(None, 5, 2, None, None, None, None) == -1
    assert native_lib.sfm_select_best(None, None, None, None, 4, 1, 0.0, 9, 0, None, None) == -1
    # empty work is a successful no-op
    assert native_lib.sfm_triangulate(None, 0, None, None, None, None) == 0
    assert native_lib.sfm_cheirality(None, 0, None, 4, 50.0, None, None) == 0
    # the fused passes (ABI 11: trailing launch options; sfm_ransac_pass_batch): sizes, null pointers, options and the
    # plan-sized workspace are checked before anything is launched
    from structure_from_motion_amd import _native

    p = C.c_void_p(0x1000)
    bad = C.byref(_native.ScoreOptions(split=-7))
    for entry in (native_lib.sfm_ransac_pass_small, native_lib.sfm_ransac_pass_large):
        assert entry(1, None, 1, 0, p, 7, 100, 1e-6, 10.0, 3, 0, p, p, p, p, p, p, p, None, p, 1 << 40, None, None) == -1        # n < 8
        assert entry(1, None, 1, 0, p, 100, 100, 1e-6, 10.0, 9, 0, p, p, p, p, p, p, p, None, p, 1 << 40, None, None) == -1      # aggregation
        assert entry(1, None, 1, 0, p, 100, 100, 1e-6, 10.0, 3, 0, p, p, None, p, p, p, p, None, p, 1 << 40, None, None) == -1   # null flags
        assert entry(1, None, 1, 0, p, 100, 100, 1e-6, 10.0, 3, 0, p, p, p, p, p, p, p, None, p, 1 << 40, None, bad) == -1       # bad options
        assert b"option" in native_lib.sfm_last_error()
        assert entry(1, None, 1, 0, p, 100, 100, 1e-6, 10.0, 3, 0, p, p, p, p, p, p, p, None, p, 16, None, None) == -1           # workspace too small
        assert b"workspace" in native_lib.sfm_last_error()
    batch = native_lib.sfm_ransac_pass_batch
    assert batch(1, None, 1, 1, 0, p, 100, 100, 0, 1e-6, 10.0, 3, p, p, p, p, p, p, p, None, p, 1 << 40, None, None) == 0         # no pairs: nothing to do
    assert batch(1, None, 1, 1, 0, p, 100, 100, 70000, 1e-6, 10.0, 3, p, p, p, p, p, p, p, None, p, 1 << 40, None, None) == -1    # grid y
    assert batch(1, None, 1, 1, 0, p, 100, 0, 4, 1e-6, 10.0, 3, p, p, p, p, p, p, p, None, p, 1 << 40, None, None) == -1          # no hypotheses
    assert batch(1, None, 1, 1, 0, p, 100, 100, 4, 1e-6, 10.0, 3, p, p, p, p, p, p, None, None, p, 1 << 40, None, None) == -1     # null result
    assert batch(1, None, 1, 1, 0, p, 100, 100, 4, 1e-6, 10.0, 3, p, p, p, p, p, p, p, None, p, 16, None, None) == -1             # workspace too small
    # a workspace sized for the process-wide options is refused by a call whose options need the matrix-pipe kernel's tables
    n, h = 9_000, 5_000
    small = native_lib.sfm_score_workspace_bytes_ex(n, h, 1, C.byref(_native.ScoreOptions(kernel="filtered")))
    forced = C.byref(_native.ScoreOptions(kernel="matrix"))
    assert native_lib.sfm_score_workspace_bytes_ex(n, h, 1, forced) > small
    ws = C.c_void_p(0x10000)
    assert native_lib.sfm_score_sed_ex(p, n, p, p, h, 1, 1e-6, p, p, p, ws, small, None, forced) == -1
    assert b"workspace" in native_lib.sfm_last_error()


def test_sizes_beyond_one_launch_are_refused(native_lib):
    """Kernels without a grid-stride loop must cover every item with a block of its own: sizes beyond one launch's
    limits (2^31-1 blocks / 2^32-1 threads in x, 65535 in y) return SFM_EINVAL before any HIP call instead of
    silently scoring a prefix (include/sfm_hip.h, "size limits of ONE call").  Pointers are fake non-null values:
    nothing may be dereferenced or launched on this path."""
    p = C.c_void_p(0x1000)
    big_h = 1 << 28          # 16 hypotheses per 256-thread block -> 2^32 threads: one too many
    assert native_lib.sfm_score_sed(p, 50_000, p, p, big_h + 16, 1, 1e-6, p, p, p, None, 0, None) == -1
    assert b"one launch" in native_lib.sfm_last_error()
    ws_bytes = native_lib.sfm_score_workspace_bytes(50_000, big_h + 16, 1)
    ws = C.c_void_p(0x10000)
    assert native_lib.sfm_score_sed(p, 50_000, p, p, big_h + 16, 1, 1e-6, p, p, p, ws, ws_bytes, None) == -1
    assert b"one launch" in native_lib.sfm_last_error()
    assert native_lib.sfm_fit_eight_point(p, 100, p, 1 << 32, 1, p, p, None, None) == -1
    assert b"one launch" in native_lib.sfm_last_error()
    assert native_lib.sfm_sample_fit_philox(1, None, 1, 0, p, 100, 1 << 32, 1, p, p, p, None) == -1
    assert native_lib.sfm_sample_philox(1, 1, 0, 1 << 32, 100, 1, p, None) == -1
    assert native_lib.sfm_cheirality(p, 1 << 32, p, 4, 50.0, p, None) == -1
    assert native_lib.sfm_triangulate(p, 1 << 32, p, p, p, None) == -1
    assert native_lib.sfm_decompose_essential(p, 1 << 32, p, p, None) == -1
    # batch is the grid's y dimension
    assert native_lib.sfm_fit_eight_point(p, 100, p, 64, 65536, p, p, None, None) == -1
    assert native_lib.sfm_sample_philox(1, 1, 0, 64, 100, 65536, p, None) == -1
    assert native_lib.sfm_score_sed(p, 100, p, p, 64, 65536, 1e-6, p, p, p, None, 0, None) == -1
    assert native_lib.sfm_cheirality_batched(p, 100, 65536, p, p, 50.0, p, None) == -1


@pytest.mark.parametrize("n,iters,seed", [(10, 100, 5), (75, 30, 0), (2, 10, 1), (1, 3, 2), (1000, 7, 123456789)])
def test_pyshuffle_replay_matches_cpython(native_lib, n, iters, seed):
    from structure_from_motion_amd import device

    rng = random.Random(seed)
    rng.random()  # move off the freshly seeded state (index != 624)
    twin = random.Random()
    twin.setstate(rng.getstate())
    S, snap = device.pyshuffle_table(n, iters, rng, snapshot_iteration=iters // 2)
    perm = list(range(n))
    for it in range(iters):
        twin.shuffle(perm)
        np.testing.assert_array_equal(S[it][:min(n, 8)], perm[:8])
        if it == iters // 2:
            np.testing.assert_array_equal(snap, perm)
    assert rng.getstate() == twin.getstate()  # generator advanced exactly as CPython would
    if n < 8:
        assert (S[:, n:] == -1).all()


@pytest.mark.parametrize("wide", ["1", "0"])
@pytest.mark.parametrize("n,iters,seed", [(5000, 40, 1), (70000, 3, 2), (1023, 20, 3), (1024, 20, 4), (526, 50, 5), (527, 50, 6),
                                          (4111, 10, 7), (8191, 6, 8), (8192, 6, 9), (300, 200, 10), (65536 + 15, 2, 11)])
def test_pyshuffle_replay_wide_and_plain_paths(native_lib, monkeypatch, n, iters, seed, wide):
    """The sixteen-at-a-time draw classification (AVX-512, segments from i = 511 up) and the one-at-a-time form give
    CPython's permutation and leave CPython's generator state, at sizes around every place the two hand over: segment
    ends (2^k - 1), the first block of a segment, the last words of a 624-word state block."""
    from structure_from_motion_amd import device

    monkeypatch.setenv("SFM_PYSHUFFLE_WIDE", wide)
    rng = random.Random(seed)
    for _ in range(seed):
        rng.random()  # a different position inside the state block per case
    twin = random.Random()
    twin.setstate(rng.getstate())
    S, snap = device.pyshuffle_table(n, iters, rng, snapshot_iteration=iters - 1)
    perm = list(range(n))
    for it in range(iters):
        twin.shuffle(perm)
        np.testing.assert_array_equal(S[it], perm[:8])
    np.testing.assert_array_equal(snap, perm)
    assert rng.getstate() == twin.getstate()


def test_pyshuffle_replay_uses_global_random(native_lib):
    from structure_from_motion_amd import device

    random.seed(5)
    S, _ = device.pyshuffle_table(10, 100)
    after = random.random()
    d = np.load(os.path.join(REPO, "tests", "golden", "g2_ransac_seed5.npz"))
    np.testing.assert_array_equal(S, d["S"])  # the samples the real reference drew after random.seed(5)
    random.seed(5)
    p = list(range(10))
    for _ in range(100):
        random.shuffle(p)
    assert random.random() == after
    random.seed(5)
    S2, _ = device.pyshuffle_table(10, 100, advance=False)
    p = list(range(10))
    random.shuffle(p)
    np.testing.assert_array_equal(S2[0], p[:8])  # state was not advanced


@pytest.mark.parametrize("n,iters,checkpoints", [(10, 100, 32), (341, 257, 7), (9, 5, 32), (50, 64, 1), (12, 1, 4)])
def test_checkpointed_shuffle_table(native_lib, n, iters, checkpoints):
    """PyShuffleTable == literal CPython shuffles: the table, the generator state afterwards, and the full
    permutation after any iteration, re-derived from the nearest checkpoint."""
    from structure_from_motion_amd import device

    rng, twin = random.Random(11), random.Random(11)
    table = device.PyShuffleTable(n, iters, rng, checkpoints=checkpoints)
    perm, history = list(range(n)), []
    for _ in range(iters):
        twin.shuffle(perm)
        history.append(list(perm))
    np.testing.assert_array_equal(table.S, np.array([h[:8] for h in history]))
    assert rng.getstate() == twin.getstate()
    for it in sorted({0, iters - 1, iters // 2, iters // 3, max(0, iters - 2)}):
        np.testing.assert_array_equal(table.permutation_after(it), history[it])
    with pytest.raises(IndexError):
        table.permutation_after(iters)
    untouched = random.Random(11)
    device.PyShuffleTable(n, iters, untouched, advance=False)
    assert untouched.getstate() == random.Random(11).getstate()


# ------------------------------------------------------------------------------------------------------
# generic driver (reference lib/ransac/tests/test_ransac.py)
# ------------------------------------------------------------------------------------------------------
def line_fitter_2d(points):
    if np.allclose(points[0], points[1]):
        raise ValueError("Cannot fit line on the same two points.")
    if len(points) != 2:
        raise ValueError(f"Two points are needed for line fitting, got {len(points)}.")
    dx = points[1][0] - points[0][0]
    if abs(dx) <= 1e-6:
        return (1.0, 0.0, -points[0][0])
    slope = (points[1][1] - points[0][1]) / dx
    return (slope, -1.0, points[0][1] - slope * points[0][0])


def line_scorer_2d(model, point):
    return abs(model[0] * point[0] + model[1] * point[1] + model[2]) / sqrt(model[0] ** 2 + model[1] ** 2)


def test_ransac_line_fit_equals_reference(golden):
    """test_ransac.py:70-123 — and bit-equal to what the real reference returned (golden g10)."""
    from lib.ransac import ransac

    d = golden("g10_line_ransac")
    all_points_list = list(d["points"])
    snapshot = d["points"].copy()
    random.seed(5)
    model, inliers = ransac.fit_with_ransac(
        data=all_points_list, model_fit_data_count=2, model_fitter=line_fitter_2d,
        inlier_scorer=line_scorer_2d, inlier_threshold=0.2,
        min_num_extra_inliers=len(all_points_list) / 2,  # a float, as in the reference test
        error_aggregation_method=ransac.ErrorAggregationMethod.RMS)
    inliers = np.array(inliers)
    assert 0 <= len(inliers) - 50 <= 3
    assert isclose(0.6, -model[0] / model[1], rel_tol=0, abs_tol=1e-7)
    assert isclose(-np.sign(model[1]) * (5 - 4 * 0.6), model[2])
    np.testing.assert_array_equal(np.array(model), d["model"])
    np.testing.assert_array_equal(inliers, d["inliers"])
    np.testing.assert_array_equal(np.array(all_points_list), snapshot)  # input not mutated


def test_ransac_defaults_and_failure():
    from lib.ransac import ransac

    pts = [np.array([float(i), 2.0 * i]) for i in range(20)]
    random.seed(0)
    model, inliers = ransac.fit_with_ransac(pts, 2, line_fitter_2d, line_scorer_2d, 1e-9)
    assert len(inliers) == 20 and isclose(-model[0] / model[1], 2.0)
    with pytest.raises(ValueError, match="No model could be found with at least 102 inliers"):
        ransac.fit_with_ransac(pts, 2, line_fitter_2d, line_scorer_2d, 1e-9, min_num_extra_inliers=100)
    with pytest.raises(ValueError):
        ransac.fit_with_ransac(pts, 2, line_fitter_2d, line_scorer_2d, 1e-9, max_iterations=0)

    def nan_scorer(model, p):
        return float("nan")

    with pytest.raises(ValueError):  # NaN errors never win (ransac.py:83)
        ransac.fit_with_ransac(pts, 2, line_fitter_2d, nan_scorer, 1.0)

    def boom(points):
        raise KeyError("fitter failure aborts the call")

    with pytest.raises(KeyError):
        ransac.fit_with_ransac(pts, 2, boom, line_scorer_2d, 1.0)


def test_aggregate_error_modes():
    from lib.ransac.ransac import ErrorAggregationMethod, _aggregate_error

    e = [1.0, 2.0, 3.0, 4.0]
    assert _aggregate_error(e, ErrorAggregationMethod.SUM) == 10.0
    assert _aggregate_error(e, ErrorAggregationMethod.SQUARE) == 30.0
    assert _aggregate_error(e, ErrorAggregationMethod.MEAN) == 2.5
    assert _aggregate_error(e, ErrorAggregationMethod.RMS) == sqrt(7.5)
    assert [m.value for m in ErrorAggregationMethod] == ["sum", "square", "mean", "rms"]


def test_device_route_detection():
    from structure_from_motion_amd.epipolar import epipolar_ransac as er
    from structure_from_motion_amd.ransac import ransac

    K = np.eye(3)
    fit = partial(er.eight_point_model_fitter, camera_matrix=K)
    score = partial(er.calculate_sed_inlier_score, camera_matrix=K)
    assert np.array_equal(ransac._device_spec(fit, score, 8), K)
    assert ransac._device_spec(fit, score, 7) is None
    assert ransac._device_spec(line_fitter_2d, line_scorer_2d, 8) is None
    assert ransac._device_spec(fit, partial(er.calculate_sed_inlier_score, camera_matrix=2 * K), 8) is None
    assert ransac._device_spec(fit, partial(line_scorer_2d), 8) is None


# ------------------------------------------------------------------------------------------------------
# fail loudly without a GPU / library
# ------------------------------------------------------------------------------------------------------
def test_numeric_entry_points_fail_loudly_without_gpu(native_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lib.common.feature import Feature
    from lib.epipolar import eight_point, epipolar_ransac
    from lib.epipolar.sed import calculate_symmetric_epipolar_distance
    from lib.epipolar.triangulation import triangulate_points
    from lib.transforms.transforms import Transform3D

    fa = [Feature(float(i), float(i * i % 7)) for i in range(10)]
    m = eight_point.create_trivial_matches(10)
    K = np.array([[50.0, 0, 256], [0, 50.0, 128], [0, 0, 1]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        epipolar_ransac.estimate_essential_mat_with_ransac(K, fa, fa, m, 0.01)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eight_point.estimate_fundamental_mat(fa, fa, m[:8])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eight_point.recover_r_t_from_e(np.eye(3), K, fa, fa)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        triangulate_points(fa, fa, K, Transform3D.identity())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        calculate_symmetric_epipolar_distance(fa[0], fa[1], np.eye(3))


def test_unwritten_mask_is_an_error():
    """The 0xFF fill of a mask slice whose selection record never arrived (select_sharded_kernel's bounded wait)
    surfaces as an error on the host, not as an empty inlier list."""
    from structure_from_motion_amd import device

    ok = np.array([0, 1, 2, 1, 0], dtype=np.uint8)
    assert device.checked_mask(ok) is ok
    assert device.checked_mask(np.zeros(0, dtype=np.uint8)).size == 0
    with pytest.raises(RuntimeError, match="inlier mask was not written"):
        device.checked_mask(np.array([0, 1, 0xFF, 0xFF], dtype=np.uint8))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from structure_from_motion_amd import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()


def test_product_code_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/; nothing that ships or
    runs on the GPU box may read /root/reference."""
    roots = [os.path.join(REPO, d) for d in ("structure_from_motion_amd", "lib", "apps", "tools", "include")]
    files = [os.path.join(REPO, "main.py")]
    for top in roots:
        for root, _, names in os.walk(top):
            files += [os.path.join(root, f) for f in names if f.endswith((".py", ".hip", ".h", ".cpp"))]
    assert len(files) > 40
    for path in files:
        text = open(path).read()
        assert "import oracle" not in text and "from oracle" not in text, path
        assert "/root/reference" not in text, path
    # bench.py: the oracle only inside cpu_baseline(); __graft_entry__: only inside smoke() / build()
    bench = open(os.path.join(REPO, "bench.py")).read()
    head, _, tail = bench.partition("def cpu_baseline(")
    body, _, rest = tail.partition("\ndef ")
    assert "oracle" not in head.replace("oracle port", "") and "from oracle" in body and "from oracle" not in rest
    assert "/root/reference" not in bench


def test_no_function_reads_an_undefined_global():
    """GPU-only code paths cannot run in the build container: at least every global a function reads must exist at
    module level (a name imported locally in one method and used in another is the typical slip)."""
    import ast
    import builtins
    import dis
    import types

    def module_level_names(tree):
        names = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__spec__", "__package__", "__builtins__"}
        stack = [(tree, True)]
        while stack:
            node, top = stack.pop()
            for child in ast.iter_child_nodes(node):
                inner = top and not isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Lambda))
                if top:
                    if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                        names.add(child.name)
                    elif isinstance(child, (ast.Import, ast.ImportFrom)):
                        names.update((a.asname or a.name).split(".")[0] for a in child.names)
                    elif isinstance(child, ast.Name) and isinstance(child.ctx, ast.Store):
                        names.add(child.id)
                stack.append((child, inner))
        return names

    def functions(code):
        for const in code.co_consts:
            if isinstance(const, types.CodeType):
                is_class_body = "__qualname__" in const.co_names and "__module__" in const.co_names
                if not is_class_body:
                    yield const
                yield from functions(const)

    files = [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py"), os.path.join(REPO, "main.py")]
    for top in ("structure_from_motion_amd", "lib", "apps", "tools"):
        for root, _, names in os.walk(os.path.join(REPO, top)):
            files += [os.path.join(root, f) for f in names if f.endswith(".py")]
    problems = []
    for path in files:
        source = open(path).read()
        known = module_level_names(ast.parse(source, path))
        for fn in functions(compile(source, path, "exec")):
            for ins in dis.get_instructions(fn):
                if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in known:
                    problems.append(f"{os.path.relpath(path, REPO)}: {fn.co_name} reads {ins.argval}")
    assert not problems, problems


# ------------------------------------------------------------------------------------------------------
# boundary value types (reference a20)
# ------------------------------------------------------------------------------------------------------
def test_value_types():
    from lib.common.feature import Feature
    from lib.feature_matching.matching import Match
    from lib.transforms.transforms import Transform3D
    from lib.epipolar.eight_point import create_trivial_matches, to_normalized_image_coords

    f = Feature(1.0, 2.0)
    f.x = 3.0
    assert f == Feature(3.0, 2.0) and Feature(x=1, y=2).y == 2
    m = Match()
    assert (m.a_index, m.b_index, m.match_score) == (-1, -1, np.inf)
    assert Match(1, 2, 0.1) < Match(3, 4, 0.2) and not (Match(1, 2, 0.3) < Match(3, 4, 0.2))
    tm = create_trivial_matches(3)
    assert [(x.a_index, x.b_index, x.match_score) for x in tm] == [(0, 0, 0.0), (1, 1, 0.0), (2, 2, 0.0)]
    K = np.array([[50.0, 0, 256], [0, 40.0, 128], [0, 0, 1]])
    nf = to_normalized_image_coords(Feature(x=50, y=60), K)
    expect = np.linalg.inv(K) @ np.array([50.0, 60.0, 1.0])
    np.testing.assert_allclose([nf.x, nf.y], expect[:2])

    R = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    T = Transform3D.from_rmat_t(R, np.array([[1.0], [2.0], [3.0]]))  # column vector is reshaped
    np.testing.assert_array_equal(T.t, [1, 2, 3])
    np.testing.assert_array_equal(T.Rmat, R)
    assert T.Tmat.shape == (4, 4) and T.Tmat[3].tolist() == [0, 0, 0, 1]
    T.t = np.array([4.0, 5.0, 6.0])
    assert T.Tmat[0, 3] == 4.0
    np.testing.assert_allclose((T @ T.inv()).Tmat, np.eye(4), atol=1e-15)
    np.testing.assert_array_equal((T * Transform3D.identity()).Tmat, T.Tmat)
    np.testing.assert_array_equal(Transform3D.from_rmat_t().Tmat, np.eye(4))
    assert str(T).startswith("Homogeneous transformation(")
    with pytest.raises(ValueError):
        Transform3D(np.eye(3))
    with pytest.raises(ValueError):
        Transform3D.from_rmat_t(np.eye(4))
    with pytest.raises(ValueError):
        Transform3D.from_rmat_t(np.eye(3), np.zeros(4))
    with pytest.raises(TypeError):
        T * 3


def test_bench_roofline_is_a_valu_issue_bound(monkeypatch, tmp_path):
    """bench.py's roofline block: bound = valu-issue with frac = priced VALU (+ MFMA issue) instructions / (1024 SIMDs x
    2.4 GHz x kernel time) <= 1 for the committed counters, both HBM views beside it, and `counters_stale` raised as soon as the
    record's fingerprint is not the one of the kernel sources in the tree; per-rank shards scale the record."""
    import importlib.util
    import json

    from structure_from_motion_amd import build

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(REPO, "profiles", "score_traffic.json")))
    n, h = rec["matches"], rec["hypotheses"]
    monkeypatch.delenv("SFM_SCORE_MATRIX", raising=False)
    assert bench.scoring_kernel_name("filtered", n, h) == rec["kernel_short"]   # the record is of the kernel the bench line runs
    roof = bench.roofline(n, h, kernel_ms=1.5, call_ms=1.6, variant="filtered")
    assert roof["bound"] == "valu-issue" and 0.3 < roof["frac"] <= 1.0 and roof["kernel"] == rec["kernel_short"]
    c = rec["counters"]
    f64 = sum(c[k] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    cycles = (c["SQ_INSTS_VALU"] - f64) * 2.0 + f64 * 4.0 + c.get("SQ_INSTS_MFMA", 0.0) * 8.0
    assert isclose(roof["frac"], cycles / 1024 / 2.4e9 / 1.5e-3, rel_tol=1e-12)
    # the bound the line carries beside the utilisation: the run's own inliers through the fp64 routine at full rate
    assert roof["fp64_floor"] is None and "utilisation" in roof["note"]
    floor = bench.roofline(n, h, 1.5, 1.6, "filtered", exact_evals=1.75e8)["fp64_floor"]
    assert isclose(floor["floor_ms"], 1.75e8 / 64 * 43 * 4.0 / 1024 / 2.4e9 * 1e3, rel_tol=1e-12)
    assert isclose(floor["frac_of_kernel"], floor["floor_ms"] / 1.5, rel_tol=1e-12) and floor["exact_evaluations"] == 175000000
    assert roof["hbm_algorithmic"]["frac"] > 1.0            # the L2-resident set: reported, labelled "not a bound"
    assert roof["hbm_physical"]["frac"] < 0.1
    assert roof["traffic"] == (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    assert roof["counters_stale"] == (rec["source_sha"] != build.score_source_sha())
    # a record taken on other sources is flagged, a record for another workload is not used
    stale = dict(rec, source_sha="0" * 16)
    path = tmp_path / "score_traffic.json"
    path.write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "COUNTERS", str(path))
    assert bench.roofline(n, h, 1.5, 1.6, "filtered")["counters_stale"] is True
    # another hypothesis count on the same point set (a multi-GPU shard): counters scale with the hypotheses, and say so
    shard = bench.roofline(n, h * 5 // 4, 3.0, 3.1, "filtered")
    assert shard["counters_from"]["scaled_from_hypotheses"] == h
    assert isclose(shard["frac"], 1.25 * cycles / 1024 / 2.4e9 / 3.0e-3, rel_tol=1e-12)
    # another point set: the record does not apply
    other = bench.roofline(n + 1, h, 1.5, 1.6, "filtered")
    assert other["frac"] is None and other["counters_stale"] is None


def test_sampler_default_is_the_exact_replay_at_every_size(monkeypatch):
    """The drop-in call keeps the reference's sample stream (cumulative random.shuffle, ransac.py:59-64) by default at
    every size — also at BASELINE configs[1] (5 000 x 10 000) and configs[2]; the size switch to the device sampler is
    an explicit opt-in (SFM_SAMPLER=auto, threshold SFM_AUTO_PHILOX_WORK)."""
    from structure_from_motion_amd.epipolar import _engine

    monkeypatch.delenv("SFM_SAMPLER", raising=False)
    monkeypatch.delenv("SFM_AUTO_PHILOX_WORK", raising=False)
    assert _engine.sampler_name(300, 2000) == "pyshuffle"
    assert _engine.sampler_name(5_000, 10_000) == "pyshuffle"
    assert _engine.sampler_name(50_000, 100_000) == "pyshuffle"
    monkeypatch.setenv("SFM_SAMPLER", "auto")
    assert _engine.sampler_name(300, 2000) == "pyshuffle"
    assert _engine.sampler_name(1000, 10_000) == "pyshuffle" and _engine.sampler_name(1001, 10_000) == "philox"
    monkeypatch.setenv("SFM_AUTO_PHILOX_WORK", "1e6")
    assert _engine.sampler_name(300, 2000) == "pyshuffle" and _engine.sampler_name(600, 2000) == "philox"
    monkeypatch.setenv("SFM_SAMPLER", "philox")
    assert _engine.sampler_name(8, 1) == "philox"
    monkeypatch.setenv("SFM_SAMPLER", "fastest")
    with pytest.raises(ValueError):
        _engine.sampler_name(8, 1)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` typed without a launcher starts its ranks as child processes (torch.distributed.run)
    before anything touches a GPU and relays their exit code.  Without a GPU the ranks stop at the library's "needs a GPU"
    error: what is checked here is the hand-over (two ranks started, their failure relayed, no WORLD_SIZE complaint)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SFM_DIST_BACKEND"] = "gloo"
    import torch

    if torch.cuda.is_available():
        pytest.skip("covered end to end by tests/test_gpu_api.py::test_bench_two_rank_rehearsal on a GPU box")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--matches", "64", "--hypotheses", "16", "--no-cpu-baseline", "--no-extras"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0
    assert "launch with torch.distributed.run" not in out.stderr          # the old refusal is gone
    assert "bench.py FAILED" in out.stderr and "local_rank: 1" in out.stderr  # two child ranks were started and reaped


def test_hostfast_helper_equals_the_python_paths(native_lib):
    """csrc/hostfast.c (CPython helper of the drop-in boundary) against the Python code it replaces: pair list from
    matches (reference epipolar_ransac.py:55-57), coordinates as arrays, deep copies of the picked pairs (reference
    ransac.py:59,76) — identical objects / values / types, and the shapes it declines fall through to the general path."""
    import copy

    from lib.common.feature import Feature
    from lib.feature_matching.matching import Match
    from structure_from_motion_amd import _native
    from structure_from_motion_amd.epipolar import _engine

    fast = _native.hostfast()
    assert fast is not None, "build() compiles csrc/_sfm_hostfast.so"
    rng = np.random.default_rng(4)
    n = 3000
    pa, pb = rng.random((n, 2)) * 600, rng.random((n, 2)) * 600
    fa = [Feature(x=float(x), y=float(y)) for x, y in pa]
    fb = [Feature(x=float(x), y=float(y)) for x, y in pb]
    perm = rng.permutation(n)
    matches = [Match(a_index=int(i), b_index=int(j)) for i, j in zip(perm, perm[::-1])]
    pairs = _engine.match_pairs(fa, fb, matches)
    want = [(fa[m.a_index], fb[m.b_index]) for m in matches]
    assert len(pairs) == n and all(p[0] is w[0] and p[1] is w[1] for p, w in zip(pairs, want))
    arr = _engine.pair_arrays(pairs)
    np.testing.assert_array_equal(arr[0], pa[perm])
    np.testing.assert_array_equal(arr[1], pb[perm[::-1]])
    order = rng.permutation(n)[:1700]
    copies = _engine.copy_pairs(pairs, order)
    assert copies == [copy.deepcopy(pairs[i]) for i in order]
    assert all(c[0] is not pairs[i][0] and c[1] is not pairs[i][1] and type(c) is tuple for c, i in zip(copies, order))
    assert type(copies[0][0]) is Feature and copies[0][0].x is pairs[order[0]][0].x     # atomic values are shared, like deepcopy
    copies[0][0].x = -1.0
    assert pairs[order[0]][0].x != -1.0                                                  # ... the instances are not
    with pytest.raises(IndexError):
        _engine.copy_pairs(pairs, np.array([n]))
    with pytest.raises(IndexError):
        _engine.match_pairs(fa, fb, [Match(a_index=n, b_index=0)])
    # shapes outside the helper's fast path: same results through the general code
    fa[3].extra = "tag"             # an extra atomic attribute travels with the copy
    fa[4].x = 7                     # an int coordinate stays an int
    fb[5].y = np.float64(2.5)       # a non-atomic value: deep-copied
    odd = _engine.match_pairs(fa, fb, [Match(a_index=3, b_index=3), Match(a_index=-1, b_index=4),
                                       Match(a_index=np.int64(5), b_index=5), Match(a_index=4, b_index=0)])
    assert odd[0][0] is fa[3] and odd[1][0] is fa[-1] and odd[2][1] is fb[5] and odd[3][0] is fa[4]
    got = _engine.copy_pairs(odd, np.arange(4))
    assert got == [copy.deepcopy(p) for p in odd]
    assert got[0][0].extra == "tag" and isinstance(got[3][0].x, int) and isinstance(got[2][1].y, np.float64)
    np.testing.assert_array_equal(_engine.pair_arrays(odd)[0][3], [7.0, fa[4].y])

    class Point:                    # not a Feature at all
        def __init__(self, x, y):
            self.x, self.y = x, y

    other = [(Point(1.0, 2.0), Point(3.0, 4.0)) for _ in range(3)]
    np.testing.assert_array_equal(_engine.pair_arrays(other)[1], [[3.0, 4.0]] * 3)
    one = _engine.copy_pairs(other, np.array([1]))
    assert one[0][0] is not other[1][0] and one[0][0].x == 1.0
    assert _engine.pair_arrays([]).shape == (2, 0, 2) and _engine.copy_pairs(pairs, np.array([], dtype=np.int64)) == []


def test_gc_paused_restores_the_collector_and_leaves_no_permanent_generation(monkeypatch):
    """_engine.gc_paused: the collector's state is restored; by default the embedding application's generations are left
    alone (a library call must not promote the caller's young objects) and the new objects wait in the
    youngest one; with SFM_GC_SPLICE=1 they are spliced into the oldest generation so that no young collection walks them
    right after; nothing stays frozen either way, cyclic garbage is still collected afterwards, and a process that keeps a
    permanent generation of its own is left alone."""
    import gc
    import weakref

    from lib.common.feature import Feature
    from structure_from_motion_amd.epipolar import _engine

    assert gc.isenabled() and gc.get_freeze_count() == 0
    monkeypatch.delenv("SFM_GC_SPLICE", raising=False)
    calls = []
    real_freeze = gc.freeze
    monkeypatch.setattr(gc, "freeze", lambda: (calls.append(1), real_freeze())[1])
    with _engine.gc_paused():
        made = [(Feature(1.0, 2.0), Feature(3.0, 4.0)) for _ in range(15_000)]
    assert gc.isenabled() and gc.get_freeze_count() == 0 and not calls    # default: the generations were not touched
    del made
    monkeypatch.setenv("SFM_GC_SPLICE", "1")
    with _engine.gc_paused():
        assert not gc.isenabled()
        with _engine.gc_paused():          # nested: the inner exit must not switch the collector back on
            made = [(Feature(1.0, 2.0), Feature(3.0, 4.0)) for _ in range(15_000)]
        assert not gc.isenabled()
    assert gc.isenabled() and gc.get_freeze_count() == 0
    assert gc.get_count()[0] < 5_000      # the 45 000 new objects are not waiting in the youngest generation
    assert len(calls) == 1
    monkeypatch.setattr(gc, "freeze", real_freeze)
    del made

    class Node:
        pass

    a, b = Node(), Node()
    a.other, b.other = b, a
    probe = weakref.ref(a)
    del a, b
    gc.collect()
    assert probe() is None

    keep = [Node() for _ in range(10)]
    gc.freeze()
    try:
        frozen = gc.get_freeze_count()
        with _engine.gc_paused():
            made = [(Feature(1.0, 2.0), Feature(3.0, 4.0)) for _ in range(15_000)]
        assert gc.get_freeze_count() == frozen      # the caller's permanent generation was not released
    finally:
        gc.unfreeze()
    gc.disable()
    try:
        with _engine.gc_paused():
            pass
        assert not gc.isenabled()                    # a collector the caller had off stays off
    finally:
        gc.enable()
    del keep, made
