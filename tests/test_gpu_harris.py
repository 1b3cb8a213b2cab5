"""GPU tests of the Harris detector stencils: bit-exact against the oracle, corner lists identical to the
real reference (tests/golden/g12_harris.npz)."""
import numpy as np
import pytest

from oracle import harris_oracle as ho

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _gpu(native_lib):
    from structure_from_motion_amd import device

    device.require_gpu()


from lib.common import correlate  # noqa: E402
from lib.harris import harris_detector as harris  # noqa: E402


@pytest.mark.parametrize("name", ["rect", "tex", "blobs"])
def test_detector_equals_reference(golden, name):
    d = golden("g12_harris")
    img = d[f"{name}_image"]
    np.testing.assert_array_equal(harris._apply_sobel_x(img), d[f"{name}_sobel_x"])       # bit-exact vs the reference
    np.testing.assert_array_equal(harris._apply_sobel_y(img), d[f"{name}_sobel_y"])
    corn = harris._calculate_cornerness_image(img, 2, 0.04)
    np.testing.assert_array_equal(corn, ho.cornerness_image(img))                            # bit-exact vs the oracle
    ref = d[f"{name}_cornerness"]
    assert np.max(np.abs(corn - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-9                 # vs np.linalg.det route
    supp = corn.copy()
    supp[supp < 0] = 0.0
    harris._non_max_suppress(supp)
    np.testing.assert_array_equal(supp != 0, d[f"{name}_suppressed"] != 0)
    corners = harris.detect_harris_corners(img, num_corners=int(d[f"{name}_n"]))
    np.testing.assert_array_equal(np.array([[c.x, c.y] for c in corners]).reshape(-1, 2), d[f"{name}_corners"])


def test_detect_harris_corners_rectangle(golden):
    """reference test_harris_detector.py:13-32 (rectangle drawn without OpenCV)."""
    image = golden("g12_harris")["rect_image"]
    corner_coordinates = harris.detect_harris_corners(image)
    # the order among the four exactly tied corners is pinned by the golden vector (test_detector_equals_reference)
    expected_corners = [(75, 50), (75, 150), (25, 50), (25, 150)]
    assert len(corner_coordinates) == 4
    for corner in corner_coordinates:
        assert any(np.allclose(e, (corner.y, corner.x), atol=1.0) for e in expected_corners)
    with pytest.raises(ValueError):
        harris.detect_harris_corners(image, num_corners=0)


def test_cross_correlate(golden):
    """reference test_correlate.py."""
    d = golden("g12_harris")
    input_image = np.ones((5, 10), dtype=float)
    output = correlate.cross_correlate(input_image, np.ones((3, 3)))
    assert np.allclose(output[1:-1, 1:-1], 9) and np.allclose(output[0], 0) and np.allclose(output[:, -1], 0)
    output = correlate.cross_correlate(input_image, np.ones((5, 5)))
    assert np.allclose(output[2:-2, 2:-2], 25)
    np.testing.assert_array_equal(output, d["cc_ones5"])
    np.testing.assert_array_equal(correlate.cross_correlate(d["cc_image"], d["cc_kernel"]), d["cc_out"])
    assert correlate.cross_correlate(d["cc_image"], d["cc_kernel"])[1, 1] == \
        1 * 1 + 5 * -2 + 4 * 3 + 2 * 2 + 5 * 1 + 7 * 0 + 9 * 7 + -5 * -5 + 4 * 1
    with pytest.raises(ValueError):
        correlate.cross_correlate(np.ones((5, 5)), np.ones((2, 2)))
    with pytest.raises(ValueError):
        correlate.cross_correlate(np.ones((2, 5)), np.ones((3, 3)))
    with pytest.raises(ValueError):
        correlate.cross_correlate(np.ones((5, 5, 3)), np.ones((3, 3)))


@pytest.mark.parametrize("shape,seed", [((3, 3), 0), ((7, 300), 1), ((121, 97), 2), ((480, 640), 3)])
def test_nms_and_detector_vs_oracle_random(shape, seed):
    """In-place raster-order suppression (wavefront kernel) on random images with many ties, and the whole
    detector, against the sequential oracle."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 6, size=shape).astype(np.float64)       # few levels -> plenty of ties and plateaus
    want = img.copy()
    ho.non_max_suppress(want)
    got = img.copy()
    harris._non_max_suppress(got)
    np.testing.assert_array_equal(got, want)
    if min(shape) >= 8:
        photo = rng.integers(0, 256, size=shape).astype(np.uint8)
        pts, supp = ho.detect_harris_corners(photo, 200)
        corners = harris.detect_harris_corners(photo, 200)
        np.testing.assert_array_equal(np.array([[c.x, c.y] for c in corners]).reshape(-1, 2), pts)
        for bs, k in ((3, 0.06), (4, 0.04)):
            np.testing.assert_array_equal(harris._calculate_cornerness_image(photo, bs, k), ho.cornerness_image(photo, bs, k))


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.int16, np.int32, np.float32, np.uint16, np.int64, np.float64])
def test_detector_image_dtypes(dtype):
    """An image of any dtype gives the corners of its float64 copy (narrow types are widened on the device, the others on the
    host: both exact) — and those of the oracle, which converts like the reference (harris_detector.py:56 works on the array as
    it comes; its Sobel correlation accumulates in float64)."""
    rng = np.random.default_rng(11)
    info = np.iinfo(dtype) if np.issubdtype(dtype, np.integer) else None
    if info is not None:
        img = rng.integers(max(info.min, -3000), min(info.max, 3000) + 1, size=(90, 120)).astype(dtype)
    else:
        img = (rng.random((90, 120)) * 255).astype(dtype)
    view = img[:, ::-1]                                      # not contiguous
    swapped = img.astype(img.dtype.newbyteorder())           # same values, foreign byte order (a no-op for one-byte types)
    for im in (img, view, swapped):
        got = harris.detect_harris_corners(im, 80)
        wide = harris.detect_harris_corners(np.ascontiguousarray(im, dtype=np.float64), 80)
        assert [(c.x, c.y) for c in got] == [(c.x, c.y) for c in wide]
        pts, _ = ho.detect_harris_corners(np.ascontiguousarray(im, dtype=np.float64), 80)
        np.testing.assert_array_equal(np.array([[c.x, c.y] for c in got]).reshape(-1, 2), pts)


def test_nms_fixpoint_equals_wavefront_and_oracle():
    """Both device formulations of the in-place raster-order suppression agree with the sequential oracle,
    including adversarial monotone ramps (dependency chains as long as the image) that force the wavefront fallback."""
    import torch
    from structure_from_motion_amd import _native, device
    from structure_from_motion_amd._native import check

    lib = _native.load()
    rng = np.random.default_rng(5)
    cases = [rng.integers(0, 4, (37, 53)).astype(float), rng.random((64, 200)),
             np.add.outer(np.arange(40.0), np.arange(150.0))[::-1, ::-1].copy(),      # increasing towards the origin
             np.add.outer(np.arange(40.0), np.arange(150.0)), np.zeros((5, 7)), np.ones((9, 9)),
             # widths around the four pixels a thread of the round kernel owns, single rows and columns, NaN and inf pixels
             rng.integers(0, 5, (6, 1)).astype(float), rng.integers(0, 5, (1, 6)).astype(float), rng.integers(0, 5, (9, 2)).astype(float),
             rng.integers(0, 5, (11, 3)).astype(float), rng.integers(0, 5, (13, 5)).astype(float), rng.integers(0, 9, (33, 1023)).astype(float),
             np.where(rng.random((30, 41)) < 0.05, np.nan, rng.integers(0, 5, (30, 41)).astype(float)),
             np.where(rng.random((30, 41)) < 0.05, np.inf, rng.integers(0, 5, (30, 41)).astype(float))]
    for img in cases:
        want = img.copy()
        ho.non_max_suppress(want)
        a = img.copy()
        harris._non_max_suppress(a)                       # fixpoint path (with its built-in fallback)
        np.testing.assert_array_equal(a, want)
        t = device.to_device(img)
        check(lib.sfm_nms_inplace(t.data_ptr(), t.shape[0], t.shape[1], device._stream()), "sfm_nms_inplace")
        np.testing.assert_array_equal(t.cpu().numpy(), want)   # wavefront kernel on its own


def test_compact_nonzero_and_topk_equal_full_image_path():
    """The device compaction feeds the same top-k as the full-image NumPy selection: random sparse images, all-zero,
    exact ties among the leaders (fallback), more survivors than the compaction buffer (fallback), NaN."""
    from structure_from_motion_amd import device
    from structure_from_motion_amd.harris import harris_detector as hd

    rng = np.random.default_rng(3)
    cases = []
    sparse = np.zeros((97, 131))
    hits = rng.choice(sparse.size, 900, replace=False)
    sparse.ravel()[hits] = rng.random(900) + 0.1
    cases.append(sparse)
    cases.append(np.zeros((40, 50)))
    tied = sparse.copy()
    tied.ravel()[hits[:5]] = 9.0  # five equal leaders
    cases.append(tied)
    cases.append(rng.random((300, 400)) + 0.5)  # 120 000 non-zeros > capacity
    withnan = sparse.copy()
    withnan[3, 3] = np.nan
    cases.append(withnan)
    one = np.zeros((5, 7))
    one[2, 3] = 4.0
    cases.append(one)
    for img in cases:
        t = device.to_device(img)
        for k in (1, 10, 600, 5000):
            got = hd._strongest_indices_device(t, k)
            want = hd._strongest_indices(img, k)
            np.testing.assert_array_equal(got, want)


def test_prune_top_keeps_every_leader():
    """sfm_prune_top: whatever the values — negative, subnormal, huge, equal, NaN of either sign — every candidate at or above
    the m-th largest is kept, nothing is invented, and with fewer than m candidates all of them are.  Then the detector's
    selection on images whose maxima crowd one key (all equal, or all within a factor 1.06 of each other: more than the
    pruning buffer) and on one whose leaders straddle a key boundary."""
    import ctypes as C

    from structure_from_motion_amd import _native, device
    from structure_from_motion_amd._native import check
    from structure_from_motion_amd.harris import harris_detector as hd

    lib = _native.load()
    rng = np.random.default_rng(21)

    def prune(values, m, capacity_out=1 << 15, found=None):
        n = len(values)
        v = device.to_device(np.asarray(values, dtype=np.float64))
        idx = device.to_device(np.arange(n, dtype=np.int32) * 3 + 1, dtype=torch.int32)
        cnt = device.to_device(np.array([n if found is None else found, 0], dtype=np.int32), dtype=torch.int32)
        ws = torch.empty((8192,), dtype=torch.int32, device=v.device)
        out_i = torch.empty((max(capacity_out, 1),), dtype=torch.int32, device=v.device)
        out_v = torch.empty((max(capacity_out, 1),), dtype=torch.float64, device=v.device)
        check(lib.sfm_prune_top(v.data_ptr(), idx.data_ptr(), cnt[0:].data_ptr(), n, m, ws.data_ptr(), capacity_out, cnt[1:].data_ptr(),
                                out_i.data_ptr(), out_v.data_ptr(), device._stream()), "sfm_prune_top")
        kept = int(cnt.cpu()[1])
        return kept, out_i[:min(kept, capacity_out)].cpu().numpy(), out_v[:min(kept, capacity_out)].cpu().numpy()

    neg_nan = np.frombuffer(np.array([0xFFF8000000000001], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    sets = [rng.random(5000), rng.standard_normal(5000) * 1e6, np.exp(rng.uniform(-700, 700, 5000)) * rng.choice([-1, 1], 5000),
            np.concatenate([rng.random(3000), np.full(40, 2.5), np.full(40, 2.5000000001)]), np.full(3000, 7.0),
            np.concatenate([rng.random(2000) * 1e-310, [0.0, -0.0]]), np.concatenate([rng.random(999), [np.nan]]),
            np.concatenate([rng.random(999), [neg_nan, np.inf, -np.inf]]), np.array([3.0]), rng.integers(0, 40, 20000).astype(float)]
    for values in sets:
        for m in (1, 2, 51, 601, 100000):
            kept, gi, gv = prune(values, m)
            order = np.sort(values[~np.isnan(values)])[::-1]
            nans = int(np.isnan(values).sum())            # a NaN counts as the largest value there is (the host then takes the literal path)
            floor = np.inf if m <= nans else (order[m - nans - 1] if m - nans <= len(order) else -np.inf)
            must = np.flatnonzero(np.isnan(values) | ((values >= floor) & (m > nans)))
            assert kept == len(gi) and len(set(gi.tolist())) == kept                      # nothing twice
            np.testing.assert_array_equal(values[(gi - 1) // 3].view(np.uint64), gv.view(np.uint64))   # pairs intact, bit for bit
            assert set((must * 3 + 1).tolist()) <= set(gi.tolist()), (m, len(must), kept)
            if m > len(values):
                assert kept == len(values)
    # the counter says how many would be kept even when they do not fit; a device-side `found` below the capacity bounds the scan
    kept, gi, gv = prune(np.full(3000, 7.0), 10, capacity_out=100)
    assert kept == 3000 and len(gi) == 100 and np.all(gv == 7.0)
    kept, gi, _ = prune(np.arange(1000.0), 5, found=10)
    assert kept >= 5 and set(gi.tolist()) <= {3 * i + 1 for i in range(10)} and {3 * i + 1 for i in range(5, 10)} <= set(gi.tolist())
    kept, gi, _ = prune(np.arange(10.0), 5, found=0)
    assert kept == 0
    # argument checks
    assert lib.sfm_prune_top(None, None, None, 4, 1, None, 4, None, None, None, None) == -1   # SFM_EINVAL
    # the detector's selection through the pruning
    flat = np.zeros((200, 300))
    flat[::2, ::2] = 5.0                                   # 15 000 equal maxima: one key, ties among the leaders
    crowded = np.zeros((300, 400))
    crowded[::2, ::2] = 1.0 + rng.random((150, 200)) * 2e-4   # 30 000 maxima inside one key's range (2^-12): more than the pruning buffer
    straddle = np.zeros((64, 64))
    straddle.ravel()[rng.choice(4096, 700, replace=False)] = np.concatenate([np.linspace(1.9, 2.1, 350), rng.random(350)])
    for img in (flat, crowded, straddle):
        t = device.to_device(img)
        for k in (1, 10, 349, 351, 600, 20000):
            np.testing.assert_array_equal(hd._strongest_indices_device(t, k), hd._strongest_indices(img, k))
