"""CPU tests: the Harris oracle (oracle/harris_oracle.py) against the real reference's outputs
(tests/golden/g12_harris.npz) and the reference's own unit vectors (test_correlate.py, test_harris_detector.py)."""
import numpy as np
import pytest

from oracle import harris_oracle as ho


@pytest.mark.parametrize("name", ["rect", "tex", "blobs"])
def test_harris_oracle_matches_reference(golden, name):
    d = golden("g12_harris")
    img = d[f"{name}_image"]
    np.testing.assert_array_equal(ho.cross_correlate(img, ho.SOBEL_X), d[f"{name}_sobel_x"])
    np.testing.assert_array_equal(ho.cross_correlate(img, ho.SOBEL_X.T), d[f"{name}_sobel_y"])
    corn = ho.cornerness_image(img)
    ref = d[f"{name}_cornerness"]
    assert corn.shape == ref.shape
    # np.linalg.det (LU + exp/log) vs the plain 2x2 determinant: rounding level, relative to the terms' size
    scale = np.maximum(np.abs(ref), 1.0)
    assert np.max(np.abs(corn - ref) / scale) <= 1e-9
    pts, supp = ho.detect_harris_corners(img, int(d[f"{name}_n"]))
    np.testing.assert_array_equal(supp != 0, d[f"{name}_suppressed"] != 0)
    np.testing.assert_array_equal(pts, d[f"{name}_corners"])


def test_reference_unit_vectors(golden):
    d = golden("g12_harris")
    np.testing.assert_array_equal(ho.cross_correlate(d["cc_image"], d["cc_kernel"]), d["cc_out"])      # test_correlate.py
    out = ho.cross_correlate(np.ones((5, 10)), np.ones((3, 3)))
    assert np.allclose(out[1:-1, 1:-1], 9) and np.allclose(out[0], 0) and np.allclose(out[:, 0], 0)
    np.testing.assert_array_equal(ho.cross_correlate(np.ones((5, 10)), np.ones((5, 5))), d["cc_ones5"])
    for bad in (np.ones((3, 2)), np.ones((2, 2))):
        with pytest.raises(ValueError):
            ho.cross_correlate(np.ones((5, 5)), bad)
    with pytest.raises(ValueError):
        ho.cross_correlate(np.ones((2, 5)), np.ones((3, 3)))
    # test_harris_detector.py: the four corners of a filled rectangle, in this order, within one pixel
    pts, _ = ho.detect_harris_corners(d["rect_image"])
    # (the order among the four exactly tied corners is pinned by the golden vector of the real reference)
    expected = {(75, 50), (75, 150), (25, 50), (25, 150)}
    assert len(pts) == 4
    for x, y in pts:
        assert any(np.allclose((ey, ex), (y, x), atol=1.0) for ey, ex in expected)
    with pytest.raises(ValueError):
        ho.detect_harris_corners(d["rect_image"], 0)


def test_in_place_suppression_is_order_dependent():
    """The raster-order, in-place semantics differ from a plain 3x3 local-maximum test."""
    img = np.array([[0.0, 0.0, 0.0, 0.0], [0.0, 1.0, 2.0, 3.0], [0.0, 0.0, 0.0, 0.0]])
    seq = img.copy()
    ho.non_max_suppress(seq)
    assert seq[1].tolist() == [0.0, 0.0, 0.0, 3.0]
    img2 = np.array([[3.0, 0.0, 0.0], [0.0, 2.0, 0.0], [0.0, 0.0, 1.0]])  # 2 is killed by 3; 1 then survives
    seq2 = img2.copy()
    ho.non_max_suppress(seq2)
    assert seq2[2, 2] == 1.0 and seq2[1, 1] == 0.0 and seq2[0, 0] == 3.0
