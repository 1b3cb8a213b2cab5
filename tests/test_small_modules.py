"""CPU tests of the two small host modules (reference lib/blur/tests, lib/data_utils/tests)."""
from pathlib import Path

import numpy as np
import pytest

from lib.blur import gaussian
from lib.data_utils.middlebury_utils import load_camera_k_r_t
from lib.transforms.transforms import Transform3D


def test_create_gaussian_kernel():
    """reference lib/blur/tests/test_gaussian.py."""
    size = 5
    kernel = gaussian.create_gaussian_kernel(size, 1.0)
    assert kernel.shape == (size, size)
    assert abs(np.sum(kernel) - 1.0) < 1e-12
    half = size // 2
    for i in range(half + 1):
        for j in range(half + 1):
            if i < half:
                assert kernel[i, j] < kernel[i + 1, j]
            if j < half:
                assert kernel[i, j] < kernel[i, j + 1]
            assert kernel[i, j] == kernel[size - i - 1, j] == kernel[i, size - j - 1]
    with pytest.raises(ValueError):
        gaussian.create_gaussian_kernel(4, 1.0)
    with pytest.raises(ValueError):
        gaussian.create_gaussian_kernel(1, 1.0)


def test_load_camera_intrinsics(tmp_path: Path):
    """reference lib/data_utils/tests/test_middlebury_utils.py (same file layout, written here)."""
    values = " ".join(str(v) for v in list(range(1, 10)) + list(range(1, 10)) + [1, 2, 3])
    other = " ".join(["0"] * 21)
    par = tmp_path / "test_par.txt"
    par.write_text(f"2\nimg0000.png {other}\nimg0001.png {values}\n")
    intrinsics, transform = load_camera_k_r_t(par, 1)
    expected = Transform3D.from_rmat_t(np.array(range(1, 10)).reshape(3, 3), np.array(range(1, 4)).reshape((3, 1)))
    np.testing.assert_almost_equal(np.array(range(1, 10)).reshape((3, 3)), intrinsics)
    np.testing.assert_almost_equal(expected.Tmat, transform.Tmat)
    with pytest.raises(ValueError):
        load_camera_k_r_t(par, 5)
    par.write_text("3\nimg0000.png " + other + "\n")
    with pytest.raises(ValueError):
        load_camera_k_r_t(par, 2)
    par.write_text("3\nnot_an_image.txt " + other + "\n")
    with pytest.raises(RuntimeError):
        load_camera_k_r_t(par, 1)
