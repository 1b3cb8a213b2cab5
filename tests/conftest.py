import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def native_lib():
    """Built C-ABI library (compiled on demand; hipcc cross-compiles without a GPU)."""
    from structure_from_motion_amd import _native, build

    build.build_all()
    return _native.load()


class COracle:
    """The plain-C / OpenMP restatement of the scoring loop (oracle/sed_score.c and oracle/fit_eight.c), loaded with
    ctypes.  Test infrastructure: the full-size parity tests feed it every hypothesis of a configuration."""

    def __init__(self):
        import ctypes as C
        import subprocess

        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle")], check=True)
        self.lib = lib = C.CDLL(os.path.join(REPO, "oracle", "libsfm_oracle.so"))
        lib.sfm_oracle_score.restype = C.c_int
        lib.sfm_oracle_score.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.sfm_oracle_sed_values.restype = None
        lib.sfm_oracle_sed_values.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        self.threads = min(len(os.sched_getaffinity(0)), 16)

    def score(self, corr, E, S, thr):
        """(cnt, s1, s2) of every hypothesis: corr [n,4] f64, E [h,9] or [h,3,3] f64, S [h,8] int32."""
        import numpy as np

        corr = np.ascontiguousarray(corr, dtype=np.float64)
        E = np.ascontiguousarray(np.asarray(E, dtype=np.float64).reshape(-1, 9))
        S = np.ascontiguousarray(S, dtype=np.int32)
        h = E.shape[0]
        assert S.shape == (h, 8) and corr.shape[1] == 4
        cnt = np.zeros(h, dtype=np.int32)
        s1 = np.zeros(h)
        s2 = np.zeros(h)
        self.lib.sfm_oracle_score(corr.ctypes.data, corr.shape[0], E.ctypes.data, S.ctypes.data, h, float(thr),
                                  cnt.ctypes.data, s1.ctypes.data, s2.ctypes.data, self.threads)
        return cnt, s1, s2


@pytest.fixture(scope="session")
def c_oracle_lib():
    return COracle()
