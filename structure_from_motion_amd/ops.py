"""``torch.ops.sfm_hip.*`` — the hot-path kernels as PyTorch-ROCm custom ops (csrc/sfm_torch_ops.cpp, a
``TORCH_LIBRARY`` above the C ABI of include/sfm_hip.h).

Op set: ``normalize_coords, sample_philox, fit_eight_point, score_sed, select_best, inlier_mask,
cheirality, triangulate`` — functional forms with Meta kernels (fake tensors, ``torch.compile``,
``torch.library.opcheck``) — and the in-place ``*_`` forms the pre-allocated engine uses.  ``device.py`` dispatches
through them; ``load()`` must have run before ``torch.ops.sfm_hip`` is touched.
"""
from __future__ import annotations

import os

import torch

from . import _native

HERE = os.path.dirname(os.path.abspath(__file__))
OPS_LIB_PATH = os.path.join(HERE, "csrc", "libsfm_torch_ops.so")

_loaded = False


def load():
    """Load libsfm_hip.so and register the ``sfm_hip`` op library with torch (idempotent).  Raises if either
    library is missing: there is no Python or CPU fallback for these ops."""
    global _loaded
    if not _loaded:
        _native.load()
        if not os.path.exists(OPS_LIB_PATH):
            raise _native.NativeLibraryError(
                f"{OPS_LIB_PATH} is missing: build it with `python -m structure_from_motion_amd.build`")
        import ctypes

        ops_abi = ctypes.CDLL(OPS_LIB_PATH).sfm_torch_ops_abi_version()   # the C-ABI version it was compiled against
        if ops_abi != _native.ABI_VERSION:
            raise _native.NativeLibraryError(
                f"libsfm_torch_ops.so was built against C-ABI {ops_abi}, libsfm_hip.so is {_native.ABI_VERSION}; "
                f"rebuild both with `python -m structure_from_motion_amd.build --force`")
        torch.ops.load_library(OPS_LIB_PATH)
        _loaded = True
    return torch.ops.sfm_hip


FUNCTIONAL_OPS = ("normalize_coords", "sample_philox", "fit_eight_point", "score_sed", "select_best", "inlier_mask",
                  "cheirality", "triangulate")
INPLACE_OPS = ("normalize_coords_", "fit_eight_point_", "sample_fit_philox_", "score_sed_", "select_best_",
               "inlier_mask_", "ransac_pass_small_")
