"""ctypes binding of csrc/libsfm_hip.so (the C ABI of include/sfm_hip.h).

Loading never needs a GPU (so symbol checks run anywhere); calling a kernel entry point does.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsfm_hip.so")

SFM_OK = 0
AGG_SUM, AGG_SQUARE, AGG_MEAN, AGG_RMS = 0, 1, 2, 3
FIT_DEGENERATE = 1
MATCH_NCC, MATCH_SSD = 0, 1
PATCH_PLAIN, PATCH_MEAN_REMOVED, PATCH_RAW64 = 0, 1, 2
INT64_MAX = (1 << 63) - 1
ABI_VERSION = 12


def match_ssd_int(bits: int, signed: bool) -> int:
    """SFM_MATCH_SSD_INT(bits, is_signed): SSD of integer images in the image dtype's modular arithmetic (ssd.py:31-36)."""
    assert bits in (8, 16, 32, 64)
    return 0x100 | (0x80 if signed else 0) | bits


class SelectResult(C.Structure):
    """struct sfm_select_result (40 bytes)."""

    _fields_ = [
        ("key", C.c_uint64),
        ("best_h", C.c_int64),
        ("best_err", C.c_double),
        ("first_flagged", C.c_int64),
        ("n_flagged", C.c_int32),
        ("best_cnt", C.c_int32),
    ]


class ScoreOptions(C.Structure):
    """struct sfm_score_options (48 bytes): launch options of the two-tier scoring kernels — they change how a call is
    launched, never its counts or decisions (include/sfm_hip.h) — and the call's measurement hook: ``timing=(before, after)``,
    two ``torch.cuda.Event`` (recorded once already: torch creates the hipEvent lazily) or raw hipEvent_t handles that the
    library records immediately around the scoring kernel of THIS call."""

    _fields_ = [
        ("kernel", C.c_int32),         # SCORE_KERNEL_AUTO / _FILTERED / _MATRIX
        ("hyps_per_wave", C.c_int32),  # VALU-filter kernel: 0 = by launch size, 1 / 2 / 4
        ("split", C.c_int32),          # ranges of the points: -1 = by launch size, 0 = none, k
        ("order", C.c_int32),          # heaviest-first order: -1 = by launch size, 0 / 1
        ("one_sided", C.c_int32),      # VALU filter: -1 / 1 one-sided (default), 0 two-sided
        ("xcd_map", C.c_int32),        # batches: -1 / 1 XCD-aware block map (default), 0 plain grid
        ("block_sync", C.c_int32),     # small pass: -1 = by size, k = barrier every k iterations, 0 never
        ("persistent", C.c_int32),     # matrix-pipe kernel, one pair: 1 persistent waves, -1 / 0 one block per 4 items (default)
        ("timing_before", C.c_void_p),  # hipEvent_t recorded right before the scoring kernel of this call (NULL: none)
        ("timing_after", C.c_void_p),   # ... and right after it
    ]

    def __init__(self, kernel=0, hyps_per_wave=0, split=-1, order=-1, one_sided=-1, xcd_map=-1, block_sync=-1, persistent=-1,
                 timing=None):
        if isinstance(kernel, str):
            kernel = {"auto": SCORE_KERNEL_AUTO, "filtered": SCORE_KERNEL_FILTERED, "matrix": SCORE_KERNEL_MATRIX}[kernel]
        before, after = timing if timing is not None else (None, None)
        handle = lambda e: None if e is None else (e if isinstance(e, int) else e.cuda_event)  # noqa: E731
        super().__init__(int(kernel), int(hyps_per_wave), int(split), int(order), int(one_sided), int(xcd_map),
                         int(block_sync), int(persistent), handle(before), handle(after))

    def with_timing(self, before, after) -> "ScoreOptions":
        """A copy of these options that has the library record ``before`` / ``after`` around the scoring kernel."""
        return ScoreOptions(self.kernel, self.hyps_per_wave, self.split, self.order, self.one_sided, self.xcd_map,
                            self.block_sync, self.persistent, timing=(before, after))


SCORE_KERNEL_AUTO, SCORE_KERNEL_FILTERED, SCORE_KERNEL_MATRIX = 0, 1, 2


def score_options_from_env(environ=None) -> "ScoreOptions":
    """The SFM_SCORE_* variables as a ScoreOptions (unset / unparsable = the library's default for that field).  The
    package calls this ONCE, when the library is loaded, and hands the result to sfm_score_set_default_options: the
    library itself never reads the environment."""
    env = os.environ if environ is None else environ

    def number(name, default):
        try:
            return int(env[name])
        except (KeyError, ValueError):
            return default

    matrix = number("SFM_SCORE_MATRIX", -1)
    hpw = number("SFM_SCORE_HPW", 0)
    return ScoreOptions(
        kernel=SCORE_KERNEL_AUTO if matrix < 0 else (SCORE_KERNEL_MATRIX if matrix > 0 else SCORE_KERNEL_FILTERED),
        hyps_per_wave=hpw if hpw in (1, 2, 4) else 0,
        split=max(-1, number("SFM_SCORE_SPLIT", -1)),
        order=min(1, max(-1, number("SFM_SCORE_ORDER", -1))),
        one_sided=min(1, max(-1, number("SFM_SCORE_ONE_SIDED", -1))),
        xcd_map=min(1, max(-1, number("SFM_SCORE_XCD", -1))),
        block_sync=max(-1, number("SFM_SCORE_SYNC", -1)),
        persistent=min(1, max(-1, number("SFM_SCORE_PERSISTENT", -1))),
    )


_P = C.c_void_p
_I64 = C.c_int64
_U64 = C.c_uint64
_D = C.c_double

# name -> argtypes; every function returns int.  Must list every symbol of include/sfm_hip.h.
SIGNATURES = {
    "sfm_normalize_correspondences": [_P, _P, _I64, _D, _D, _D, _D, _P, _P],
    "sfm_sample_philox": [_U64, _U64, _I64, _I64, _I64, _I64, _P, _P],
    "sfm_refine_inliers": [_P, _I64, _I64, _P, _P, _P, _D, C.c_int, C.c_int, _P, _P, _P, _P],
    "sfm_sample_fit_philox": [_U64, _P, _U64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _P],
    "sfm_sample_philox_dev": [_P, _U64, _I64, _I64, _I64, _I64, _P, _P],
    "sfm_sample_philox_at": [_U64, _U64, _P, _I64, _I64, _P, _P],
    "sfm_fit_eight_point": [_P, _I64, _P, _I64, _I64, _P, _P, _P, _P],
    "sfm_fit_eight_point_traced": [_P, _I64, _P, _I64, _I64, _P, _P, _P, _P],
    "sfm_fit_stage": [C.c_int, _P, _P, _P],
    "sfm_hartley_normalize": [_P, _I64, _P, _P],
    "sfm_score_sed": [_P, _I64, _P, _P, _I64, _I64, _D, _P, _P, _P, _P, _I64, _P],
    "sfm_score_sed_ex": [_P, _I64, _P, _P, _I64, _I64, _D, _P, _P, _P, _P, _I64, _P, _P],
    "sfm_score_set_default_options": [_P],
    "sfm_score_get_default_options": [_P],
    "sfm_score_kernel_choice_ex": [_I64, _I64, _I64, _P],
    "sfm_debug_matrix_filter": [_P, _I64, _P, _I64, _D, _P, _I64, _P, _P, _P, _P],
    "sfm_ransac_pass_small": [_U64, _P, C.c_int, _I64, _P, _I64, _I64, _D, _D, C.c_int, _I64, _P, _P, _P, _P, _P, _P,
                              _P, _P, _P, _I64, _P, _P],
    "sfm_ransac_pass_large": [_U64, _P, C.c_int, _I64, _P, _I64, _I64, _D, _D, C.c_int, _I64, _P, _P, _P, _P, _P, _P,
                              _P, _P, _P, _I64, _P, _P],
    "sfm_ransac_pass_batch": [_U64, _P, _U64, C.c_int, _I64, _P, _I64, _I64, _I64, _D, _D, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                              _I64, _P, _P],
    "sfm_select_best": [_P, _P, _P, _P, _I64, _I64, _D, C.c_int, _I64, _P, _P],
    "sfm_fold_select_records": [_P, _I64, _I64, _P, _P, _P, _P],
    "sfm_fold_select_records_host": [_P, _I64, _I64, _P, _P, _P],
    "sfm_inlier_mask": [_P, _I64, _P, _P, _I64, _I64, _P, _D, _P, _P],
    "sfm_sed_values": [_P, _I64, _P, _P, _P],
    "sfm_cheirality": [_P, _I64, _P, _I64, _D, _P, _P],
    "sfm_triangulate": [_P, _I64, _P, _P, _P, _P],
    "sfm_decompose_essential": [_P, _I64, _P, _P, _P],
    "sfm_cheirality_batched": [_P, _I64, _I64, _P, _P, _D, _P, _P],
    "sfm_pose_vote": [_P, _I64, _I64, _P, _P, _P, _P],
    "sfm_triangulate_selected": [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P],
    "sfm_patch_extract": [_P, _I64, _I64, _P, _I64, C.c_int, C.c_int, _I64, _P, _P, _P, _P],
    "sfm_pair_scores": [C.c_int, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I64, C.c_int, _P, _P],
    "sfm_match_row_summary": [_P, _I64, _I64, _P, _P, _P, _P],
    "sfm_match_summary": [C.c_int, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I64, C.c_int, _P, _I64, _P, _P, _P, _P],
    "sfm_cross_correlate": [_P, _I64, _I64, _P, C.c_int, _P, _P],
    "sfm_harris_cornerness": [_P, _P, _I64, _I64, C.c_int, _D, C.c_int, _I64, _I64, _P, _P],
    "sfm_nms_inplace": [_P, _I64, _I64, _P],
    "sfm_nms_round": [_P, _P, _I64, _I64, _P, _P],
    "sfm_nms_finalize": [_P, _P, _I64, _I64, _P],
    "sfm_compact_nonzero": [_P, _I64, C.c_int32, _P, _P, _P, _P],
    "sfm_prune_top": [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, _P],
    "sfm_pyshuffle_table": [_P, _P, _I64, _I64, _P, _P, _I64, _P],
    "sfm_score_kernel_choice": [_I64, _I64, _I64],
}
OTHER_SYMBOLS = ["sfm_last_error", "sfm_abi_version", "sfm_score_workspace_bytes", "sfm_score_workspace_bytes_ex",
                 "sfm_fit_trace_doubles", "sfm_match_summary_workspace_bytes"]

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libsfm_hip.so (built by structure_from_motion_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m structure_from_motion_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.sfm_last_error.restype = C.c_char_p
    lib.sfm_last_error.argtypes = []
    lib.sfm_abi_version.restype = C.c_int
    lib.sfm_abi_version.argtypes = []
    lib.sfm_fit_trace_doubles.restype = C.c_int
    lib.sfm_fit_trace_doubles.argtypes = []
    lib.sfm_score_workspace_bytes.restype = C.c_int64
    lib.sfm_score_workspace_bytes.argtypes = [_I64, _I64, _I64]
    lib.sfm_score_workspace_bytes_ex.restype = C.c_int64
    lib.sfm_score_workspace_bytes_ex.argtypes = [_I64, _I64, _I64, _P]
    lib.sfm_match_summary_workspace_bytes.restype = C.c_int64
    lib.sfm_match_summary_workspace_bytes.argtypes = [_I64, _I64]
    if lib.sfm_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"libsfm_hip.so ABI {lib.sfm_abi_version()} != expected {ABI_VERSION}; rebuild it")
    # the SFM_SCORE_* variables, translated once per process (the library does not read the environment)
    options = score_options_from_env()
    if lib.sfm_score_set_default_options(C.byref(options)) != SFM_OK:
        raise NativeLibraryError("SFM_SCORE_* environment: " + lib.sfm_last_error().decode("utf-8", "replace"))
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != SFM_OK:
        msg = load().sfm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({status}): {msg}")


_hostfast = False   # not looked for yet


def hostfast():
    """The CPython helper module for the bulk conversions of the drop-in boundary (csrc/hostfast.c), or None when it has
    not been built — callers then run the equivalent pure-Python code (same results, several times slower at 50 000
    matches).  Host code only: nothing numeric depends on it."""
    global _hostfast
    if _hostfast is False:
        import importlib.util

        path = os.path.join(HERE, "csrc", "_sfm_hostfast.so")
        _hostfast = None
        if os.path.exists(path):
            try:
                spec = importlib.util.spec_from_file_location("_sfm_hostfast", path)
                module = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(module)
                _hostfast = module
            except ImportError:
                _hostfast = None
    return _hostfast
