"""Device plumbing: torch ROCm tensors for memory and streams; the kernels are reached as PyTorch-ROCm custom ops
(``torch.ops.sfm_hip.*``, csrc/sfm_torch_ops.cpp — the op set of SURVEY.md §8b) and, for the entry points outside
that set, by ctypes calls into the same C ABI (libsfm_hip.so) underneath.

Nothing here computes on the CPU.  Every wrapper only enqueues work on the current torch HIP stream;
host synchronisation happens where a caller reads a result back (``.cpu()`` / ``read_select``).
"""
from __future__ import annotations

import ctypes as C
import os
import random as _pyrandom
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _native, ops
from ._native import ScoreOptions, SelectResult, check  # noqa: F401  (ScoreOptions: re-exported for callers and tests)

F64 = torch.float64
SELECT_BYTES = C.sizeof(SelectResult)
assert SELECT_BYTES == 40


def require_gpu() -> torch.device:
    """The hot path has no CPU fallback: fail loudly without a ROCm device or the HIP library."""
    ops.load()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "structure_from_motion_amd: no ROCm GPU visible; the RANSAC / triangulation hot path "
            "only runs as HIP kernels on MI355X (there is no CPU fallback)."
        )
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device wrappers need contiguous ROCm tensors"
    return t.data_ptr()


def _as_int64(seed: int) -> int:
    """A 64-bit seed as the signed value torch op schemas carry (same bit pattern)."""
    seed &= 2**64 - 1
    return seed - 2**64 if seed >= 2**63 else seed


def to_device(array, dtype=F64) -> torch.Tensor:
    dev = require_gpu()
    return torch.as_tensor(np.ascontiguousarray(array), dtype=dtype).to(dev)


# ------------------------------------------------------------------------------------------------------
# thin wrappers, one per C-ABI entry point
# ------------------------------------------------------------------------------------------------------
def normalize_correspondences(pix_a: torch.Tensor, pix_b: torch.Tensor, K, out=None) -> torch.Tensor:
    """pix_a, pix_b: [..., 2] f64 on device -> corr [..., 4]."""
    op = ops.load()
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    if out is None:
        return op.normalize_coords(pix_a, pix_b, fx, fy, cx, cy)
    op.normalize_coords_(pix_a, pix_b, fx, fy, cx, cy, out)
    return out


def sample_philox(seed: int, h_begin: int, h_count: int, n: int, batch: int = 1, seed_stride: int = 1,
                  out=None, device=None) -> torch.Tensor:
    if out is None:
        return ops.load().sample_philox(_as_int64(seed), seed_stride, h_begin, h_count, n, batch,
                                        device or require_gpu())
    lib = _native.load()  # filling a caller's buffer: straight through the C ABI
    check(lib.sfm_sample_philox(seed & (2**64 - 1), seed_stride, h_begin, h_count, n, batch, _ptr(out),
                                _stream()), "sfm_sample_philox")
    return out


def sample_philox_dev(seed_dev: torch.Tensor, h_begin: int, h_count: int, n: int, batch: int = 1,
                      seed_stride: int = 1, out=None) -> torch.Tensor:
    """Philox sampler whose seed is the int64 word ``seed_dev[0]`` in device memory (graph-replayable)."""
    lib = _native.load()
    if out is None:
        out = torch.empty((batch, h_count, 8), dtype=torch.int32, device=seed_dev.device)
    check(lib.sfm_sample_philox_dev(_ptr(seed_dev), seed_stride, h_begin, h_count, n, batch, _ptr(out),
                                    _stream()), "sfm_sample_philox_dev")
    return out


def sample_philox_at(seed: int, h_index: torch.Tensor, n: int, seed_stride: int = 1, out=None) -> torch.Tensor:
    """h_index: int64 [B] on device -> S [B,1,8]: the sample of hypothesis h_index[b] (no host sync)."""
    lib = _native.load()
    B = h_index.shape[0]
    if out is None:
        out = torch.empty((B, 1, 8), dtype=torch.int32, device=h_index.device)
    check(lib.sfm_sample_philox_at(seed & (2**64 - 1), seed_stride, _ptr(h_index), n, B, _ptr(out),
                                   _stream()), "sfm_sample_philox_at")
    return out


def fit_eight_point(corr: torch.Tensor, S: torch.Tensor, E=None, flags=None, lambda2=None):
    """corr [B,N,4], S [B,H,8] -> E [B,H,9], flags [B,H]."""
    B, N, _ = corr.shape
    H = S.shape[1]
    assert S.shape == (B, H, 8) and S.dtype == torch.int32
    if lambda2 is None:
        op = ops.load()
        if E is None and flags is None:
            return op.fit_eight_point(corr, S)
        if E is None:
            E = torch.empty((B, H, 9), dtype=F64, device=corr.device)
        if flags is None:
            flags = torch.empty((B, H), dtype=torch.int32, device=corr.device)
        op.fit_eight_point_(corr, S, E, flags)
        return E, flags
    lib = _native.load()  # with the second-smallest eigenvalue written out: a diagnostic outside the op set
    if E is None:
        E = torch.empty((B, H, 9), dtype=F64, device=corr.device)
    if flags is None:
        flags = torch.empty((B, H), dtype=torch.int32, device=corr.device)
    check(lib.sfm_fit_eight_point(_ptr(corr), N, _ptr(S), H, B, _ptr(E), _ptr(flags), _ptr(lambda2),
                                  _stream()), "sfm_fit_eight_point")
    return E, flags


TRACE_FIELDS = {  # name -> (offset, shape) inside one trace record of sfm_fit_eight_point_traced
    "norm_a": (0, (8, 2)), "norm_b": (16, (8, 2)), "T1": (32, (3,)), "T2": (35, (3,)),
    "yty": (38, (9, 9)), "eigenvalues": (119, (9,)), "f_est": (128, (3, 3)), "f_rank2": (137, (3, 3)),
}


def sample_fit_philox(corr: torch.Tensor, seed, h_begin: int, S: torch.Tensor, E: torch.Tensor, flags: torch.Tensor,
                      seed_stride: int = 1) -> None:
    """Philox sampling and the eight-point fit in one launch (fills S, E, flags).  ``seed``: an int, or an int64
    device tensor whose first word is read at kernel run time (graph replay)."""
    on_device = isinstance(seed, torch.Tensor)
    ops.load().sample_fit_philox_(corr, 0 if on_device else _as_int64(seed), seed if on_device else None,
                                  seed_stride, h_begin, S, E, flags)


def fit_eight_point_traced(corr: torch.Tensor, S: torch.Tensor):
    """corr [B,N,4], S [B,H,8] -> E [B,H,9], flags [B,H], dict of intermediate arrays (numpy, [B,H,...])."""
    lib = _native.load()
    B, N, _ = corr.shape
    H = S.shape[1]
    nd = lib.sfm_fit_trace_doubles()
    E = torch.empty((B, H, 9), dtype=F64, device=corr.device)
    flags = torch.empty((B, H), dtype=torch.int32, device=corr.device)
    trace = torch.empty((B, H, nd), dtype=F64, device=corr.device)
    check(lib.sfm_fit_eight_point_traced(_ptr(corr), N, _ptr(S), H, B, _ptr(E), _ptr(flags), _ptr(trace),
                                         _stream()), "sfm_fit_eight_point_traced")
    raw = trace.cpu().numpy()
    fields = {}
    for name, (off, shape) in TRACE_FIELDS.items():
        size = int(np.prod(shape))
        fields[name] = raw[..., off:off + size].reshape((B, H) + shape)
    return E, flags, fields


def fit_stage(stage: int, data: np.ndarray, out_size: int) -> np.ndarray:
    """Run one single-problem stage of the fit (see sfm_fit_stage in include/sfm_hip.h)."""
    lib = _native.load()
    inp = to_device(np.ascontiguousarray(data, dtype=np.float64).reshape(-1))
    out = torch.empty((out_size,), dtype=F64, device=inp.device)
    check(lib.sfm_fit_stage(stage, _ptr(inp), _ptr(out), _stream()), "sfm_fit_stage")
    return out.cpu().numpy()


def hartley_normalize(coords: np.ndarray):
    """(n,2) -> normalised (n,2), forward transform T (3,3) (reference _normalize_coords)."""
    lib = _native.load()
    n = coords.shape[0]
    inp = to_device(np.ascontiguousarray(coords, dtype=np.float64))
    out = torch.empty((2 * n + 3,), dtype=F64, device=inp.device)
    check(lib.sfm_hartley_normalize(_ptr(inp), n, _ptr(out), _stream()), "sfm_hartley_normalize")
    raw = out.cpu().numpy()
    scale, cx, cy = raw[2 * n:]
    T = np.array([[scale, 0.0, -scale * cx], [0.0, scale, -scale * cy], [0.0, 0.0, 1.0]])
    return raw[:2 * n].reshape(n, 2).copy(), T


def score_workspace_bytes(n: int, h_count: int, batch: int, options: Optional[ScoreOptions] = None) -> int:
    """Bytes of scoring workspace a call with ``options`` (default: the process-wide set) needs — sized by what it will launch
    (``sfm_score_workspace_bytes_ex``)."""
    size = int(_native.load().sfm_score_workspace_bytes_ex(n, h_count, batch, None if options is None else C.byref(options)))
    if size < 0:
        raise ValueError("sfm_score_workspace_bytes_ex: negative size or an option out of range")
    return size


def score_workspace(n: int, h_count: int, batch: int, device, options: Optional[ScoreOptions] = None) -> torch.Tensor:
    """Scratch buffer that enables the two-tier scoring kernels (see include/sfm_hip.h), sized for calls made with
    ``options`` (default: the process-wide set)."""
    return torch.empty((score_workspace_bytes(n, h_count, batch, options),), dtype=torch.uint8, device=device)


def score_sed(corr: torch.Tensor, E: torch.Tensor, S: torch.Tensor, thr: float, cnt=None, s1=None,
              s2=None, workspace: Optional[torch.Tensor] = None, exact_only: bool = False,
              options: Optional[ScoreOptions] = None):
    """Per-hypothesis (extra-inlier count, sum sed, sum sed^2).  Uses the two-tier kernel (fp32 pre-filter
    + exact fp64) unless ``exact_only``; both give identical counts / decisions.  ``options`` (launch options of the
    two-tier kernels: which filter kernel, ranges, order ...; ``sfm_score_sed_ex``) default to the process-wide set, which
    the SFM_SCORE_* variables initialised when the library was loaded."""
    op = ops.load()
    B, N, _ = corr.shape
    H = E.shape[1]
    exact = exact_only or os.environ.get("SFM_SCORE_KERNEL", "filtered") == "exact"
    if options is not None:   # per-call launch options (and timing events): straight through the C ABI
        if cnt is None:
            cnt = torch.empty((B, H), dtype=torch.int32, device=corr.device)
        if s1 is None:
            s1 = torch.empty((B, H), dtype=F64, device=corr.device)
        if s2 is None:
            s2 = torch.empty((B, H), dtype=F64, device=corr.device)
        if exact:
            workspace = None   # NULL selects the all-fp64 kernel
        elif workspace is None:
            workspace = score_workspace(N, H, B, corr.device, options)
        assert S.dtype == torch.int32 and corr.dtype == F64 and E.dtype == F64
        with torch.cuda.device(corr.device):
            check(_native.load().sfm_score_sed_ex(_ptr(corr), N, _ptr(E), _ptr(S), H, B, float(thr), _ptr(cnt), _ptr(s1),
                                                  _ptr(s2), _ptr(workspace), workspace.numel() if workspace is not None else 0,
                                                  _stream(), C.byref(options)), "sfm_score_sed_ex")
        return cnt, s1, s2
    if cnt is None and s1 is None and s2 is None and (exact or workspace is None):
        return op.score_sed(corr, E, S, float(thr), exact)
    if cnt is None:
        cnt = torch.empty((B, H), dtype=torch.int32, device=corr.device)
    if s1 is None:
        s1 = torch.empty((B, H), dtype=F64, device=corr.device)
    if s2 is None:
        s2 = torch.empty((B, H), dtype=F64, device=corr.device)
    if exact:
        workspace = None
    elif workspace is None:
        workspace = score_workspace(N, H, B, corr.device)
    op.score_sed_(corr, E, S, float(thr), cnt, s1, s2, workspace)
    return cnt, s1, s2


def default_score_options() -> ScoreOptions:
    """The process-wide launch options of the scoring kernels (what calls without ``options`` use)."""
    out = ScoreOptions()
    check(_native.load().sfm_score_get_default_options(C.byref(out)), "sfm_score_get_default_options")
    return out


def set_default_score_options(options: Optional[ScoreOptions]) -> None:
    """Replace the process-wide launch options (``None`` = the library's built-in defaults).  The package sets them once
    from the SFM_SCORE_* variables when the library is loaded; bench.py switches kernels with this for its variants."""
    check(_native.load().sfm_score_set_default_options(None if options is None else C.byref(options)),
          "sfm_score_set_default_options")


SMALL_PASS_MAX_POINTS, SMALL_PASS_MAX_HYPOTHESES = 8192, 32768


def small_pass_eligible(batch: int, n: int, h: int) -> bool:
    """Whether a pass can run as the lean small pass (``sfm_ransac_pass_small``): one pair, at most 8192
    correspondences and 32768 hypotheses, the default two-tier scoring kernel.  ``SFM_SMALL_PASS=0`` keeps the
    separate calls (for A/B comparisons)."""
    return (batch == 1 and 8 <= n <= SMALL_PASS_MAX_POINTS and 1 <= h <= SMALL_PASS_MAX_HYPOTHESES
            and os.environ.get("SFM_SMALL_PASS", "1") != "0"
            and os.environ.get("SFM_SCORE_KERNEL", "filtered") != "exact")


def _pass_with_options(entry: str, corr, seed, seed_dev, use_philox, h_begin, thr, min_extra, aggregation, h_offset, S, E, flags,
                       cnt, s1, s2, result, mask, workspace, options: ScoreOptions) -> None:
    """A fused pass with per-call launch options (``sfm_ransac_pass_small`` / ``_large`` through the C ABI: the torch ops carry
    no options argument and run with the process-wide defaults)."""
    assert corr.shape[0] == 1, "one image pair per call"
    n, h = corr.shape[1], S.shape[1]
    with torch.cuda.device(corr.device):
        check(getattr(_native.load(), entry)(seed & (2**64 - 1), _ptr(seed_dev), 1 if use_philox else 0, h_begin, _ptr(corr), n, h,
                                             float(thr), float(min_extra), int(aggregation), h_offset, _ptr(S), _ptr(E),
                                             _ptr(flags), _ptr(cnt), _ptr(s1), _ptr(s2), _ptr(result), _ptr(mask),
                                             _ptr(workspace), workspace.numel(), _stream(), C.byref(options)), entry)


def ransac_pass_small(corr, S, E, flags, cnt, s1, s2, result, mask, workspace, thr: float, min_extra: float,
                      aggregation: int, h_offset: int = 0, philox=None, options: Optional[ScoreOptions] = None) -> None:
    """One whole pass of a small problem with lean launches (fit + workspace preparation, scoring, sharded selection, mask).
    ``philox=(seed, h_begin)``: samples drawn in the kernel (``seed`` an int or an int64 device tensor), else the
    table already in ``S``.  Same outputs as the separate calls.  ``options``: launch options of this call (default: the
    process-wide set)."""
    if philox is None:
        seed, seed_dev, use_philox, h_begin = 0, None, False, 0
    else:
        seed, h_begin = philox
        on_device = isinstance(seed, torch.Tensor)
        seed, seed_dev, use_philox = (0, seed, True) if on_device else (_as_int64(seed), None, True)
    if options is not None:
        return _pass_with_options("sfm_ransac_pass_small", corr, seed, seed_dev, use_philox, h_begin, thr, min_extra, aggregation,
                                  h_offset, S, E, flags, cnt, s1, s2, result, mask, workspace, options)
    ops.load().ransac_pass_small_(corr, seed, seed_dev, use_philox, h_begin, float(thr), float(min_extra),
                                  int(aggregation), h_offset, S, E, flags, cnt, s1, s2, result, mask, workspace)


def ransac_pass_large(corr, S, E, flags, cnt, s1, s2, result, mask, workspace, thr: float, min_extra: float,
                      aggregation: int, h_offset: int = 0, philox=None, options: Optional[ScoreOptions] = None) -> None:
    """One whole pass of a large problem (one pair) in eight launches instead of eighteen (``sfm_ransac_pass_large``): fit,
    partial maxima + zeroing, both operand tables, cost pre-pass, class histogram, scan + scatter, the scoring kernel, fold of
    the point ranges + selection + mask.  Arguments and outputs as ``ransac_pass_small``."""
    if philox is None:
        seed, seed_dev, use_philox, h_begin = 0, None, False, 0
    else:
        seed, h_begin = philox
        on_device = isinstance(seed, torch.Tensor)
        seed, seed_dev, use_philox = (0, seed, True) if on_device else (_as_int64(seed), None, True)
    if options is not None:
        return _pass_with_options("sfm_ransac_pass_large", corr, seed, seed_dev, use_philox, h_begin, thr, min_extra, aggregation,
                                  h_offset, S, E, flags, cnt, s1, s2, result, mask, workspace, options)
    ops.load().ransac_pass_large_(corr, seed, seed_dev, use_philox, h_begin, float(thr), float(min_extra),
                                  int(aggregation), h_offset, S, E, flags, cnt, s1, s2, result, mask, workspace)


def ransac_pass_batch(corr, S, E, flags, cnt, s1, s2, result, mask, workspace, thr: float, min_extra: float, aggregation: int,
                      philox=None, options: Optional[ScoreOptions] = None) -> None:
    """One whole pass of a BATCH of image pairs (``sfm_ransac_pass_batch``): corr [B,N,4], every other array with the leading
    pair dimension.  ``philox=(seed, h_begin, seed_stride)``: pair b draws from Philox(seed + b * seed_stride) inside the fit
    kernel (``seed`` an int or an int64 device tensor), else the tables already in ``S``.  Same outputs as the separate calls."""
    B, N, _ = corr.shape
    H = S.shape[1]
    if philox is None:
        seed, seed_dev, use_philox, h_begin, seed_stride = 0, None, False, 0, 0
    else:
        seed, h_begin, seed_stride = philox
        on_device = isinstance(seed, torch.Tensor)
        seed, seed_dev, use_philox = (0, seed, True) if on_device else (seed & (2**64 - 1), None, True)
    with torch.cuda.device(corr.device):
        check(_native.load().sfm_ransac_pass_batch(seed, _ptr(seed_dev), seed_stride, 1 if use_philox else 0, h_begin, _ptr(corr), N, H, B,
                                                   float(thr), float(min_extra), int(aggregation), _ptr(S), _ptr(E), _ptr(flags),
                                                   _ptr(cnt), _ptr(s1), _ptr(s2), _ptr(result), _ptr(mask), _ptr(workspace),
                                                   workspace.numel(), _stream(), None if options is None else C.byref(options)),
              "sfm_ransac_pass_batch")


def batch_pass_eligible(batch: int) -> bool:
    """Whether ``RansacWorkspace.run`` takes the fused batched pass (several pairs; ``SFM_LARGE_PASS=0`` keeps the separate calls)."""
    return (batch > 1 and os.environ.get("SFM_LARGE_PASS", "1") != "0"
            and os.environ.get("SFM_SCORE_KERNEL", "filtered") != "exact")


def large_pass_eligible(batch: int, n: int, h: int) -> bool:
    """Whether ``RansacWorkspace.run`` takes the fused large pass: one pair whose scoring call would launch the matrix-pipe
    kernel (the size rule of ``sfm_score_kernel_choice``).  ``SFM_LARGE_PASS=0`` keeps the separate calls."""
    if batch != 1 or os.environ.get("SFM_LARGE_PASS", "1") == "0" or os.environ.get("SFM_SCORE_KERNEL", "filtered") == "exact":
        return False
    return _native.load().sfm_score_kernel_choice(n, h, 1) == _native.SCORE_KERNEL_MATRIX


def select_best(cnt, s1, s2, flags, min_extra: float, aggregation: int, h_offset: int = 0, out=None):
    """-> int64 tensor [B,5] viewing the sfm_select_result records."""
    op = ops.load()
    if out is None:
        return op.select_best(cnt, s1, s2, flags, float(min_extra), int(aggregation), h_offset)
    op.select_best_(cnt, s1, s2, flags, float(min_extra), int(aggregation), h_offset, out)
    return out


def inlier_mask(corr, E, S, result, thr: float, out=None):
    op = ops.load()
    if out is None:
        return op.inlier_mask(corr, E, S, result, float(thr))
    op.inlier_mask_(corr, E, S, result, float(thr), out)
    return out


def refine_inliers(corr, E, mask, err, thr: float, aggregation: int, iterations: int = 1):
    """Local optimisation: refit E on all inliers, re-score, keep if better.
    corr [B,N,4], E [B,9], mask uint8 [B,N], err f64 [B] -> (E_out [B,9], mask_out uint8 [B,N],
    info int64 view [B,2]: {error bits, count | accepted << 32}).  No host synchronisation."""
    lib = _native.load()
    B, N, _ = corr.shape
    E_out = torch.empty((B, 9), dtype=F64, device=corr.device)
    mask_out = torch.empty((B, N), dtype=torch.uint8, device=corr.device)
    info = torch.empty((B, 2), dtype=torch.int64, device=corr.device)
    check(lib.sfm_refine_inliers(_ptr(corr), N, B, _ptr(E.contiguous()), _ptr(mask.contiguous()),
                                 _ptr(err.contiguous()), float(thr), int(aggregation), int(iterations),
                                 _ptr(E_out), _ptr(mask_out), _ptr(info), _stream()), "sfm_refine_inliers")
    return E_out, mask_out, info


def read_refine_info(info: torch.Tensor):
    """Host copy of sfm_refine_info records -> list of (error, count, accepted) (synchronises)."""
    raw = info.cpu().numpy()
    return [(float(raw[i, 0:1].view(np.float64)[0]), int(raw[i, 1] & 0xFFFFFFFF), int(raw[i, 1] >> 32))
            for i in range(raw.shape[0])]


def read_select(result: torch.Tensor) -> List[SelectResult]:
    """Copy the select records to the host (synchronises)."""
    raw = result.cpu().numpy().tobytes()
    return [SelectResult.from_buffer_copy(raw[i * SELECT_BYTES:(i + 1) * SELECT_BYTES])
            for i in range(result.shape[0])]


def sed_values(corr: torch.Tensor, E: torch.Tensor) -> torch.Tensor:
    lib = _native.load()
    n = corr.shape[0]
    out = torch.empty((n,), dtype=F64, device=corr.device)
    check(lib.sfm_sed_values(_ptr(corr), n, _ptr(E), _ptr(out), _stream()), "sfm_sed_values")
    return out


def cheirality(corr: torch.Tensor, pose_rt: torch.Tensor, distance_threshold: float) -> torch.Tensor:
    return ops.load().cheirality(corr, pose_rt, float(distance_threshold))


def triangulate(corr: torch.Tensor, P1: torch.Tensor, P2: torch.Tensor) -> torch.Tensor:
    return ops.load().triangulate(corr, P1, P2)


def decompose_essential(E: torch.Tensor, out=None):
    """E [B,9] -> pose_rt [B,4,12] (rows R(9)|t(3) in the reference's candidate order), status [B]."""
    lib = _native.load()
    B = E.shape[0]
    if out is None:
        poses = torch.empty((B, 4, 12), dtype=F64, device=E.device)
        status = torch.empty((B,), dtype=torch.int32, device=E.device)
    else:
        poses, status = out
    check(lib.sfm_decompose_essential(_ptr(E), B, _ptr(poses), _ptr(status), _stream()),
          "sfm_decompose_essential")
    return poses, status


# ------------------------------------------------------------------------------------------------------
# host sampler: exact replay of the reference's cumulative random.shuffle (ransac.py:59-64)
# ------------------------------------------------------------------------------------------------------
def pyshuffle_table(n: int, iterations: int, rng=_pyrandom, snapshot_iteration: int = -1,
                    advance: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """Sample table S[iterations, 8] the reference would draw from the current state of ``rng``
    (the ``random`` module or a ``random.Random``), via the C++ MT19937 replay.  With ``advance`` the
    generator state is advanced exactly as ``iterations`` calls of ``random.shuffle`` would.
    Optionally also returns the full permutation after ``snapshot_iteration``."""
    lib = _native.load()
    version, internal, gauss = rng.getstate()
    state = (C.c_uint32 * 624)(*internal[:624])
    index = C.c_int32(internal[624])
    S = np.empty((iterations, 8), dtype=np.int32)
    snap = np.empty((n,), dtype=np.int32) if snapshot_iteration >= 0 else None
    check(lib.sfm_pyshuffle_table(
        C.cast(state, C.c_void_p), C.cast(C.byref(index), C.c_void_p), n, iterations,
        S.ctypes.data_as(C.c_void_p), None, snapshot_iteration,
        snap.ctypes.data_as(C.c_void_p) if snap is not None else None), "sfm_pyshuffle_table")
    if advance:
        rng.setstate((version, tuple(state) + (index.value,), gauss))
    return S, snap


class PyShuffleTable:
    """The sample table of ``iterations`` cumulative ``random.shuffle`` calls (``pyshuffle_table``) that also keeps
    up to ``checkpoints`` snapshots of (generator state, permutation) on the way, so that the full permutation after
    any iteration — the order in which the reference returns the winner's inliers — is re-derived by replaying at
    most ``iterations / checkpoints`` shuffles instead of all of them up to the winner."""

    def __init__(self, n: int, iterations: int, rng=_pyrandom, advance: bool = True, checkpoints: int = 32):
        lib = _native.load()
        self.n, self.iterations = n, iterations
        self.stride = max(1, -(-iterations // max(1, checkpoints)))
        version, internal, gauss = rng.getstate()
        state = (C.c_uint32 * 624)(*internal[:624])
        index = C.c_int32(internal[624])
        perm = np.arange(n, dtype=np.int32)
        self.S = np.empty((iterations, 8), dtype=np.int32)
        self._saved = []  # (state bytes, index, permutation) before each segment
        for start in range(0, iterations, self.stride):
            count = min(self.stride, iterations - start)
            self._saved.append((bytes(state), index.value, perm.copy()))
            check(lib.sfm_pyshuffle_table(
                C.cast(state, C.c_void_p), C.cast(C.byref(index), C.c_void_p), n, count,
                self.S[start:].ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p), -1, None),
                "sfm_pyshuffle_table")
        if advance:
            rng.setstate((version, tuple(state) + (index.value,), gauss))

    def permutation_after(self, iteration: int) -> np.ndarray:
        """The shuffled index list as it stood after ``iteration`` (0-based) — ``data`` of ransac.py:62-64."""
        if not 0 <= iteration < self.iterations:
            raise IndexError(iteration)
        lib = _native.load()
        state_bytes, index_value, perm = self._saved[iteration // self.stride]
        state = (C.c_uint32 * 624).from_buffer_copy(state_bytes)
        index = C.c_int32(index_value)
        perm = perm.copy()
        count = iteration % self.stride + 1
        scratch = np.empty((count, 8), dtype=np.int32)
        check(lib.sfm_pyshuffle_table(
            C.cast(state, C.c_void_p), C.cast(C.byref(index), C.c_void_p), self.n, count,
            scratch.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p), -1, None), "sfm_pyshuffle_table")
        return perm


# ------------------------------------------------------------------------------------------------------
# the RANSAC engine: sample table -> fit -> score -> select -> mask, all enqueued on one stream
# ------------------------------------------------------------------------------------------------------
@dataclass
class RansacOutcome:
    best_h: int                # winning hypothesis (global index), -1 if none
    error: float               # aggregated inlier error of the winner
    E: Optional[np.ndarray]    # (3,3) essential matrix of the winner
    sample: Optional[np.ndarray]   # (8,) indices of the winner's sample, in sample order
    mask: Optional[np.ndarray]     # (N,) uint8: 1 survivor, 2 sample point, 0 outlier
    n_flagged: int             # hypotheses whose sample was degenerate (eight_point.py:415-421)
    first_flagged: int         # lowest such hypothesis index, or -1
    extra_inliers: int


def checked_mask(mask: np.ndarray) -> np.ndarray:
    """Inlier mask as read back from the device (0 out, 1 inlier, 2 sample).  The mask blocks of a small pass's
    selection launch wait for the selecting blocks' record with a bounded number of polls and fill their slice with
    0xFF if it never arrives (``select_sharded_kernel``): that must surface here, not as an empty inlier list."""
    if mask.size and int(mask.max()) > 2:
        raise RuntimeError("sfm_hip: the inlier mask was not written (the selection record of the pass never arrived)")
    return mask


class RansacWorkspace:
    """Pre-allocated device buffers for B pairs x H hypotheses x N correspondences."""

    def __init__(self, batch: int, n: int, h: int, device=None):
        dev = device or require_gpu()
        self.batch, self.n, self.h = batch, n, h
        self.S = torch.empty((batch, h, 8), dtype=torch.int32, device=dev)
        self.E = torch.empty((batch, h, 9), dtype=F64, device=dev)
        self.flags = torch.empty((batch, h), dtype=torch.int32, device=dev)
        self.cnt = torch.empty((batch, h), dtype=torch.int32, device=dev)
        self.s1 = torch.empty((batch, h), dtype=F64, device=dev)
        self.s2 = torch.empty((batch, h), dtype=F64, device=dev)
        self.result = torch.empty((batch, SELECT_BYTES // 8), dtype=torch.int64, device=dev)
        self.mask = torch.empty((batch, n), dtype=torch.uint8, device=dev)
        self.score_ws = score_workspace(n, h, batch, dev)   # (for the process-wide options of this moment: see _fit_workspace)
        self.frozen = False   # set by ShardedRansac.capture: the buffers' addresses are baked into a graph

    def _fit_workspace(self, options: Optional[ScoreOptions]) -> None:
        """The scoring workspace is sized by what a call launches: other options (per call, or process-wide defaults changed
        since this engine was built) may need more — grown here, outside any captured graph, rather than refused by the library."""
        need = score_workspace_bytes(self.n, self.h, self.batch, options)
        if need > self.score_ws.numel():
            if self.frozen:   # a captured HIP graph holds the address of the present buffer
                raise RuntimeError("RansacWorkspace: these launch options need a larger scoring workspace than the one a captured "
                                   "graph of this engine refers to; build a new engine (or capture again) with them")
            self.score_ws = torch.empty((need,), dtype=torch.uint8, device=self.score_ws.device)

    def run(self, corr: torch.Tensor, thr: float, min_extra: float, aggregation: int,
            h_offset: int = 0, with_mask: bool = True, philox=None, options: Optional[ScoreOptions] = None) -> None:
        """fit + score + select (+ mask) for the sample table currently in ``self.S`` — or, with
        ``philox=(seed, h_begin, seed_stride)``, for Philox samples drawn inside the fit kernel (which also fills
        ``self.S``); ``seed`` may be an int64 device tensor (read at kernel run time).  ``options``: launch options of the
        scoring launch of THIS pass (timing events included); default: the process-wide set."""
        if not torch.cuda.is_current_stream_capturing():
            self._fit_workspace(options)
        if small_pass_eligible(self.batch, self.n, self.h):
            # workspace preparation rides in the fit launch, selection over 32 blocks (seed_stride only matters for batches)
            ransac_pass_small(corr, self.S, self.E, self.flags, self.cnt, self.s1, self.s2, self.result,
                              self.mask if with_mask else None, self.score_ws, thr, min_extra, aggregation, h_offset,
                              None if philox is None else (philox[0], philox[1]), options)
            return
        if large_pass_eligible(self.batch, self.n, self.h) and (not with_mask or h_offset == 0):
            # eight launches instead of eighteen: setup and tables fused, the ranges folded inside the selection launch
            ransac_pass_large(corr, self.S, self.E, self.flags, self.cnt, self.s1, self.s2, self.result,
                              self.mask if with_mask else None, self.score_ws, thr, min_extra, aggregation, h_offset,
                              None if philox is None else (philox[0], philox[1]), options)
            return
        if batch_pass_eligible(self.batch) and h_offset == 0:
            # a batch of pairs: partial maxima + zeroing, point tables, fits (+ the hypotheses' operand rows), pre-pass, sort,
            # scoring, and per pair one block that folds the ranges, selects and writes the mask
            ransac_pass_batch(corr, self.S, self.E, self.flags, self.cnt, self.s1, self.s2, self.result,
                              self.mask if with_mask else None, self.score_ws, thr, min_extra, aggregation, philox, options)
            return
        if philox is None:
            fit_eight_point(corr, self.S, self.E, self.flags)
        else:
            seed, h_begin, seed_stride = philox
            sample_fit_philox(corr, seed, h_begin, self.S, self.E, self.flags, seed_stride)
        score_sed(corr, self.E, self.S, thr, self.cnt, self.s1, self.s2, workspace=self.score_ws, options=options)
        select_best(self.cnt, self.s1, self.s2, self.flags, min_extra, aggregation, h_offset,
                    self.result)
        if with_mask:
            assert h_offset == 0, "mask needs local hypothesis indices"
            inlier_mask(corr, self.E, self.S, self.result, thr, self.mask)

    def outcome(self, b: int = 0, h_offset: int = 0) -> RansacOutcome:
        rec = read_select(self.result)[b]
        first = -1 if rec.first_flagged == _native.INT64_MAX else int(rec.first_flagged)
        if rec.best_h < 0:
            return RansacOutcome(-1, float("inf"), None, None, None, int(rec.n_flagged), first, 0)
        local = int(rec.best_h) - h_offset
        E = self.E[b, local].cpu().numpy().reshape(3, 3).copy()
        sample = self.S[b, local].cpu().numpy().astype(np.int64)
        mask = checked_mask(self.mask[b].cpu().numpy().copy())
        return RansacOutcome(int(rec.best_h), float(rec.best_err), E, sample, mask, int(rec.n_flagged),
                             first, int(rec.best_cnt))


def ransac_essential(corr: torch.Tensor, S, thr: float, min_extra: float, aggregation: int) -> RansacOutcome:
    """One image pair: corr [N,4] on device, S [H,8] (numpy or tensor) -> RansacOutcome."""
    n = corr.shape[0]
    S_t = S if isinstance(S, torch.Tensor) else to_device(S, torch.int32)
    h = S_t.shape[0]
    ws = RansacWorkspace(1, n, h, corr.device)
    ws.S.copy_(S_t.reshape(1, h, 8))
    ws.run(corr.reshape(1, n, 4), thr, min_extra, aggregation)
    return ws.outcome(0)
