"""Same-box baseline: the scoring call of ROUND 3's library (tools/r04/base/libsfm_hip_r03.so, cross-built from commit bca5a51;
bound directly with ctypes — its ABI is 9) on the inputs the current library prepares, beside the current library's call.
N, H, THR from the environment; SFM_SCORE_MATRIX is read by the old library itself (per call)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from structure_from_motion_amd import device as dev, synthetic  # noqa: E402

n, h = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000))
thr = float(os.environ.get("THR", 1.5e-6))
old = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "base", os.environ.get("OLD_LIB", "libsfm_hip_r03.so")))
old.sfm_score_workspace_bytes.restype = C.c_int64
old.sfm_score_workspace_bytes.argtypes = [C.c_int64] * 3
old.sfm_score_sed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
old.sfm_score_set_timing_events.argtypes = [C.c_void_p, C.c_void_p]
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ref = dev.score_sed(corr, E, S, thr, exact_only=True)
ws_old = torch.empty((int(old.sfm_score_workspace_bytes(n, h, 1)),), dtype=torch.uint8, device=corr.device)
ws_new = dev.score_workspace(n, h, 1, corr.device)
out = [torch.empty((1, h), dtype=torch.int32, device="cuda"), torch.empty((1, h), dtype=torch.float64, device="cuda"),
       torch.empty((1, h), dtype=torch.float64, device="cuda")]
before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
before.record(); after.record(); torch.cuda.synchronize()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call_old():
    rc = old.sfm_score_sed(corr.data_ptr(), n, E.data_ptr(), S.data_ptr(), h, 1, thr, out[0].data_ptr(), out[1].data_ptr(),
                           out[2].data_ptr(), ws_old.data_ptr(), ws_old.numel(), stream)
    assert rc == 0, rc


def call_new():
    dev.score_sed(corr, E, S, thr, *out, workspace=ws_new)


for name, call, hook in (("round-3 library", call_old, lambda a, b: old.sfm_score_set_timing_events(a, b)),
                         ("this library   ", call_new, lambda a, b: dev._native.load().sfm_score_set_timing_events(a, b))):
    hook(before.cuda_event, after.cuda_event)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    bad = int((out[0] != ref[0]).sum().item())
    kernel, whole = [], []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(int(os.environ.get("REPS", 10))):
        a.record(); call(); b.record(); torch.cuda.synchronize()
        kernel.append(before.elapsed_time(after)); whole.append(a.elapsed_time(b))
    hook(None, None)
    print(f"{name} N={n} H={h} thr={thr:g} MATRIX={os.environ.get('SFM_SCORE_MATRIX', '-')}: counts differing {bad}; scoring kernel "
          f"{np.median(kernel):.3f} ms (min {min(kernel):.3f}), whole scoring call {np.median(whole):.3f} ms", flush=True)
