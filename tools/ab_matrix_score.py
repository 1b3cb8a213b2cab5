"""Diagnostic: the scoring call of the bench workload (N, H, THR from the environment) under the launch options the SFM_SCORE_*
variables spell (translated once, when the library is loaded: run it once per setting), checked against the all-fp64 kernel:
counts bit-equal, sums to 1e-12; kernel time by HIP events carried in the call's options (sfm_score_options.timing_*)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.environ["SFM_TREE"]) if os.environ.get("SFM_TREE") else   # a tree prepared by tools/ab.py
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import device as dev, synthetic
n, h = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000))
thr = float(os.environ.get("THR", 1.5e-6))
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)   # (sized for the process-wide options: the SFM_SCORE_* variables)
ref = dev.score_sed(corr, E, S, thr, exact_only=True)
out = [torch.empty((1, h), dtype=torch.int32, device="cuda"), torch.empty((1, h), dtype=torch.float64, device="cuda"), torch.empty((1, h), dtype=torch.float64, device="cuda")]
before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
before.record(); after.record(); torch.cuda.synchronize()
timed = dev.default_score_options().with_timing(before, after)
for _ in range(2):
    dev.score_sed(corr, E, S, thr, *out, workspace=ws, options=timed)
torch.cuda.synchronize()
bad = int((out[0] != ref[0]).sum().item())
ok = torch.isfinite(ref[1])
e1 = float(((out[1] - ref[1]).abs()[ok] / ref[1].abs()[ok].clamp_min(1e-300)).max().item())
e2 = float(((out[2] - ref[2]).abs()[ok] / ref[2].abs()[ok].clamp_min(1e-300)).max().item())
print(f"N={n} H={h} MATRIX={os.environ.get('SFM_SCORE_MATRIX', '-')} SPLIT={os.environ.get('SFM_SCORE_SPLIT', '-')}: "
      f"counts differing {bad} of {h}; max rel diff of the sums {e1:.2e} {e2:.2e}", flush=True)
if bad:
    idx = torch.nonzero(out[0][0] != ref[0][0])[:8, 0].tolist()
    print("  first differing:", [(i, int(out[0][0, i]), int(ref[0][0, i])) for i in idx], flush=True)
kernel, call = [], []
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(int(os.environ.get("REPS", 10))):
    a.record()
    dev.score_sed(corr, E, S, thr, *out, workspace=ws, options=timed)
    b.record()
    torch.cuda.synchronize()
    kernel.append(before.elapsed_time(after))
    call.append(a.elapsed_time(b))
first = [t.clone() for t in out]
dev.score_sed(corr, E, S, thr, *out, workspace=ws)
torch.cuda.synchronize()
same = all(torch.equal(x.view(torch.int64) if x.dtype == torch.float64 else x, y.view(torch.int64) if y.dtype == torch.float64 else y) for x, y in zip(first, out))
from structure_from_motion_amd import _native
lib = _native.load()
if hasattr(lib, "sfm_debug_matrix_stats"):   # diagnostic build (-DSFM_MATRIX_STATS=1)
    import ctypes
    st = (ctypes.c_ulonglong * 4)()
    lib.sfm_debug_matrix_stats(st, 1)
    dev.score_sed(corr, E, S, thr, *out, workspace=ws)
    torch.cuda.synchronize()
    lib.sfm_debug_matrix_stats(st, 1)
    rounds, pops, pushes = st[0], st[1], st[2]
    print(f"  per launch: {rounds} exact-tier evaluations per lane summed over waves ({rounds * 64} lane slots), {pops} points popped "
          f"-> lane utilisation {pops / max(1, rounds * 64):.3f}; push-loop iterations {pushes} "
          f"({pushes / max(1, (h + 31) // 32 * ((n + 31) // 32)):.2f} per wave-step); survivors / evaluations {pops / (n * h):.4f}", flush=True)
print(f"  kernel {np.median(kernel):.3f} ms (min {min(kernel):.3f}), whole call {np.median(call):.3f} ms; run-to-run bit-identical: {same}", flush=True)
