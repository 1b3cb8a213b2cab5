"""Brute-force NCC matcher timing (reference apps/sfm.py:73-87 shape: 600 x 600 corners, window 9) and a
large case; the numpy oracle is timed beside it on a bounded sample."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from structure_from_motion_amd import device as dev
from structure_from_motion_amd.feature_matching import _device_match
from oracle import match_oracle as mo

rng = np.random.default_rng(0)
H, W, ws = 480, 640, 9
ia = rng.integers(0, 256, (H, W)).astype(np.uint8)
ib = np.roll(ia, (2, 3), axis=(0, 1))
out = []
for nA, nB in [(600, 600), (20000, 20000)]:
    fa = np.column_stack([rng.integers(0, W, nA), rng.integers(0, H, nA)]).astype(np.float64)
    fb = np.column_stack([rng.integers(0, W, nB), rng.integers(0, H, nB)]).astype(np.float64)
    fa_t, fb_t = dev.to_device(fa), dev.to_device(fb)
    def run_matrix():  # |A| x |B| scores to HBM, then the row scan over them
        sc = _device_match.score_matrix(0, ia, ib, fa_t, fb_t, ws)
        return _device_match.row_summary(sc)
    def run():         # product path: fused tile summaries, no score matrix
        return _device_match.match_summary(0, ia, ib, fa_t, fb_t, ws)
    reps = 5
    timings = {}
    for name, fn in (("matrix", run_matrix), ("fused", run)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        timings[name] = (time.perf_counter() - t0) / reps
    dt = timings["fused"]
    K = ws * ws
    rec = dict(nA=nA, nB=nB, window=ws, ms=dt * 1e3, ms_matrix_path=timings["matrix"] * 1e3, pairs_per_s=nA * nB / dt,
               fp64_gflops=nA * nB * 2 * K / dt / 1e9, frac_of_fp64_valu_peak=nA * nB * 2 * K / dt / 78.6e12)
    # CPU oracle on a bounded sample of rows
    rows = min(nA, 300)
    t0 = time.perf_counter()
    mo.row_summary(mo.ncc_scores(ia, ib, fa[:rows], fb, ws))
    cpu = time.perf_counter() - t0
    rec["cpu_oracle_pairs_per_s"] = rows * nB / cpu
    rec["cpu_sample"] = f"{rows} rows x {nB} (numpy, 1 thread)"
    out.append(rec)
    print(json.dumps(rec), flush=True)
