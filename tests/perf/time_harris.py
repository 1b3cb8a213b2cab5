"""Harris detector timing on a VGA image (the demo's image size class), oracle timed beside it."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from structure_from_motion_amd.harris import harris_detector as harris
from oracle import harris_oracle as ho
rng = np.random.default_rng(0)
for shape in [(480, 640), (1080, 1920)]:
    img = rng.integers(0, 256, shape).astype(np.uint8)
    harris.detect_harris_corners(img, 600); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        c = harris.detect_harris_corners(img, 600)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    s = harris._suppressed_cornerness(img, 2, 0.04); torch.cuda.synchronize()
    dev_only = time.perf_counter() - t0
    rec = dict(shape=shape, ms_total=dt * 1e3, ms_device_stencils=dev_only * 1e3, mpix_per_s=shape[0] * shape[1] / dt / 1e6, corners=len(c))
    if shape == (480, 640):
        t0 = time.perf_counter(); ho.detect_harris_corners(img, 600); rec["oracle_numpy_s"] = time.perf_counter() - t0
    print(json.dumps(rec), flush=True)
