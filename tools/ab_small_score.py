"""A/B of small-pass variants selected by environment switches the library reads once per process (SFM_SCORE_HPW, experiment
builds): runs passes at the
sizes given as N:H pairs on the command line, prints the time per pass and writes cnt / s1 / s2 / record of the last
pass of each size to OUT (npz) so that two runs can be compared with --compare A B."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for key in a.files:
        if key.startswith("cnt") or key.startswith("rec_h") or key.startswith("mask"):
            same = np.array_equal(a[key], b[key])
        else:
            same = np.allclose(a[key], b[key], rtol=1e-13, atol=0, equal_nan=True)
        print(f"{key}: {'same' if same else 'DIFFERENT'}", flush=True)
        bad += not same
    sys.exit(1 if bad else 0)

import torch  # noqa: E402

from structure_from_motion_amd import device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

out = {}
steps = int(os.environ.get("STEPS", 200))
for spec in sys.argv[1:]:
    n, h = (int(x) for x in spec.split(":"))
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    engine = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS)
    print(f"n={n} h={h}: first pass ...", flush=True)
    engine.step(100)
    torch.cuda.synchronize()
    print("  done", flush=True)
    for s in range(10):
        engine.step(101 + s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        engine.step(1000 + s)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / steps * 1e6
    print(f"n={n} h={h} hpw={os.environ.get('SFM_SCORE_HPW', 'default')}: {us:.1f} us/pass "
          f"({n * h / us * 1e6:.3e} evals/s)", flush=True)
    ws = engine.ws
    out[f"cnt_{spec}"] = ws.cnt.cpu().numpy()
    out[f"s1_{spec}"] = ws.s1.cpu().numpy()
    out[f"s2_{spec}"] = ws.s2.cpu().numpy()
    out[f"rec_h_{spec}"] = ws.result.cpu().numpy()[:, 1]
    out[f"mask_{spec}"] = ws.mask.cpu().numpy()
if os.environ.get("OUT"):
    np.savez(os.environ["OUT"], **out)
