"""Small passes (C2 = 5000 x 10000 and the demo's 300 x 2000): microseconds per pass of the engine, fused small pass
(default) vs the separate calls (SFM_SMALL_PASS=0).  Under rocprofv3 --kernel-trace --stats this gives the per-kernel
split.  N, H, STEPS from the environment."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

n, h, steps = int(os.environ.get("N", 5000)), int(os.environ.get("H", 10000)), int(os.environ.get("STEPS", 300))
device.require_gpu()
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
engine = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS)
for s in range(20):
    engine.step(100 + s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    engine.step(1000 + s)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / steps * 1e6
print(f"n={n} h={h} small_pass={os.environ.get('SFM_SMALL_PASS', '1')}: {us:.1f} us/pass ({n * h / us * 1e6:.3e} evals/s)", flush=True)
