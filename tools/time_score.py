"""Diagnostic: time the exact vs the two-tier scoring kernel on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import device as dev, synthetic
n, h = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000))
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)
out = [torch.empty((1, h), dtype=torch.int32, device="cuda"), torch.empty((1, h), dtype=torch.float64, device="cuda"), torch.empty((1, h), dtype=torch.float64, device="cuda")]
for name, kw in [("exact", dict(exact_only=True)), ("filtered", dict(workspace=ws))]:
    for _ in range(2):
        dev.score_sed(corr, E, S, 1.5e-6, *out, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        dev.score_sed(corr, E, S, 1.5e-6, *out, **kw)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{name}: {ms:.3f} ms  -> {n*h/ms/1e6:.1f} G evals/s", flush=True)
