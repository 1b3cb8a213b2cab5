"""Diagnostic: does longest-first ordering of hypotheses shorten the two-tier scoring kernel (tail effect)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import device as dev, synthetic
n, h = 50000, 100000
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)
cnt, s1, s2 = dev.score_sed(corr, E, S, 1.5e-6, workspace=ws)
def timeit(E_, S_, label):
    out = [torch.empty_like(cnt), torch.empty_like(s1), torch.empty_like(s2)]
    for _ in range(2): dev.score_sed(corr, E_, S_, 1.5e-6, *out, workspace=ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): dev.score_sed(corr, E_, S_, 1.5e-6, *out, workspace=ws)
    b.record(); torch.cuda.synchronize()
    print(f"{label}: {a.elapsed_time(b)/5:.3f} ms", flush=True)
timeit(E, S, "original order")
order = torch.argsort(cnt[0], descending=True)
timeit(E[:, order].contiguous(), S[:, order].contiguous(), "longest first (sorted by inlier count)")
order2 = torch.argsort(cnt[0], descending=False)
timeit(E[:, order2].contiguous(), S[:, order2].contiguous(), "shortest first")
