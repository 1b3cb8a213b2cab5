"""Diagnostic: time the eight-point fit kernel alone (100k hypotheses) and a C5-shaped batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import device as dev, synthetic
for (B, n, h) in [(1, 50000, 100000), (256, 10000, 2000), (1, 5000, 10000), (1, 300, 2000)]:
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4).repeat(B, 1, 1).contiguous()
    S = dev.sample_philox(5, 0, h, n, batch=B)
    E = torch.empty((B, h, 9), dtype=torch.float64, device="cuda"); fl = torch.empty((B, h), dtype=torch.int32, device="cuda")
    for _ in range(2): dev.fit_eight_point(corr, S, E, fl)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): dev.fit_eight_point(corr, S, E, fl)
    b.record(); torch.cuda.synchronize()
    print(f"fit B={B} h={h}: {a.elapsed_time(b)/10*1e3:.1f} us, flagged {int(fl.sum())}", flush=True)
