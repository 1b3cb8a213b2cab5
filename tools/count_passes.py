"""Diagnostic: how many (hypothesis, point) pairs pass the fp32 filter (SFM_SCORE_ABLATE=1 makes the
kernel return the pass count instead of the inlier count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import device as dev, synthetic
n, h = 50000, 100000
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
cnt, s1, s2 = dev.score_sed(corr, E, S, 1.5e-6)
c = cnt.cpu().numpy()[0].astype(np.int64)
print("mode", os.environ.get("SFM_SCORE_ABLATE", "0"), "sum", c.sum(), "mean frac", c.mean() / n, "median", np.median(c) / n,
      "p99", np.percentile(c, 99) / n, "max", c.max() / n)
