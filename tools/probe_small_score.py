"""Diagnostic: scoring-call time of a small problem by hypotheses-per-wave and threshold (a tiny threshold leaves
tier 1 only).  N, H from the environment; the hypotheses per wave are a launch option of each call (ScoreOptions) — the
library does not read SFM_SCORE_* per call, and setting them after import changes nothing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import device as dev, synthetic  # noqa: E402

n, h = int(os.environ.get("N", 5000)), int(os.environ.get("H", 10000))
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device, dev.ScoreOptions(kernel="filtered", split=0))
for hpw in (1, 2, 4):
    options = dev.ScoreOptions(kernel="filtered", hyps_per_wave=hpw)
    for thr in (1.5e-6, 1e-10):
        for _ in range(3):
            dev.score_sed(corr, E, S, thr, workspace=ws, options=options)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            dev.score_sed(corr, E, S, thr, workspace=ws, options=options)
        b.record()
        torch.cuda.synchronize()
        print(f"n={n} h={h} hpw={hpw} thr={thr:g}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per scoring call")
