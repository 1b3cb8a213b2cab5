"""Diagnostic (needs a build with -DSFM_WAVE_STAMPS=1): per-wave start / end times of the scoring kernel for one small
pass — is the pass bound by throughput or by its longest waves?  N, H from the environment (C2 by default).
s_memrealtime ticks at 100 MHz (10 ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import _native, device as dev, synthetic  # noqa: E402

n, h = int(os.environ.get("N", 5000)), int(os.environ.get("H", 10000))
lib = _native.load()
lib.sfm_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_int64]
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)
for _ in range(3):
    cnt, s1, s2 = dev.score_sed(corr, E, S, 1.5e-6, workspace=ws)
torch.cuda.synchronize()
hpw = 4
while hpw > 1 and (h + hpw - 1) // hpw < 5120:
    hpw //= 2
waves = (h + hpw - 1) // hpw
st = np.zeros((waves, 4), dtype=np.uint64)
assert lib.sfm_debug_read_wave_stamps(st.ctypes.data, waves) == 0
t0 = st[:, 0].min()
begin = (st[:, 0] - t0).astype(np.float64) * 0.01   # us
end = (st[:, 1] - t0).astype(np.float64) * 0.01
dur = end - begin
c = cnt.cpu().numpy()[0].astype(np.int64).reshape(waves, hpw).max(axis=1) if h % hpw == 0 else None
print(f"n={n} h={h} hpw={hpw} waves={waves}: kernel span {end.max():.1f} us; wave duration median {np.median(dur):.1f}, "
      f"p90 {np.percentile(dur, 90):.1f}, p99 {np.percentile(dur, 99):.1f}, max {dur.max():.1f} us")
print(f"sum of wave durations / (span x 5120 slots) = {dur.sum() / (end.max() * 5120):.2f}")
print("start times: median %.1f, p90 %.1f, max %.1f us" % (np.median(begin), np.percentile(begin, 90), begin.max()))
if c is not None:
    good = c > 0.3 * n
    print(f"waves with a hypothesis that fits (> 30 % inliers): {good.sum()}; their duration median {np.median(dur[good]):.1f} "
          f"max {dur[good].max():.1f}; the others median {np.median(dur[~good]):.1f} max {dur[~good].max():.1f}")
    late = np.argsort(end)[-5:]
    for w in late:
        print(f"  wave {w}: start {begin[w]:.1f} end {end[w]:.1f} us, max inliers {c[w]}")
hist, edges = np.histogram(end, bins=10)
print("waves finishing per tenth of the span:", hist.tolist())
