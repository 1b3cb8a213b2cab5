"""Diagnostic (needs a build with -DSFM_WAVE_STAMPS=1): per-wave start / end of the scoring launch of one lean small
pass (sfm_ransac_pass_small), with each wave's hypothesis, its inlier count and the number of exact-tier batches.
N, H from the environment (C2 by default).  s_memrealtime ticks at 100 MHz (10 ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import _native, device as dev, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

n, h = int(os.environ.get("N", 5000)), int(os.environ.get("H", 10000))
lib = _native.load()
lib.sfm_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_int64]
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
ws = dev.RansacWorkspace(1, n, h)
assert dev.small_pass_eligible(1, n, h)
for s in range(3):
    ws.run(corr, 1.5e-6, 10, AGG_RMS, philox=(1000 + s, 0, 0))
torch.cuda.synchronize()
hpw = int(os.environ.get("SFM_SCORE_HPW", 2 if (h + 1) // 2 >= 5120 else 1))
waves = (h + hpw - 1) // hpw
if int(os.environ.get("SFM_SCORE_RESIDENT", "0")) > 0:
    waves = int(os.environ["SFM_SCORE_RESIDENT"])
st = np.zeros((waves, 4), dtype=np.uint64)
assert lib.sfm_debug_read_wave_stamps(st.ctypes.data, waves) == 0
t0 = st[:, 0].min()
begin = (st[:, 0] - t0).astype(np.float64) * 0.01   # us
end = (st[:, 1] - t0).astype(np.float64) * 0.01
dur = end - begin
hyp = st[:, 2].astype(np.int64)
drains = st[:, 3].astype(np.int64)
cnt = ws.cnt.cpu().numpy()[0].astype(np.int64)[hyp]
print(f"n={n} h={h} hpw={hpw} waves={waves}: kernel span {end.max():.1f} us; wave duration median {np.median(dur):.1f}, "
      f"p90 {np.percentile(dur, 90):.1f}, p99 {np.percentile(dur, 99):.1f}, max {dur.max():.1f} us")
print("start times: median %.1f, p90 %.1f, max %.1f us" % (np.median(begin), np.percentile(begin, 90), begin.max()))
first = begin < 1.0
print(f"first generation: {first.sum()} waves, end median {np.median(end[first]):.1f} p99 {np.percentile(end[first], 99):.1f} max {end[first].max():.1f}")
if (~first).any():
    print(f"later waves: {(~first).sum()}, start median {np.median(begin[~first]):.1f}, duration median {np.median(dur[~first]):.1f} "
          f"max {dur[~first].max():.1f}, end max {end[~first].max():.1f}")
print("duration by exact-tier batches (drains): ")
for lo, hi in [(0, 1), (1, 3), (3, 8), (8, 20), (20, 40), (40, 200)]:
    sel = (drains >= lo) & (drains < hi)
    if sel.any():
        print(f"  {lo:3d}..{hi - 1:3d}: {sel.sum():5d} waves, wave index median {int(np.median(np.nonzero(sel)[0])):5d}, start median {np.median(begin[sel]):5.1f}, "
              f"duration median {np.median(dur[sel]):5.1f} max {dur[sel].max():5.1f}, end max {end[sel].max():5.1f}")
late = np.argsort(end)[-8:]
for w in late:
    print(f"  wave {w}: start {begin[w]:.1f} end {end[w]:.1f} us, hypothesis {hyp[w]}, inliers {cnt[w]}, drains {drains[w]}")
hist, edges = np.histogram(end, bins=10)
print("waves finishing per tenth of the span:", hist.tolist())
