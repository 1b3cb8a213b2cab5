"""Diagnostic (needs a build with SFM_EXTRA_HIPCC_FLAGS=-DSFM_WAVE_STAMPS=1): where the time of the LDS-resident small
scoring kernel goes — per-wave phase times and per-item durations of one pass.  N, H from the environment (C2 by default).
s_memrealtime ticks at 100 MHz (10 ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import _native, device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

n, h = int(os.environ.get("N", 5000)), int(os.environ.get("H", 10000))
lib = _native.load()
lib.sfm_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_int64]
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
engine = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS)
for s in range(5):
    engine.step(1000 + s)
torch.cuda.synchronize()
blocks = min(256, h)
waves = blocks * 16
raw = np.zeros((waves * 8, 4), dtype=np.uint64)   # 32 words per wave
assert lib.sfm_debug_read_wave_stamps(raw.ctypes.data, waves * 8) == 0
st = raw.reshape(waves, 32).astype(np.int64)
t0 = st[:, 0].min()
us = lambda x: (x - t0) * 0.01
begin, p1s, p1e, p2e, end = (us(st[:, k]) for k in range(5))
items, batches = st[:, 5], st[:, 6]
cnt = engine.ws.cnt.cpu().numpy()[0]
print(f"n={n} h={h}: kernel span {end.max():.1f} us; waves start {np.median(begin):.1f} (max {begin.max():.1f}); "
      f"items start {np.median(p1s):.1f} (max {p1s.max():.1f}), end median {np.median(p1e):.1f} p90 {np.percentile(p1e, 90):.1f} "
      f"max {p1e.max():.1f}; end max {end.max():.1f}")
print(f"items per wave: median {np.median(items):.0f} min {items.min()} max {items.max()}; exact-tier batches per wave median "
      f"{np.median(batches):.0f} max {batches.max()}")
dur, kind, nb = [], [], []
for w in range(waves):
    for i in range(min(int(items[w]), 8)):
        word, b, e = st[w, 8 + 3 * i: 11 + 3 * i]
        dur.append((e - b) * 0.01)
        kind.append(0)
        nb.append(word >> 32)
dur, kind, nb = np.array(dur), np.array(kind), np.array(nb)
print(f"{len(dur)} items: duration median {np.median(dur):.2f} p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} "
      f"max {dur.max():.2f} us")
for lo, hi in ((0, 2), (3, 4), (5, 8), (9, 32), (33, 10**6)):
    m = (nb >= lo) & (nb <= hi) & (kind == 0)
    if m.any():
        print(f"  items with {lo}..{hi} batches (not dense): {int(m.sum())}, duration median {np.median(dur[m]):.2f} max {dur[m].max():.2f} us")
busy = (p1e - p1s)
print(f"item time per wave / span: mean {busy.mean() / end.max():.2f}; waves idle at the final barrier: "
      f"mean wait {np.mean(p1e.reshape(-1, 16).max(axis=1, keepdims=True) - p1e.reshape(-1, 16)):.1f} us")
print(f"inlier counts: hypotheses with > 30 % inliers {int((cnt > 0.3 * n).sum())}, 10..30 % {int(((cnt > 0.1 * n) & (cnt <= 0.3 * n)).sum())}")
