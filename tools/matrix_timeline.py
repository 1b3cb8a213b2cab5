"""Diagnostic (needs a build with -DSFM_MATRIX_STAMPS=1 — e.g. `python tools/ab.py prepare stamps --flags=-DSFM_MATRIX_STAMPS=1`
and PYTHONPATH=tools/tmp/trees/stamps): where a wave of the matrix-pipe scoring kernel spends its time —
s_memrealtime stamps (100 MHz: 10 ns ticks) at wave start, operands loaded, step loop done, queues drained, sample points
fixed, results handed off — and how the waves are spread over time and over the chip.  N, H, THR, SFM_SCORE_SPLIT from the
environment."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("SFM_TREE"):   # a tree prepared by tools/ab.py
    sys.path.insert(0, os.path.abspath(os.environ["SFM_TREE"]))
from structure_from_motion_amd import _native, device as dev, synthetic  # noqa: E402

n, h = int(os.environ.get("N", 50000)), int(os.environ.get("H", 100000))
thr = float(os.environ.get("THR", 1.5e-6))
lib = _native.load()
lib.sfm_debug_read_matrix_stamps.argtypes = [C.c_void_p, C.c_int64]
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
corr = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).reshape(1, n, 4)
S = dev.sample_philox(5, 0, h, n)
E, flags = dev.fit_eight_point(corr, S)
ws = dev.score_workspace(n, h, 1, corr.device)   # (sized for the process-wide options: the SFM_SCORE_* variables)
before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
before.record(); after.record(); torch.cuda.synchronize()
for _ in range(3):
    cnt, s1, s2 = dev.score_sed(corr, E, S, thr, workspace=ws, options=dev.default_score_options().with_timing(before, after))
torch.cuda.synchronize()
kernel_ms = before.elapsed_time(after)
st = np.zeros((65536, 10), dtype=np.uint64)
assert lib.sfm_debug_read_matrix_stamps(st.ctypes.data, 65536) == 0
st = st[st[:, 0] != 0]
if os.environ.get("STAMPS_OUT"):
    np.save(os.environ["STAMPS_OUT"], st)
waves = len(st)
t0 = st[:, 0].min()
T = (st[:, :6].astype(np.int64) - np.int64(t0)).astype(np.float64) * 0.01   # us
span = T[:, 5].max()
names = ["operand loads (order -> E, B operands)", "step loop (tier 1 + rounds inside)", "final drain of the queues",
         "sample fix-up", "lane fold + hand-off"]
print(f"n={n} h={h} thr={thr:g}: {waves} waves stamped, kernel {kernel_ms * 1e3:.0f} us by events, span of the stamps {span:.0f} us")
dur = T[:, 5] - T[:, 0]
print(f"wave lifetime: median {np.median(dur):.1f} us, p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}; "
      f"sum / span = {dur.sum() / span:.0f} waves resident on average (capacity 4096 at 4 per SIMD)")
for k, name in enumerate(names):
    d = T[:, k + 1] - T[:, k]
    print(f"  {name:45s} median {np.median(d):8.2f} us  mean {d.mean():8.2f}  p90 {np.percentile(d, 90):8.2f}  max {d.max():8.2f}  "
          f"share of all wave time {d.sum() / dur.sum():.3f}")
unit = (st[:, 7] >> np.uint64(32)).astype(np.int64)
h0 = (st[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64)
slots = (st[:, 8] >> np.uint64(32)).astype(np.float64) * 64.0
pops = (st[:, 8] & np.uint64(0xFFFFFFFF)).astype(np.float64)
clock = st[:, 9].astype(np.float64) / np.maximum(dur, 1e-3) * 1e-3   # shader cycles per us -> GHz
print(f"shader clock while a wave runs: median {np.median(clock):.2f} GHz, p10 {np.percentile(clock, 10):.2f}, p90 {np.percentile(clock, 90):.2f}")
true_inliers = float(cnt.sum().item())
print(f"true inliers (sum of the counts): {true_inliers:.4g} = {true_inliers / max(pops.sum(), 1):.3f} of the points popped; the rest passed tier 1 and failed the exact test")
print(f"exact tier: {pops.sum():.4g} points popped in {slots.sum():.4g} lane slots -> lane utilisation {pops.sum() / max(slots.sum(), 1):.3f}")
edges = [0, 2048, 4096, 6144, 8192, 16384, 32768, 65536, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (h0 >= lo) & (h0 < hi)
    if m.any():
        print(f"  order slots {lo}-{min(hi, h)}: {int(m.sum())} items, lifetime median {np.median(dur[m]):7.1f} us (sum {dur[m].sum() / 1e3:7.1f} ms), "
              f"popped {pops[m].sum():.3g} ({pops[m].sum() / max(pops.sum(), 1):.3f} of all), utilisation {pops[m].sum() / max(slots[m].sum(), 1):.3f}, "
              f"drain median {np.median((T[:, 3] - T[:, 2])[m]):.1f} us, loop {np.median((T[:, 2] - T[:, 1])[m]):.1f}")
for u in sorted(set(unit.tolist()))[:3] + [int(unit.max())]:
    m = unit == u
    print(f"  range {u}: {m.sum()} waves, lifetime median {np.median(dur[m]):.1f} us; sample fix-up median {np.median((T[:, 4] - T[:, 3])[m]):.1f}")
xcc = (st[:, 6] >> np.uint64(32)).astype(np.int64) & 0xF
hw = st[:, 6].astype(np.int64) & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
print("waves per XCC:", np.bincount(xcc, minlength=8).tolist())
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
counts = np.bincount(key)
counts = counts[counts > 0]
print(f"SIMDs used: {len(counts)}; waves per SIMD: min {counts.min()}, median {int(np.median(counts))}, max {counts.max()}")
# start-time profile: how fast are waves launched, how full is the chip over time
order = np.argsort(T[:, 0])
starts = T[order, 0]
print("start times (us) of the k-th wave: " + ", ".join(f"{k}: {starts[min(k, waves - 1)]:.1f}" for k in (0, 1023, 4095, 8191, 16383, waves - 1)))
edges = np.linspace(0, span, 11)
resident = [(int(((T[:, 0] <= e) & (T[:, 5] > e)).sum())) for e in edges[:-1] + np.diff(edges) / 2]
print("waves resident at the middle of each tenth of the span:", resident)
# per SIMD: gap between one wave's end and the next wave's start on a busy SIMD
gaps = []
for k in np.unique(key)[:64]:
    m = np.nonzero(key == k)[0]
    ends = np.sort(T[m, 5]); begins = np.sort(T[m, 0])
    busy = sum(T[m, 5] - T[m, 0])
    gaps.append(busy / span)
print(f"first 64 SIMDs: summed wave lifetime / span = median {np.median(gaps):.2f} (waves resident per SIMD on average)")
