"""Config C5 timing: 256 image pairs x 10k correspondences x 2000 hypotheses, E-estimation + cheirality
vote + triangulation end-to-end on one GPU (BASELINE.json configs[4])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from structure_from_motion_amd import batched, device as dev, synthetic
from structure_from_motion_amd._native import AGG_RMS
B, n, h = int(os.environ.get("B", 256)), int(os.environ.get("N", 10000)), int(os.environ.get("H", 2000))
rng = np.random.default_rng(0)
base = [synthetic.two_view_scene(n, seed=6 + b) for b in range(min(B, 16))]
K = base[0][2]
pa = dev.to_device(np.stack([base[b % len(base)][0] for b in range(B)]))
pb = dev.to_device(np.stack([base[b % len(base)][1] for b in range(B)]))
pipe = batched.TwoViewBatch(B, n, h)
for _ in range(2):
    pipe.run(pa, pb, K, seed=6, thr=1.5e-6, min_extra=10, aggregation=AGG_RMS)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for r in range(reps):
    pipe.run(pa, pb, K, seed=100 + r, thr=1.5e-6, min_extra=10, aggregation=AGG_RMS)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
res = pipe.results()
ok = sum(r.status == batched.OK for r in res)
pts = sum(len(r.points) for r in res if r.status == batched.OK)
print(f"C5: {B} pairs x {n} x {h}: {dt*1e3:.2f} ms per batch -> {B*n*h/dt/1e9:.1f} G correspondence-evals/s, "
      f"{B/dt:.0f} pairs/s; {ok}/{B} pairs OK, {pts} points triangulated", flush=True)
