"""Time one BASELINE configuration as a plain loop of passes, from ANY source tree of this repository (``--tree``: a checkout
of an earlier commit or a build with experiment flags, prepared by tools/ab.py) — the unit of every same-box A/B of round 5,
and the program rocprofv3 wraps for the per-kernel tables under profiles/.

    python tools/steptime.py --config c3 [--tree DIR] [--steps 50] [--warmup 5] [--options kernel=matrix,split=8]

Configs (bench.py's scenes and seeds): c3 = 50 000 x 100 000 (the bench workload), c4 = 50 000 x 125 000 (one of eight ranks'
share of C4), c4x4 / c4x2 = x 250 000 / x 500 000 (the share at 4 / 2 ranks), c2 = 5 000 x 10 000 (lean small pass),
c5 = 256 pairs x 10 000 x 2 000 (batched pipeline), mid = 20 000 x 40 000, wide = 50 000 x 20 000.
Prints one JSON line: ms per pass (wall clock over the timed passes, one synchronisation at each end) and, for c3-like
configs, the winner (so that two trees can be seen to agree)."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--tree", default=None, help="root of the source tree to import from (default: this repository)")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--options", default="", help="process-wide scoring options, e.g. kernel=matrix,split=8,persistent=1")
ap.add_argument("--thr", type=float, default=1.5e-6, help="inlier threshold (1e-14: nothing survives tier 1 — the filter alone)")
args = ap.parse_args()
root = os.path.abspath(args.tree) if args.tree else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from structure_from_motion_amd import batched, device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

THR, MIN_EXTRA = args.thr, 10
device.require_gpu()
if args.options:
    fields = {}
    for item in args.options.split(","):
        k, v = item.split("=")
        fields[k] = v if k == "kernel" else int(v)
    device.set_default_score_options(device.ScoreOptions(**fields))

SINGLE = {"c3": (50_000, 100_000), "c4": (50_000, 125_000), "c4x4": (50_000, 250_000), "c4x2": (50_000, 500_000),
          "c2": (5_000, 10_000), "mid": (20_000, 40_000), "wide": (50_000, 20_000), "c1": (300, 2_000),
          "e2": (50_000, 40_000), "e3": (50_000, 60_000), "e2n": (20_000, 100_000), "e1n": (100_000, 15_000),
          "f2": (100_000, 20_000), "f4": (100_000, 40_000), "g2": (200_000, 10_000), "g2n": (10_000, 200_000), "g3": (150_000, 20_000)}   # 2, 3, 2, 1.5 x 10^9 evaluations
out = {"config": args.config, "tree": root, "steps": args.steps}
if args.config in SINGLE:
    n, h = SINGLE[args.config]
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    eng = distributed.ShardedRansac(corr, h, THR, MIN_EXTRA, AGG_RMS)
    run = lambda r: eng.step(1000 + r)  # noqa: E731
    finish = lambda: {"best_h": eng.outcome().best_h, "inliers": int((eng.outcome().mask != 0).sum())}  # noqa: E731
    evals = float(n) * h
elif args.config == "c5":
    B, n, h = 256, 10_000, 2_000
    base = [synthetic.two_view_scene(n, seed=300 + b, outlier_fraction=0.25) for b in range(16)]
    pix_a = device.to_device(np.stack([base[b % 16][0] for b in range(B)]))
    pix_b = device.to_device(np.stack([base[b % 16][1] for b in range(B)]))
    pipe = batched.TwoViewBatch(B, n, h)
    run = lambda r: pipe.run(pix_a, pix_b, base[0][2], seed=70 + 1000 * r, thr=THR, min_extra=MIN_EXTRA,  # noqa: E731
                             aggregation=AGG_RMS)
    finish = lambda: {"pairs_ok": sum(r.status == batched.OK for r in pipe.results()),  # noqa: E731
                      "inliers": int(sum(len(r.inlier_order) for r in pipe.results() if r.inlier_order is not None))}
    evals = float(B) * n * h
else:
    raise SystemExit(f"unknown config {args.config}")

for r in range(args.warmup):
    run(r - 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(args.steps):
    run(r)
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / max(args.steps, 1)
out.update({"ms_per_pass": sec * 1e3, "evals_per_s": evals / sec if sec > 0 else None})
out.update(finish())
print(json.dumps(out), flush=True)
