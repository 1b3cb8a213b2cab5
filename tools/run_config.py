"""One of the other BASELINE configurations, or one of the widened rows of SURVEY.md §8f, as a plain loop (for rocprofv3 passes;
tools/collect_config_counters.sh):
   python tools/run_config.py c2|c5|<widened workload> [passes]  — C2 = 5 000 x 10 000 (the lean small pass), C5 = 256 pairs x
10 000 x 2 000 (the batched pipeline), with bench.py's scenes and seeds; widened workloads: the names of
tools/widened_workloads.py::BUILDERS (f1_match_20000x20000_ncc9, f2_harris_vga, f4_refine_50000, pose_tail_c5, ...)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import batched, device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
THR, MIN_EXTRA = 1.5e-6, 10
if which == "c2":
    n, h = 5_000, 10_000
    pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
    corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
    eng = distributed.ShardedRansac(corr, h, THR, MIN_EXTRA, AGG_RMS)
    for r in range(passes):
        eng.step(1000 + r)
    torch.cuda.synchronize()
    print("c2", eng.outcome().best_h)
elif which != "c5":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import widened_workloads

    run, info = widened_workloads.BUILDERS[which]()
    for r in range(passes):
        run()
    torch.cuda.synchronize()
    print(which, {k: v for k, v in info.items() if k not in ("keepalive", "stages")})
else:
    B, n, h = 256, 10_000, 2_000
    base = [synthetic.two_view_scene(n, seed=300 + b, outlier_fraction=0.25) for b in range(16)]
    pix_a = device.to_device(np.stack([base[b % 16][0] for b in range(B)]))
    pix_b = device.to_device(np.stack([base[b % 16][1] for b in range(B)]))
    pipe = batched.TwoViewBatch(B, n, h)
    for r in range(passes):
        pipe.run(pix_a, pix_b, base[0][2], seed=70 + 1000 * r, thr=THR, min_extra=MIN_EXTRA, aggregation=AGG_RMS)
    torch.cuda.synchronize()
    print("c5", sum(r.status == batched.OK for r in pipe.results()))
