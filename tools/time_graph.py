"""Eager launch sequence vs one captured HIP graph per RANSAC pass, on the launch-bound small workloads
(C1-like 300 x 2000, C2 5000 x 10000) and the compute-bound headline one.  Run on the GPU box:
    python tools/time_graph.py > gpurun_out/graph.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import device, distributed, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402


def run(engine, steps):
    for s in range(10):
        engine.step(100 + s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        engine.step(1000 + s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    device.require_gpu()
    for n, h, steps in [(300, 2000, 2000), (5000, 10000, 1000), (50000, 100000, 20)]:
        pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
        corr = device.normalize_correspondences(device.to_device(pa), device.to_device(pb), K)
        eager = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS)
        graphed = distributed.ShardedRansac(corr, h, 1.5e-6, 10, AGG_RMS)
        graphed.capture()
        te, tg = run(eager, steps), run(graphed, steps)
        te2, tg2 = run(eager, steps), run(graphed, steps)
        print(f"n={n} h={h}: eager {min(te, te2):.1f} us/pass, graph {min(tg, tg2):.1f} us/pass "
              f"({n * h / min(tg, tg2) * 1e6:.3e} evals/s)", flush=True)


if __name__ == "__main__":
    main()
