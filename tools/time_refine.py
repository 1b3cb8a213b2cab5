"""Timing of the local-optimisation kernel (sfm_refine_inliers) on winners of real RANSAC passes.
    python tools/time_refine.py > gpurun_out/refine.txt
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import device, synthetic  # noqa: E402
from structure_from_motion_amd._native import AGG_RMS  # noqa: E402


def main():
    device.require_gpu()
    for B, n, h, min_extra in [(1, 5000, 2000, 500), (1, 50000, 2000, 5000), (256, 10000, 500, 1000)]:
        scenes = [synthetic.two_view_scene(n, seed=6 + b) for b in range(min(B, 4))]
        pa = device.to_device(np.stack([scenes[b % len(scenes)][0] for b in range(B)]))
        pb = device.to_device(np.stack([scenes[b % len(scenes)][1] for b in range(B)]))
        corr = device.normalize_correspondences(pa, pb, scenes[0][2])
        ws = device.RansacWorkspace(B, n, h)
        device.sample_philox(5, 0, h, n, batch=B, out=ws.S)
        ws.run(corr, 1.5e-6, min_extra, AGG_RMS)
        best = ws.result[:, 1].clamp(min=0)
        E = ws.E[torch.arange(B, device=best.device), best].contiguous()
        err = ws.result.view(torch.float64)[:, 2].contiguous()
        for rounds in (1, 4):
            for _ in range(3):
                out = device.refine_inliers(corr, E, ws.mask, err, 1.5e-6, AGG_RMS, rounds)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                out = device.refine_inliers(corr, E, ws.mask, err, 1.5e-6, AGG_RMS, rounds)
            b.record()
            torch.cuda.synchronize()
            info = device.read_refine_info(out[2])
            before = int((ws.mask[0] != 0).sum())
            print(f"B={B} n={n} rounds={rounds}: {a.elapsed_time(b) / 20 * 1e3:.1f} us/launch; pair 0: "
                  f"{before} -> {info[0][1]} inliers, {info[0][2]} refits accepted", flush=True)


if __name__ == "__main__":
    main()
