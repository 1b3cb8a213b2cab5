"""Soak test of the two-tier scoring kernel against the all-fp64 kernel (counts must be identical, sums equal to
summation order): thousands of random (matrix family, scale, coordinate range, size) combinations with the threshold
placed ON the SED distribution (quantiles of real SED values, values +- a few ulps), where the fp32 tier's bounds are
actually exercised.  Product code only (no oracle).  Usage: python tools/soak_filter.py [trials] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import device as dev, synthetic  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev.require_gpu()
    rng = np.random.default_rng(seed)
    n_max = int(os.environ.get("SOAK_N_MAX", 20000))   # (>= 32768 with SFM_SCORE_SPLIT=8: the scoring launch replays the cost pre-pass)
    pa, pb, K, *_ = synthetic.two_view_scene(n_max, seed=6)
    corr_full = dev.normalize_correspondences(dev.to_device(pa), dev.to_device(pb), K).cpu().numpy()
    checked, boundary, t0 = 0, 0, time.time()
    for trial in range(trials):
        # SOAK_N_MIN / SOAK_H_MIN / SOAK_H_MAX move the size ranges (e.g. thousands of hypotheses: cost order, ranges of the points)
        n = int(rng.integers(int(os.environ.get("SOAK_N_MIN", 8)), n_max))
        h = int(rng.integers(int(os.environ.get("SOAK_H_MIN", 1)), int(os.environ.get("SOAK_H_MAX", 40))))
        corr = corr_full[rng.permutation(len(corr_full))[:n]].copy()
        if trial % 7 == 0:
            corr *= float(10.0 ** rng.uniform(0, 3))          # pixel-like coordinate ranges
        if trial % 13 == 0:
            corr[:, :2] *= float(10.0 ** rng.uniform(-3, 3))  # lopsided: image a and b on different scales
        corr_d = dev.to_device(corr).reshape(1, n, 4)
        S = dev.sample_philox(trial, 0, h, n)
        family = trial % 6
        if family in (0, 1):                                  # fitted hypotheses
            E = dev.fit_eight_point(corr_d, S)[0].cpu().numpy().reshape(h, 3, 3)
        else:
            E = rng.normal(size=(h, 3, 3))
            if family == 2:
                E[:, :, 2] *= 1e-6
            elif family == 3:
                E *= 10.0 ** rng.integers(-8, 9, size=(h, 3, 3))
            elif family == 4:
                E[:, 1] = E[:, 0] * (1.0 + 1e-9 * rng.normal(size=(h, 1)))
            else:                                             # lopsided rows vs columns: one-sided bound at its weakest
                E[:, :2, :] *= 10.0 ** rng.uniform(-6, 6)
            E[:, 2, 2] = 1.0
        E = E * 10.0 ** float(rng.integers(-30, 31))
        E = np.nan_to_num(E, nan=1.0, posinf=1e300, neginf=-1e300)
        E_d = dev.to_device(E.reshape(1, h, 9))
        # thresholds from the SED distribution of one of the hypotheses
        k = int(rng.integers(0, h))
        sed = dev.sed_values(corr_d.reshape(n, 4), dev.to_device(E[k].reshape(9))).cpu().numpy()
        sed = sed[np.isfinite(sed) & (sed > 0)]
        if len(sed) == 0:
            continue
        picks = [float(np.quantile(sed, q)) for q in rng.uniform(0, 1, 2)] + [float(rng.choice(sed))]
        v = float(rng.choice(sed))
        picks += [float(np.nextafter(v, 0.0)), float(np.nextafter(v, np.inf))]
        for thr in picks:
            exact = dev.score_sed(corr_d, E_d, S, thr, exact_only=True)
            filt = dev.score_sed(corr_d, E_d, S, thr)
            ce, cf = exact[0].cpu().numpy(), filt[0].cpu().numpy()
            if not np.array_equal(ce, cf):
                bad = np.nonzero(ce != cf)[1]
                print(f"MISMATCH trial {trial} family {family} n {n} h {h} thr {thr!r}: hyps {bad[:5]} "
                      f"exact {ce[0, bad[:5]]} filtered {cf[0, bad[:5]]}", flush=True)
                np.savez("gpurun_out/soak_failure.npz", corr=corr, E=E, S=S.cpu().numpy(), thr=thr)
                sys.exit(1)
            for a, b in ((exact[1], filt[1]), (exact[2], filt[2])):
                a, b = a.cpu().numpy(), b.cpu().numpy()
                ok = np.isfinite(a) & np.isfinite(b)
                if not np.allclose(a[ok], b[ok], rtol=1e-12, atol=0):
                    print(f"SUM MISMATCH trial {trial} thr {thr!r}", flush=True)
                    sys.exit(1)
            checked += n * h
            boundary += int(ce.sum())
        if trial % int(os.environ.get("SOAK_REPORT", 200)) == 0:
            print(f"trial {trial}: {checked:.3e} evaluations compared, {boundary:.3e} inliers, {time.time() - t0:.0f} s",
                  flush=True)
    print(f"OK: {trials} trials, {checked:.3e} evaluations, counts identical everywhere ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
