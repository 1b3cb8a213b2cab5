"""CPU study (NumPy only, no GPU, no oracle): how many (point, hypothesis) pairs would pass tier 1 of the matrix-pipe scoring kernel
if the denominator chain (13 of the 40 operand slots: csrc/sfm_score_matrix.h) were replaced by something cheaper — the question
behind "two matrix instructions per step instead of three" (32 slots).  Hypotheses: eight-point fits of random samples of the bench
scene (plain SVD fits: the distribution matters here, not the bits).

    python tools/filter_bound_study.py [hypotheses=600]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from structure_from_motion_amd import synthetic  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n = 50_000
pa, pb, K, *_ = synthetic.two_view_scene(n, seed=6)
Ki = np.linalg.inv(K)
a = (Ki @ np.column_stack([pa, np.ones(n)]).T).T
b = (Ki @ np.column_stack([pb, np.ones(n)]).T).T
xa, ya, xb, yb = a[:, 0], a[:, 1], b[:, 0], b[:, 1]
rng = np.random.default_rng(5)
thr, kappa = 1.5e-6, 1.0 / 32.0
T = thr * (1 + kappa) * (1 + 1 / 1024)
names = ["true inliers", "now: 4 r^2 / (dA + dB), 12 + 1 slots", "squares + linear terms, cross terms by AM-GM: 8 + 1 slots",
         "linear terms, quadratic part at its data-set maximum: 4 + 1 slots", "one constant per hypothesis: no slots", "one-sided r^2 / dB: 6 slots"]
tot = np.zeros(len(names), dtype=np.int64)
for _ in range(H):
    s = rng.choice(n, 8, replace=False)
    Y = np.column_stack([xb[s] * xa[s], xb[s] * ya[s], xb[s], yb[s] * xa[s], yb[s] * ya[s], yb[s], xa[s], ya[s], np.ones(8)])
    e = np.linalg.svd(Y)[2][-1].reshape(3, 3)
    U_, S_, Vt = np.linalg.svd(e)
    e = (U_ @ np.diag([1.0, 1.0, 0.0]) @ Vt).ravel()
    la0, la1, la2 = e[0] * xa + e[1] * ya + e[2], e[3] * xa + e[4] * ya + e[5], e[6] * xa + e[7] * ya + e[8]
    r = xb * la0 + yb * la1 + la2
    lb0, lb1 = e[0] * xb + e[3] * yb + e[6], e[1] * xb + e[4] * yb + e[7]
    dA, dB = la0 ** 2 + la1 ** 2, lb0 ** 2 + lb1 ** 2
    U = (dA + dB) / 4
    lin = (e[2] ** 2 + e[5] ** 2 + 2 * (e[0] * e[2] + e[3] * e[5]) * xa + 2 * (e[1] * e[2] + e[4] * e[5]) * ya
           + e[6] ** 2 + e[7] ** 2 + 2 * (e[0] * e[6] + e[1] * e[7]) * xb + 2 * (e[3] * e[6] + e[4] * e[7]) * yb)
    Qa = (e[0] * xa + e[1] * ya) ** 2 + (e[3] * xa + e[4] * ya) ** 2
    Qb = (e[0] * xb + e[3] * yb) ** 2 + (e[1] * xb + e[4] * yb) ** 2
    ca, cb = abs(e[0] * e[1] + e[3] * e[4]), abs(e[0] * e[3] + e[1] * e[4])
    Uq = (lin + (e[0] ** 2 + e[3] ** 2 + ca) * xa ** 2 + (e[1] ** 2 + e[4] ** 2 + ca) * ya ** 2
          + (e[0] ** 2 + e[1] ** 2 + cb) * xb ** 2 + (e[3] ** 2 + e[4] ** 2 + cb) * yb ** 2) / 4
    r2 = r * r
    tot += [(r2 * (1 / dA + 1 / dB) <= thr).sum(), (r2 <= T * U).sum(), (r2 <= T * Uq).sum(),
            (r2 <= T * (lin + Qa.max() + Qb.max()) / 4).sum(), (r2 <= T * U.max()).sum(), (r2 <= T * dB).sum()]
print(f"{H} hypotheses x {n} points, thr {thr:g}; pairs that pass the test (the error slack of the 16-bit operands left out)")
for name, v in zip(names, tot):
    print(f"  {name:70s} {v:10d}   {v / tot[0]:.3f} x the true inliers   {v / tot[1]:.3f} x today's")
