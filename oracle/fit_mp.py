"""CPU ORACLE helper (test infrastructure): the eight-point fit of oracle/sfm_oracle.py evaluated in
multi-precision arithmetic (mpmath), as a tie-breaker between two double-precision implementations
(LAPACK in the oracle, Jacobi on the GPU).  Same algorithm as reference eight_point.py:136-170."""
import mpmath as mp
import numpy as np


def fit_eight_point_mp(ca: np.ndarray, cb: np.ndarray, dps: int = 40) -> np.ndarray:
    """ca, cb: (8,2) float64 coordinates -> E (3,3) as float64 rounded from a `dps`-digit computation."""
    with mp.workdps(dps):
        def hartley(c):
            pts = [(mp.mpf(float(x)), mp.mpf(float(y))) for x, y in c]
            cx = sum(p[0] for p in pts) / 8
            cy = sum(p[1] for p in pts) / 8
            cen = [(p[0] - cx, p[1] - cy) for p in pts]
            scale = mp.sqrt(2) / (sum(mp.sqrt(x * x + y * y) for x, y in cen) / 8)
            T = mp.matrix([[scale, 0, -scale * cx], [0, scale, -scale * cy], [0, 0, 1]])
            return [(x * scale, y * scale) for x, y in cen], T

        na, T1 = hartley(ca)
        nb, T2 = hartley(cb)
        Y = mp.matrix(8, 9)
        for i in range(8):
            xa, ya = na[i]
            xb, yb = nb[i]
            row = [xb * xa, xb * ya, xb, yb * xa, yb * ya, yb, xa, ya, mp.mpf(1)]
            for j in range(9):
                Y[i, j] = row[j]
        A = Y.T * Y
        w, V = mp.eigsy(A)
        k = min(range(9), key=lambda i: abs(w[i]))
        F = mp.matrix(3, 3)
        for r in range(3):
            for c in range(3):
                F[r, c] = V[3 * r + c, k]
        U, S, Vh = mp.svd_r(F)
        S2 = mp.diag([S[0], S[1], 0])
        Fr = U * S2 * Vh
        E = T2.T * Fr * T1
        E = E / E[2, 2]
        second = sorted(w)[1]
        return np.array([[float(E[r, c]) for c in range(3)] for r in range(3)]), float(second)
