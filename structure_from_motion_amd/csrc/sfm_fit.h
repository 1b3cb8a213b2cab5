// Device routines of the normalised eight-point fit, shared by the per-hypothesis kernel (sfm_kernels.hip)
// and the refit-on-inliers kernel (sfm_refine.hip).  Reference: lib/epipolar/eight_point.py:136-170, 308-446.
#pragma once
#include "sfm_common.h"
#include "sfm_math.h"

namespace sfmfit {

struct Hartley {
    double scale, cx, cy;  // forward transform T = [[s,0,-s*cx],[0,s,-s*cy],[0,0,1]]
};

// reference eight_point.py:308-338 on 8 points, same operation order as NumPy:
// centroid = sequential sum / 8; mean norm = pairwise tree over 8 (NumPy's unrolled pairwise sum).
SFM_DEVICE Hartley hartley8(double (&x)[8], double (&y)[8]) {
    double sx = x[0], sy = y[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        sx += x[k];
        sy += y[k];
    }
    Hartley h;
    h.cx = sx / 8.0;
    h.cy = sy / 8.0;
    double nrm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        x[k] -= h.cx;
        y[k] -= h.cy;
        nrm[k] = sqrt(x[k] * x[k] + y[k] * y[k]);
    }
    const double total = ((nrm[0] + nrm[1]) + (nrm[2] + nrm[3])) + ((nrm[4] + nrm[5]) + (nrm[6] + nrm[7]));
    h.scale = sqrt(2.0) / (total / 8.0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        x[k] *= h.scale;
        y[k] *= h.scale;
    }
    return h;
}

// The four stages of the fit as device routines (shared by the hypothesis kernel and the single-problem
// stage kernel that backs the reference's private helpers).

// Y^T Y, upper triangle, accumulated in point order (eight_point.py:363-393)
SFM_DEVICE void build_yty(const double (&xa)[8], const double (&ya)[8], const double (&xb)[8],
                          const double (&yb)[8], double (&a)[45]) {
#pragma unroll
    for (int i = 0; i < 45; ++i) a[i] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double col[9] = {xb[k] * xa[k], xb[k] * ya[k], xb[k], yb[k] * xa[k], yb[k] * ya[k],
                               yb[k],         xa[k],         ya[k], 1.0};
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int q = p; q < 9; ++q) a[idx++] += col[p] * col[q];
    }
}

// eight_point.py:396-427: eigen-decomposition, degeneracy predicate (any but the smallest eigenvalue
// <= 1e-10), eigenvector of the eigenvalue of smallest magnitude.  `a` is destroyed.
SFM_DEVICE int null_vector_of_yty(double (&a)[45], double (&f)[9], double (&w)[9], double& second) {
    double v[81];
    sfm::jacobi_eig9(a, v, w);
    double smallest = w[0];
    second = INFINITY;
    double best_abs = fabs(w[0]);
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = v[k * 9 + 0];
#pragma unroll
    for (int j = 1; j < 9; ++j) {
        const bool lt = w[j] < smallest;
        second = lt ? smallest : fmin(second, w[j]);
        smallest = lt ? w[j] : smallest;
        const bool closer = fabs(w[j]) < best_abs;
        best_abs = closer ? fabs(w[j]) : best_abs;
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = closer ? v[k * 9 + j] : f[k];
    }
    return (second <= 1e-10) ? SFM_FIT_DEGENERATE : 0;
}

// Null vector of the 8x9 design matrix Y and the degeneracy predicate of eight_point.py:396-427, without forming
// Y^T Y: Householder QR of Y^T gives the null vector directly (the eigenvector of Y^T Y for its ~0 eigenvalue, to
// better accuracy than an eigen-solve of the squared matrix) and an upper-triangular R with the singular values of Y.
// The reference's predicate "second-smallest eigenvalue of Y^T Y <= 1e-10" is sigma_min(R)^2 <= 1e-10; it is decided
// by the rigorous bound sigma_min(R) >= 1 / ||R^-1||_F whenever that is conclusive (virtually always), and by a
// Jacobi SVD of R otherwise or when the caller wants lambda_2 itself.  sq (if computed) = squared singular values.
SFM_DEVICE int null_vector_of_design(const double (&xa)[8], const double (&ya)[8], const double (&xb)[8],
                                     const double (&yb)[8], bool need_second, double (&f)[9], double& second,
                                     double (&sq)[8]) {
#pragma clang fp contract(fast)   // QR, back substitution: no reference rounding to reproduce (sfm_math.h, hestenes_rotate)
    double col[8][9];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        col[k][0] = xb[k] * xa[k]; col[k][1] = xb[k] * ya[k]; col[k][2] = xb[k];
        col[k][3] = yb[k] * xa[k]; col[k][4] = yb[k] * ya[k]; col[k][5] = yb[k];
        col[k][6] = xa[k];         col[k][7] = ya[k];         col[k][8] = 1.0;
    }
    double rdiag[8];
    sfm::qr_null_vector(col, rdiag, f);
    // ||R^-1||_F^2 column by column: solve R x = e_j by back substitution (x has j+1 non-zeros).  The 36 pivots are
    // divided out through 8 reciprocals: the result only feeds a bound that is tested with a 1e-6 margin, and 28 IEEE
    // divisions fewer shorten every lane's dependent chain (the fit is latency-bound: one wave per SIMD)
    double inv_diag[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) inv_diag[i] = sfm::rcp_newton(rdiag[i]);
    double fro2 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double x[8];
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            if (i > j) {
                x[i] = 0.0;
            } else {
                double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k > i && k <= j) acc -= col[k][i] * x[k];  // R(i,k) = col[k][i] for i < k
                x[i] = acc * inv_diag[i];
                fro2 += x[i] * x[i];
            }
        }
    }
    const double lower = 1.0 / fro2;                          // <= sigma_min(R)^2 = lambda_2
    const bool conclusive = lower > 1e-10 * (1.0 + 1e-6);     // NaN / inf (rank-deficient R) -> not conclusive
    int flag = 0;
    second = lower;
#pragma unroll
    for (int k = 0; k < 8; ++k) sq[k] = 0.0;
    if (need_second || !__all(conclusive)) {  // wave-uniform
        double g[8][8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k) g[c][k] = (k < c) ? col[c][k] : ((k == c) ? rdiag[c] : 0.0);
        sfm::singular_values_sq<8>(g, sq);
        double smallest = sq[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) smallest = fmin(smallest, sq[k]);
        // fmin drops NaN: a NaN anywhere in the design (NaN input) must flag the sample
        const bool poisoned = !(fro2 == fro2) && !(smallest > 1e-10);
        second = smallest;
        flag = (smallest <= 1e-10 || poisoned) ? SFM_FIT_DEGENERATE : 0;
    }
    return flag;
}

// ---- N-point variant (refit on many inliers): Y^T Y is symmetric positive semi-definite and, after Hartley
// normalisation of many points, has one small eigenvalue well separated from the rest.  Its eigenvector is found
// by inverse iteration on a Cholesky factor (a few hundred flops) instead of a full Jacobi eigen-decomposition with
// eigenvectors (tens of thousands), and the reference's degeneracy predicate "second-smallest eigenvalue <= 1e-10"
// (eight_point.py:415-421) by Sylvester's law of inertia: the number of negative pivots of the LDL^T factorisation of
// A - 1e-10 I is the number of eigenvalues below 1e-10.
SFM_DEVICE constexpr int tri9(int p, int q) { return p * 9 - p * (p - 1) / 2 + (q - p); }  // p <= q, packed upper triangle

SFM_DEVICE int eigenvalues_below9(const double (&a)[45], double tau) {
    double l[45], d[9];  // unit lower factor stored at tri9(j, i) for j < i
    int negative = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double dj = a[tri9(j, j)] - tau;
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= l[tri9(k, j)] * l[tri9(k, j)] * d[k];
        if (dj == 0.0) dj = -1e-300;  // a zero pivot: an eigenvalue AT tau counts as below (the predicate is <=)
        d[j] = dj;
        negative += (dj < 0.0) ? 1 : 0;
#pragma unroll
        for (int i = j + 1; i < 9; ++i) {
            double v = a[tri9(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= l[tri9(k, i)] * l[tri9(k, j)] * d[k];
            l[tri9(j, i)] = v / dj;
        }
    }
    return negative;  // NaN pivots compare false: a NaN matrix counts nothing here and is caught by the caller
}

// Unit eigenvector of the smallest eigenvalue of the PSD matrix `a` (packed upper triangle).  Wave-uniform loop.
SFM_DEVICE void smallest_eigenvector_psd9(const double (&a)[45], double (&x)[9]) {
    double trace = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) trace += a[tri9(j, j)];
    const double sigma = fmax(trace * 1e-15, 1e-300);  // keeps the factorisation positive when an eigenvalue is ~0
    double l[45], inv[9];                               // Cholesky factor of a + sigma I at tri9(j, i), j <= i
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double dj = a[tri9(j, j)] + sigma;
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= l[tri9(k, j)] * l[tri9(k, j)];
        dj = fmax(dj, sigma);
        const double ljj = sqrt(dj);
        l[tri9(j, j)] = ljj;
        inv[j] = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < 9; ++i) {
            double v = a[tri9(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= l[tri9(k, i)] * l[tri9(k, j)];
            l[tri9(j, i)] = v * inv[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) x[j] = 1.0 / 3.0;
    for (int iteration = 0; iteration < 200; ++iteration) {
        double y[9], z[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {  // L y = x
            double v = x[i];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= l[tri9(k, i)] * y[k];
            y[i] = v * inv[i];
        }
#pragma unroll
        for (int i = 8; i >= 0; --i) {  // L^T z = y
            double v = y[i];
#pragma unroll
            for (int k = i + 1; k < 9; ++k) v -= l[tri9(i, k)] * z[k];
            z[i] = v * inv[i];
        }
        double norm2 = 0.0, dot = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            norm2 += z[j] * z[j];
            dot += z[j] * x[j];
        }
        const double scale = ((dot < 0.0) ? -1.0 : 1.0) / sqrt(norm2);
        double change = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double next = z[j] * scale;
            change = fmax(change, fabs(next - x[j]));
            x[j] = next;
        }
        if (!(change > 1e-16)) break;  // converged (or NaN: nothing more to do)
    }
}

// Exact scaling of a matrix of arbitrary magnitude into the range the Jacobi routines are written for: the power of two
// that brings the largest entry into [0.5, 1) (their convergence test compares squared quantities, which under- or
// overflow for entries beyond ~1e+-77; the eight-point fit's own matrices are unit-norm, a caller's E or F need not be).
// Multiplying by it and dividing the results back is exact.  1 for a zero / non-finite matrix.
SFM_DEVICE double pow2_unit_scale(const double* m9) {
    double largest = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) largest = fmax(largest, fabs(m9[k]));
    if (!(largest > 0.0) || !(largest < INFINITY)) return 1.0;
    int exponent;
    (void)frexp(largest, &exponent);
    return ldexp(1.0, -exponent);
}

// rank-2 enforcement (eight_point.py:430-446): drop the smallest singular direction of f (row-major 3x3).
// One-sided Jacobi on the columns of F leaves G = F V with orthogonal columns g_k = sigma_k u_k; the projection is
//   F_r = sum over the two columns kept of  u_k u_k^T F = g_k (g_k^T F) / |g_k|^2,
// which needs the two LARGE columns only (their directions are accurate to a few ulps however small sigma_3 is) and
// no V: the rotations do not have to be accumulated (12 of the ~60 instructions of each, ~20 rotations per fit).
// F_r has rank <= 2 by construction.
// `ratio2` (optional) receives (sigma_3 / sigma_1)^2 of f: how far the unconstrained estimate was from rank 2.
SFM_DEVICE void enforce_rank2(const double (&f)[9], double (&fr)[3][3], double* ratio2 = nullptr) {
#pragma clang fp contract(fast)
    double g[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) g[c][r] = f[r * 3 + c];
    sfm::hestenes_orthogonalise<3>(g);
    double n2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) n2[c] = g[c][0] * g[c][0] + g[c][1] * g[c][1] + g[c][2] * g[c][2];
    const int drop = (n2[0] <= n2[1] && n2[0] <= n2[2]) ? 0 : ((n2[1] <= n2[2]) ? 1 : 2);
    const double smallest = fmin(n2[0], fmin(n2[1], n2[2])), largest = fmax(n2[0], fmax(n2[1], n2[2]));
    if (ratio2 != nullptr) *ratio2 = smallest / largest;
    // The direction of a kept column is accurate to ~eps sigma_1 / sigma_k: fine for any F an eight-point sample of
    // real correspondences gives (sigma_2 ~ sigma_1 after Hartley normalisation), not for a nearly rank-ONE matrix.
    // There (sigma_2 < 1e-6 sigma_1 in some lane: wave-uniform, practically never) the wave takes the route that
    // accumulates V, whose error is eps sigma_1 whatever the singular values are.
    const double middle = ((n2[0] + n2[1]) + n2[2]) - (smallest + largest);
    if (__any(!(middle > 1e-12 * largest))) {
        double vv[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) g[c][r] = f[r * 3 + c];
        sfm::hestenes_svd<3>(g, vv);
        double m2[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) m2[c] = g[c][0] * g[c][0] + g[c][1] * g[c][1] + g[c][2] * g[c][2];
        const int out = (m2[0] <= m2[1] && m2[0] <= m2[2]) ? 0 : ((m2[1] <= m2[2]) ? 1 : 2);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc += (k == out) ? 0.0 : g[k][r] * vv[k][c];
                fr[r][c] = acc;
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) fr[r][c] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // w = g_k^T F / |g_k|^2 (a row vector); the dropped column contributes nothing
        const double scale = (k == drop || !(n2[k] > 0.0)) ? 0.0 : 1.0 / n2[k];   // a zero column has nothing to project on
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            w[c] = ((g[k][0] * f[0 * 3 + c] + g[k][1] * f[1 * 3 + c]) + g[k][2] * f[2 * 3 + c]) * scale;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) fr[r][c] += g[k][r] * w[c];
    }
}

// E = T2^T F T1 (eight_point.py:163) for forward Hartley transforms t1 (first image), t2 (second image)
SFM_DEVICE void unnormalise(const double (&fr)[3][3], const Hartley& t1, const Hartley& t2, double (&e)[9]) {
#pragma clang fp contract(fast)
    const double tx1 = -t1.scale * t1.cx, ty1 = -t1.scale * t1.cy;
    const double tx2 = -t2.scale * t2.cx, ty2 = -t2.scale * t2.cy;
    double m[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m[0][c] = t2.scale * fr[0][c];
        m[1][c] = t2.scale * fr[1][c];
        m[2][c] = (tx2 * fr[0][c] + ty2 * fr[1][c]) + fr[2][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        e[r * 3 + 0] = m[r][0] * t1.scale;
        e[r * 3 + 1] = m[r][1] * t1.scale;
        e[r * 3 + 2] = (m[r][0] * tx1 + m[r][1] * ty1) + m[r][2];
    }
}

// ransac.py:96-108 over the 8 sample points plus `count` extra inliers (sum1 = sum e, sum2 = sum e^2)
SFM_DEVICE double aggregate_error(int aggregation, int count, double sum1, double sum2) {
    const double nn = (double)(count + 8);
    switch (aggregation) {
        case SFM_AGG_SUM: return sum1;
        case SFM_AGG_SQUARE: return sum2;
        case SFM_AGG_MEAN: return sum1 / nn;
        default: return sqrt(sum2 / nn);
    }
}

}  // namespace sfmfit
