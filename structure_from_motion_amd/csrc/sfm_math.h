// fp64 device primitives for the epipolar hot path (gfx950).
//
// Everything here is per-lane register math: all array indices are compile-time constants after
// unrolling, so the small matrices live in VGPRs (no scratch).  The translation unit is compiled with
// -ffp-contract=off: every multiply and add rounds separately, matching the NumPy elementwise
// semantics of the reference; fused operations are written explicitly with fma() where wanted, and the routines
// that have no reference rounding to reproduce (QR, Jacobi rotations, DLT null vectors: converged iterations checked
// by tolerance) re-enable contraction for their own bodies with `#pragma clang fp contract(fast)`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#define SFM_DEVICE __device__ __forceinline__
#ifndef SFM_SED_EXACT_DIVISION
#define SFM_SED_EXACT_DIVISION 0
#endif

namespace sfm {

// --------------------------------------------------------------------------------------------------
// Symmetric epipolar distance of one correspondence under one E (row-major e[9]).
// Operation order is the contract shared with oracle/sfm_oracle.py::sed_values:
//   line_b = E^T b, r = line_b . a   (reference sed.py:21,25:  coord_b.T @ e @ coord_a)
//   line_a = E a                      (sed.py:24)
//   sed = (1/(la0^2+la1^2) + 1/(lb0^2+lb1^2)) * r^2   (sed.py:27-29)
// --------------------------------------------------------------------------------------------------
SFM_DEVICE double sed_value(const double e[9], double xa, double ya, double xb, double yb) {
    const double lb0 = (xb * e[0] + yb * e[3]) + e[6];
    const double lb1 = (xb * e[1] + yb * e[4]) + e[7];
    const double lb2 = (xb * e[2] + yb * e[5]) + e[8];
    const double r = (lb0 * xa + lb1 * ya) + lb2;
    const double la0 = (e[0] * xa + e[1] * ya) + e[2];
    const double la1 = (e[3] * xa + e[4] * ya) + e[5];
    const double da = la0 * la0 + la1 * la1;
    const double db = lb0 * lb0 + lb1 * lb1;
    return (1.0 / da + 1.0 / db) * (r * r);
}

// The same distance for a caller that thresholds it and sums it (the exact tier of the scoring kernels): r, da and db as above,
// bit for bit; the two IEEE divisions — 26 of sed_value's 59 fp64 instructions — replaced by ONE reciprocal of da * db (v_rcp_f64:
// 2^-24) refined by one cubic step (y (1 + e + e^2), e = 1 - q y: 2^-72 before rounding), without the scaling / fix-up instructions
// of a full division:  sed' = (da + db) * rcp(da * db) * r^2.  Six roundings of 2^-53 against sed_value's three: the two differ
// by at most 1.2e-15 relative while da * db is far from the ends of the exponent range.  The gate's band is a hundred times that:
//   sed' <= thr - band  an inlier,  sed' > thr + band  an outlier (NaN: every comparison false) — sed_value would say the same;
//   in between, or with a da * db whose reciprocal is not a finite number (zero, deep subnormal, inf, NaN: sed' is then NaN),
//   sed_value's own division sequence decides and is the value.
// So the DECISION is always sed_value's; the sums carry the 1.2e-15 (their summation order already differs from the reference's
// by more: DESIGN.md section 4).  43 fp64-rate instructions per evaluation with the caller's two sums, against 63.
struct SedGate {
    double thr, lo, hi;
};
SFM_DEVICE SedGate sed_gate(double thr) {   // thr NaN: nothing is an inlier; thr = inf: lo is NaN and every finite value takes the division
    const double band = 1e-13 * fabs(thr) + 1e-290;
    return SedGate{thr, thr - band, thr + band};
}
SFM_DEVICE bool sed_inlier(const double e[9], double xa, double ya, double xb, double yb, const SedGate& gate, double& sed_out) {
    const double lb0 = (xb * e[0] + yb * e[3]) + e[6];
    const double lb1 = (xb * e[1] + yb * e[4]) + e[7];
    const double lb2 = (xb * e[2] + yb * e[5]) + e[8];
    const double r = (lb0 * xa + lb1 * ya) + lb2;
    const double la0 = (e[0] * xa + e[1] * ya) + e[2];
    const double la1 = (e[3] * xa + e[4] * ya) + e[5];
    const double da = la0 * la0 + la1 * la1;
    const double db = lb0 * lb0 + lb1 * lb1;
    const double r2 = r * r;
#if SFM_SED_EXACT_DIVISION   // (A/B builds: tools/r04/fastdiv.sh)
    sed_out = (1.0 / da + 1.0 / db) * r2;
    return sed_out <= gate.thr;
#endif
    const double q = da * db;
    double y = __builtin_amdgcn_rcp(q);
    const double err = fma(-q, y, 1.0);
    y = fma(y, fma(err, err, err), y);
    double sed = ((da + db) * y) * r2;
    // No range check of q is needed: when 1 / q is not a finite number (q zero, deep subnormal, inf, NaN) the residual `err` is
    // -inf or NaN, so y and sed are NaN — and a NaN is "not above hi" and "not at most lo": unclear, like a value inside the band.
    // (A q in the last two binades next to either end passes with one or two bits of q or y lost to the subnormal grid: 5e-16,
    // far inside the band.)
    const bool in_lo = sed <= gate.lo, in_hi = !(sed > gate.hi);
    bool in = in_lo;
    if (in_lo != in_hi) {
        asm volatile("" ::: "memory");   // (a real branch: taken by no lane of almost every wave)
        sed = (1.0 / da + 1.0 / db) * r2;
        in = sed <= gate.thr;
    }
    sed_out = sed;
    return in;
}

// --------------------------------------------------------------------------------------------------
// Jacobi rotation parameters that annihilate the off-diagonal g of [[a, g], [g, b]]:
// returns (c, s, t) with t = tan(theta) the smaller root.  g == 0 gives the identity.
// --------------------------------------------------------------------------------------------------
SFM_DEVICE void jacobi_cs(double a, double b, double g, double& c, double& s, double& t) {
    const double zeta = (b - a) / (2.0 * g);
    // |zeta| may be inf (g tiny) -> t = 0; NaN only when g == 0 and a == b, handled by the select.
    const double az = fabs(zeta);
    double tt = 1.0 / (az + sqrt(1.0 + az * az));
    tt = (zeta < 0.0) ? -tt : tt;
    const bool live = (g != 0.0) && (az == az);
    t = live ? tt : 0.0;
    c = 1.0 / sqrt(1.0 + t * t);
    s = t * c;
}

// --------------------------------------------------------------------------------------------------
// Cyclic two-sided Jacobi eigen-decomposition of a symmetric 9x9 matrix held as its upper triangle.
// On return w[9] are the eigenvalues (unsorted) and V[k*9 + j] is component k of eigenvector j.
// Replaces np.linalg.eig at reference eight_point.py:410 (the matrix there is symmetric PSD).
// --------------------------------------------------------------------------------------------------
template <int P, int Q>
struct TriIndex {  // index of (P,Q), P <= Q, in a row-major packed upper triangle of a 9x9
    static constexpr int value = P * 9 - (P * (P - 1)) / 2 + (Q - P);
};

template <int P, int Q>
SFM_DEVICE double& sym(double* a) {
    if constexpr (P <= Q) return a[TriIndex<P, Q>::value];
    else return a[TriIndex<Q, P>::value];
}

template <int P, int Q, int K>
SFM_DEVICE void rotate_offdiag(double* a, double c, double s) {
    if constexpr (K != P && K != Q) {
        const double akp = sym<K, P>(a);
        const double akq = sym<K, Q>(a);
        sym<K, P>(a) = c * akp - s * akq;
        sym<K, Q>(a) = s * akp + c * akq;
    }
}

template <int P, int Q, int... K>
SFM_DEVICE void rotate_all(double* a, double* v, double c, double s, std::integer_sequence<int, K...>) {
    (rotate_offdiag<P, Q, K>(a, c, s), ...);
    ((void)([&] {
         const double vkp = v[K * 9 + P];
         const double vkq = v[K * 9 + Q];
         v[K * 9 + P] = c * vkp - s * vkq;
         v[K * 9 + Q] = s * vkp + c * vkq;
     }()),
     ...);
}

template <int P, int Q>
SFM_DEVICE void jacobi_rotate9(double* a, double* v) {
    const double app = sym<P, P>(a);
    const double aqq = sym<Q, Q>(a);
    const double apq = sym<P, Q>(a);
    double c, s, t;
    jacobi_cs(app, aqq, apq, c, s, t);
    sym<P, P>(a) = app - t * apq;
    sym<Q, Q>(a) = aqq + t * apq;
    sym<P, Q>(a) = 0.0;
    rotate_all<P, Q>(a, v, c, s, std::make_integer_sequence<int, 9>{});
}

template <int P, int Q>
SFM_DEVICE void jacobi_sweep_from(double* a, double* v) {
    jacobi_rotate9<P, Q>(a, v);
    if constexpr (Q + 1 < 9) jacobi_sweep_from<P, Q + 1>(a, v);
    else if constexpr (P + 2 < 9) jacobi_sweep_from<P + 1, P + 2>(a, v);
}

SFM_DEVICE double offdiag_sq(const double* a) {
    // sum of squares of the strict upper triangle of the packed array
    double acc = 0.0;
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
#pragma unroll
        for (int q = p; q < 9; ++q) {
            if (q != p) acc += a[idx] * a[idx];
            ++idx;
        }
    }
    return acc;
}

// a: packed upper triangle (45), destroyed; v: 81 outputs; w: 9 outputs.  Returns sweeps used.
SFM_DEVICE int jacobi_eig9(double* a, double* v, double* w) {
#pragma unroll
    for (int i = 0; i < 81; ++i) v[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i * 9 + i] = 1.0;
    double diag_sq = 0.0;
    {
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            diag_sq += a[idx] * a[idx];
            idx += 9 - p;
        }
    }
    // Converged when the off-diagonal mass is below rounding of the diagonal scale.  Jacobi converges
    // quadratically, so the last sweep typically drives it to exactly zero / denormal.
    const double stop = diag_sq * 1e-36;
    int sweep = 0;
#pragma unroll 1
    for (; sweep < 24; ++sweep) {
        const double off = offdiag_sq(a);
        const bool more = off > stop;
        if (!__any(more)) break;  // wave-uniform exit: all lanes sweep until the slowest is done
        jacobi_sweep_from<0, 1>(a, v);
    }
    {
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            w[p] = a[idx];
            idx += 9 - p;
        }
    }
    return sweep;
}

// ~1 ulp reciprocal and reciprocal square root (hardware seed + two Newton steps); used where the result feeds an
// iteration that is checked for convergence anyway, not where the reference's rounding must be reproduced.
SFM_DEVICE double rcp_newton(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    return fma(fma(-d, y, 1.0), y, y);
}
SFM_DEVICE double rsqrt_newton(double a) {
    double y = __builtin_amdgcn_rsq(a);
    y = y * fma(-0.5 * a * y, y, 1.5);
    return y * fma(-0.5 * a * y, y, 1.5);
}

// Rotation parameters for one-sided Jacobi from the Gram entries alpha = |g_i|^2, beta = |g_j|^2,
// gamma = g_i . g_j, with one square root, one division and one reciprocal square root:
//   t = 2 gamma sign(beta - alpha) / (|beta - alpha| + sqrt((beta - alpha)^2 + 4 gamma^2)),  c = rsqrt(1 + t^2), s = t c.
// (Same t as jacobi_cs; c^2 + s^2 = 1 to a couple of ulps, which only rescales the rotated pair uniformly.)
SFM_DEVICE void jacobi_cs_gram(double alpha, double beta, double gamma, bool rot, double& c, double& s) {
    // The square root, the division and the reciprocal square root go through the hardware seeds + Newton steps
    // (~1 ulp each, a third of the instructions of the IEEE sequences): an error of an ulp in t leaves a residual
    // gamma of the size the rotation's own roundings leave anyway, and the sweeps run until none is needed.
    const double diff = beta - alpha;
    const double q = fma(diff, diff, 4.0 * (gamma * gamma));
    const double den = fabs(diff) + q * rsqrt_newton(q);  // q = 0: NaN, caught below
    double t = (2.0 * gamma) * rcp_newton(den);
    t = (diff < 0.0) ? -t : t;
    t = (rot && t == t) ? t : 0.0;  // no rotation (or 0/0): identity
    c = rsqrt_newton(fma(t, t, 1.0));
    s = t * c;
}

// --------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi SVD of an NxN matrix stored column-wise: g[col][row].
// On return the columns of g are sigma_k * u_k (mutually orthogonal) and v[col][row] holds V.
// Works on A directly (never on A^T A), so small singular directions keep high relative accuracy.
// --------------------------------------------------------------------------------------------------
template <int N, int I, int J>
SFM_DEVICE bool hestenes_rotate(double (&g)[N][N], double (&v)[N][N]) {
    // No reference rounding to reproduce in here (the result is a converged iteration): let multiplies and adds fuse.
    // The translation unit is compiled with -ffp-contract=off for the NumPy-order arithmetic of the SED / Hartley code.
#pragma clang fp contract(fast)
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        alpha += g[I][k] * g[I][k];
        beta += g[J][k] * g[J][k];
        gamma += g[I][k] * g[J][k];
    }
    // relative orthogonality test |gamma| > 1e-15 sqrt(alpha beta), squared (no square root); columns that are exactly
    // zero are left alone.  Rotation parameters with one square root, one division and one reciprocal square root
    // (jacobi_cs_gram) instead of the two divisions and two square roots of jacobi_cs: these rotations are most of the
    // eight-point fit's dependent chain (3 x 3 rank-2 enforcement: ~6 sweeps of 3).
    const bool rot = gamma * gamma > 1e-30 * (alpha * beta);
    double c, s;
    jacobi_cs_gram(alpha, beta, gamma, rot, c, s);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double gi = g[I][k], gj = g[J][k];
        g[I][k] = c * gi - s * gj;
        g[J][k] = s * gi + c * gj;
        const double vi = v[I][k], vj = v[J][k];
        v[I][k] = c * vi - s * vj;
        v[J][k] = s * vi + c * vj;
    }
    return rot;
}

template <int N, int I, int J>
SFM_DEVICE bool hestenes_sweep_from(double (&g)[N][N], double (&v)[N][N]) {
    bool any = hestenes_rotate<N, I, J>(g, v);
    if constexpr (J + 1 < N) any |= hestenes_sweep_from<N, I, J + 1>(g, v);
    else if constexpr (I + 2 < N) any |= hestenes_sweep_from<N, I + 1, I + 2>(g, v);
    return any;
}

template <int N>
SFM_DEVICE void hestenes_svd(double (&g)[N][N], double (&v)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < N; ++k) v[i][k] = (i == k) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        const bool rotated = hestenes_sweep_from<N, 0, 1>(g, v);
        if (!__any(rotated)) break;
    }
}

// One-sided Jacobi without accumulating V: on return the columns of g are orthogonal and their norms are
// the singular values.
template <int N, int I, int J>
SFM_DEVICE bool hestenes_rotate_novec(double (&g)[N][N]) {
#pragma clang fp contract(fast)
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        alpha += g[I][k] * g[I][k];
        beta += g[J][k] * g[J][k];
        gamma += g[I][k] * g[J][k];
    }
    const bool rot = gamma * gamma > 1e-30 * (alpha * beta);
    double c, s;
    jacobi_cs_gram(alpha, beta, gamma, rot, c, s);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double gi = g[I][k], gj = g[J][k];
        g[I][k] = c * gi - s * gj;
        g[J][k] = s * gi + c * gj;
    }
    return rot;
}

template <int N, int I, int J>
SFM_DEVICE bool hestenes_sweep_novec_from(double (&g)[N][N]) {
    bool any = hestenes_rotate_novec<N, I, J>(g);
    if constexpr (J + 1 < N) any |= hestenes_sweep_novec_from<N, I, J + 1>(g);
    else if constexpr (I + 2 < N) any |= hestenes_sweep_novec_from<N, I + 1, I + 2>(g);
    return any;
}

// one-sided Jacobi without V: on return the columns of g are orthogonal (g = G V for some orthogonal V)
template <int N>
SFM_DEVICE void hestenes_orthogonalise(double (&g)[N][N]) {
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        const bool rotated = hestenes_sweep_novec_from<N, 0, 1>(g);
        if (!__any(rotated)) break;
    }
}

// squared singular values of the NxN matrix whose columns are g[col][.] (g is destroyed)
template <int N>
SFM_DEVICE void singular_values_sq(double (&g)[N][N], double (&sq)[N]) {
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        const bool rotated = hestenes_sweep_novec_from<N, 0, 1>(g);
        if (!__any(rotated)) break;
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) acc += g[c][k] * g[c][k];
        sq[c] = acc;
    }
}

// --------------------------------------------------------------------------------------------------
// Householder QR of the 9x8 matrix whose columns are the eight design rows y_i (9-vectors):
//   A = Q [R; 0].  The null vector of Y = A^T is the last column of Q; the singular values of Y are those
// of the 8x8 upper-triangular R.  No pivoting (fixed indices -> registers); backward stable.
// On return col[j][i], i <= j, holds R(i, j); nullvec = Q e_9 (unit norm).
// --------------------------------------------------------------------------------------------------
SFM_DEVICE void qr_null_vector(double (&col)[8][9], double (&rdiag)[8], double (&nullvec)[9]) {
    // dot products and rank-one updates as fused multiply-adds: half the instructions, and one rounding less each
#pragma clang fp contract(fast)
    // Reflection J: H = I - beta v v^T with v = x - alpha e_J kept in col[J][J..8]; alpha = R(J,J) goes to rdiag.
    // alpha takes the sign opposite to x_J, so v_J = x_J - alpha has no cancellation.
    double beta[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        double norm2 = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) norm2 += (i >= J) ? col[J][i] * col[J][i] : 0.0;
        // (IEEE square root and division here: the Newton forms that pay in the Jacobi rotations made this kernel
        // slower at full occupancy — 111 vs 90 us for 256 x 2000 fits, profiles/r02/README.md)
        const double norm = sqrt(norm2);
        const double x0 = col[J][J];
        const double alpha = (x0 > 0.0) ? -norm : norm;
        const double v0 = x0 - alpha;
        const double vtv = norm2 - x0 * x0 + v0 * v0;
        const double b = (vtv > 0.0) ? 2.0 / vtv : 0.0;
        beta[J] = b;
        rdiag[J] = alpha;
        col[J][J] = v0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c > J) {
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) dot += (i >= J) ? col[J][i] * col[c][i] : 0.0;
                const double scale = b * dot;
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i >= J) col[c][i] -= scale * col[J][i];
            }
        }
    }
    // q9 = H_0 H_1 ... H_7 e_9: apply the reflections to e_9 in reverse order
#pragma unroll
    for (int i = 0; i < 9; ++i) nullvec[i] = (i == 8) ? 1.0 : 0.0;
#pragma unroll
    for (int J = 7; J >= 0; --J) {
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) dot += (i >= J) ? col[J][i] * nullvec[i] : 0.0;
        const double scale = beta[J] * dot;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i >= J) nullvec[i] -= scale * col[J][i];
    }
}

// --------------------------------------------------------------------------------------------------
// Right singular vector of the smallest singular value of a 4x4 matrix given by rows.
// Replaces `np.linalg.svd(A)[2][-1]` of reference triangulation.py:34-35 (sign and scale are irrelevant: the
// caller divides by the last component).
// --------------------------------------------------------------------------------------------------
template <int I, int J>
SFM_DEVICE bool rotate_rows4(double (&g)[4][4]) {
#pragma clang fp contract(fast)   // converged iterations / tolerance-checked: multiplies and adds may fuse
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        alpha += g[I][k] * g[I][k];
        beta += g[J][k] * g[J][k];
        gamma += g[I][k] * g[J][k];
    }
    const bool rot = fabs(gamma) > 1e-15 * sqrt(alpha * beta);
    double c, s;
    jacobi_cs_gram(alpha, beta, gamma, rot, c, s);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double gi = g[I][k], gj = g[J][k];
        g[I][k] = c * gi - s * gj;
        g[J][k] = s * gi + c * gj;
    }
    return rot;
}

// Fast path for the DLT null vector: Householder QR of A (no squaring of the condition number), the null-vector
// estimate R^-1 e4, then inverse iteration x <- normalise(R^-1 R^-T x), which converges like (sigma4 / sigma3)^2 per
// step — three steps to 1e-15 for correspondences that satisfy the epipolar constraint (the RANSAC inliers the pose
// stage works on; verified against LAPACK's SVD to 6e-14 on 2000 noisy inliers, pixel and normalised units).
// Up to 16 steps (the loop ends as soon as every lane of the wave has converged; with a cap of 4 a third of the waves
// of the C5 batch still fell through to Jacobi).  Returns false if they did not converge (rays nearly parallel, or
// gross outliers): the caller then uses the Jacobi route for the whole wave.
SFM_DEVICE bool null_vector4_qr(const double rows[4][4], double x[4]) {
#pragma clang fp contract(fast)   // converged iterations / tolerance-checked: multiplies and adds may fuse
    double r[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[i][j] = rows[i][j];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double below = 0.0;  // squared norm of column k strictly below the diagonal
#pragma unroll
        for (int i = k + 1; i < 4; ++i) below += r[i][k] * r[i][k];
        const double norm = sqrt(r[k][k] * r[k][k] + below);
        const double alpha = (r[k][k] > 0.0) ? -norm : norm;
        const double vk = r[k][k] - alpha;
        const double vn = vk * vk + below;
        const double beta = (vn > 0.0) ? 2.0 * rcp_newton(vn) : 0.0;
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
            double w = vk * r[k][j];
#pragma unroll
            for (int i = k + 1; i < 4; ++i) w += r[i][k] * r[i][j];
            w *= beta;
            r[k][j] -= w * vk;
#pragma unroll
            for (int i = k + 1; i < 4; ++i) r[i][j] -= w * r[i][k];
        }
        r[k][k] = alpha;
    }
    // guard the diagonal: an exactly singular A (noise-free rays) leaves r44 = 0, its null vector is still R^-1 e4
    const double scale = fmax(fmax(fabs(r[0][0]), fabs(r[1][1])), fmax(fabs(r[2][2]), fabs(r[3][3])));
    const double tiny = scale * 1e-16 + 1e-300;
    double inv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double d = (fabs(r[i][i]) < tiny) ? ((r[i][i] < 0.0) ? -tiny : tiny) : r[i][i];
        inv[i] = rcp_newton(d);
    }
    x[3] = 1.0;
    x[2] = -(r[2][3] * x[3]) * inv[2];
    x[1] = -(r[1][2] * x[2] + r[1][3] * x[3]) * inv[1];
    x[0] = -((r[0][1] * x[1] + r[0][2] * x[2]) + r[0][3] * x[3]) * inv[0];
    {
        const double s = rsqrt_newton((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] *= s;
    }
    bool converged = false;
#pragma unroll 1
    for (int step = 0; step < 16; ++step) {
        double y[4], z[4];
        y[0] = x[0] * inv[0];
        y[1] = (x[1] - r[0][1] * y[0]) * inv[1];
        y[2] = (x[2] - (r[0][2] * y[0] + r[1][2] * y[1])) * inv[2];
        y[3] = (x[3] - ((r[0][3] * y[0] + r[1][3] * y[1]) + r[2][3] * y[2])) * inv[3];
        z[3] = y[3] * inv[3];
        z[2] = (y[2] - r[2][3] * z[3]) * inv[2];
        z[1] = (y[1] - (r[1][2] * z[2] + r[1][3] * z[3])) * inv[1];
        z[0] = (y[0] - ((r[0][1] * z[1] + r[0][2] * z[2]) + r[0][3] * z[3])) * inv[0];
        const double n2 = (z[0] * z[0] + z[1] * z[1]) + (z[2] * z[2] + z[3] * z[3]);
        const double dot = (z[0] * x[0] + z[1] * x[1]) + (z[2] * x[2] + z[3] * x[3]);
        const double s = ((dot < 0.0) ? -1.0 : 1.0) * rsqrt_newton(n2);
        double change = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double next = z[i] * s;
            change = fmax(change, fabs(next - x[i]));
            x[i] = next;
        }
        converged = change < 1e-15;  // NaN / inf anywhere: stays false
        if (__all(converged)) break;  // wave-uniform
    }
    return converged;
}

SFM_DEVICE void null_vector4_jacobi(const double rows[4][4], double x[4]) {
#pragma clang fp contract(fast)   // converged iterations / tolerance-checked: multiplies and adds may fuse
    // One-sided Jacobi on the ROWS of A.  Orthogonalising rows is a left multiplication by an orthogonal matrix
    // (A = U S V^T  =>  U^T A = S V^T), so at convergence the rows are sigma_k v_k^T: nothing has to be
    // accumulated.  The wanted vector v_4 is then formed as the 4-D cross product of the three dominant rows,
    // which stays exact when sigma_4 is 0 and the fourth row itself has vanished.
    double g[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) g[r][c] = rows[r][c];
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = rotate_rows4<0, 1>(g);
        rotated |= rotate_rows4<0, 2>(g);
        rotated |= rotate_rows4<0, 3>(g);
        rotated |= rotate_rows4<1, 2>(g);
        rotated |= rotate_rows4<1, 3>(g);
        rotated |= rotate_rows4<2, 3>(g);
        if (!__any(rotated)) break;
    }
    double n2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc += g[r][c] * g[r][c];
        n2[r] = acc;
    }
    int m = 0;
    double best = n2[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const bool smaller = n2[r] < best;
        best = smaller ? n2[r] : best;
        m = smaller ? r : m;
    }
    // the other three rows (a, b, c), selected without dynamic indexing
    double a[4], b[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = (m == 0) ? g[1][k] : g[0][k];
        b[k] = (m <= 1) ? g[2][k] : g[1][k];
        c[k] = (m <= 2) ? g[3][k] : g[2][k];
    }
    // 4-D cross product: x_i = (-1)^i det of the 3x3 minor without column i
    auto det3 = [](double a0, double a1, double a2, double b0, double b1, double b2, double c0, double c1, double c2) {
        return a0 * (b1 * c2 - b2 * c1) - a1 * (b0 * c2 - b2 * c0) + a2 * (b0 * c1 - b1 * c0);
    };
    x[0] = det3(a[1], a[2], a[3], b[1], b[2], b[3], c[1], c[2], c[3]);
    x[1] = -det3(a[0], a[2], a[3], b[0], b[2], b[3], c[0], c[2], c[3]);
    x[2] = det3(a[0], a[1], a[3], b[0], b[1], b[3], c[0], c[1], c[3]);
    x[3] = -det3(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2]);
}

// Null vector (smallest right singular vector) of the 4x4 DLT matrix: the QR / inverse-iteration route, with the Jacobi
// route for waves in which some lane did not converge.
SFM_DEVICE void null_vector4(const double rows[4][4], double x[4]) {
    const bool ok = null_vector4_qr(rows, x);
    if (!__all(ok)) {  // wave-uniform
        double xj[4];
        null_vector4_jacobi(rows, xj);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = ok ? x[i] : xj[i];
    }
}

// DLT triangulation of one pair (reference triangulation.py:9-39).  P1, P2: rows 0..2 of the camera
// matrices, 4 columns each (row-major 12 doubles).  X = null(A)[:3] / null(A)[3], unguarded.
SFM_DEVICE void triangulate_dlt(const double* P1, const double* P2, double xa, double ya, double xb,
                                double yb, double X[3]) {
    double A[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = ya * P1[8 + k] - P1[4 + k];
        A[1][k] = P1[k] - xa * P1[8 + k];
        A[2][k] = yb * P2[8 + k] - P2[4 + k];
        A[3][k] = P2[k] - xb * P2[8 + k];
    }
    double x[4];
    null_vector4(A, x);
    X[0] = x[0] / x[3];
    X[1] = x[1] / x[3];
    X[2] = x[2] / x[3];
}

// --------------------------------------------------------------------------------------------------
// Philox-4x32-10 (Salmon et al., SC'11), bit-identical to oracle/sfm_oracle.py::philox4x32_10.
// --------------------------------------------------------------------------------------------------
SFM_DEVICE void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// First 8 positions of a Fisher-Yates shuffle of range(n) for hypothesis h (sparse bookkeeping of the
// at most 8 displaced positions).  Draw k: j = k + ((u_k * (n-k)) >> 32).
SFM_DEVICE void philox_sample8(uint64_t seed, uint64_t h, uint32_t n, int32_t out[8]) {
    uint32_t u[8];
    {
        uint32_t c[4] = {(uint32_t)h, (uint32_t)(h >> 32), 0u, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[0] = c[0]; u[1] = c[1]; u[2] = c[2]; u[3] = c[3];
        uint32_t d[4] = {(uint32_t)h, (uint32_t)(h >> 32), 1u, 0u};
        philox4x32_10(d, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[4] = d[0]; u[5] = d[1]; u[6] = d[2]; u[7] = d[3];
    }
    uint32_t dpos[8], dval[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t j = (uint32_t)k + (uint32_t)(((uint64_t)u[k] * (uint64_t)(n - (uint32_t)k)) >> 32);
        uint32_t vj = j, vk = (uint32_t)k;
#pragma unroll
        for (int m = 0; m < k; ++m) {
            vj = (dpos[m] == j) ? dval[m] : vj;
            vk = (dpos[m] == (uint32_t)k) ? dval[m] : vk;
        }
        out[k] = (int32_t)vj;
        dpos[k] = j;
        dval[k] = vk;
    }
}

// 64-lane sums with a fixed combination order (deterministic, no atomics): four DPP row rotations give every lane
// the sum of its 16-lane row, the four row sums are then read out (v_readlane) and added in row order.  All VALU — a
// __shfl_xor butterfly goes through the LDS crossbar (ds_bpermute) six times in a dependent chain, which is what a
// short kernel epilogue waits on.
template <int N>
SFM_DEVICE int dpp_row_ror(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0x120 | N, 0xf, 0xf, false);
}
template <int N>
SFM_DEVICE double dpp_row_ror(double x) {
    return __hiloint2double(dpp_row_ror<N>(__double2hiint(x)), dpp_row_ror<N>(__double2loint(x)));
}
SFM_DEVICE double read_lane(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
}
SFM_DEVICE double wave_sum(double x) {
    x += dpp_row_ror<8>(x);
    x += dpp_row_ror<4>(x);
    x += dpp_row_ror<2>(x);
    x += dpp_row_ror<1>(x);
    return ((read_lane(x, 0) + read_lane(x, 16)) + read_lane(x, 32)) + read_lane(x, 48);
}
SFM_DEVICE int wave_sum(int x) {
    x += dpp_row_ror<8>(x);
    x += dpp_row_ror<4>(x);
    x += dpp_row_ror<2>(x);
    x += dpp_row_ror<1>(x);
    return ((__builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16)) + __builtin_amdgcn_readlane(x, 32)) +
           __builtin_amdgcn_readlane(x, 48);
}

}  // namespace sfm
