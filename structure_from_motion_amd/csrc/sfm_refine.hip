// Local optimisation of a RANSAC winner: refit the essential matrix on ALL its inliers with the N-point form of the
// reference's normalised eight-point algorithm, re-score every correspondence, keep the refit if it explains
// more points.  SURVEY.md §8(f) rank 4 — an extension the reference does not have (it returns the 8-point model
// of the winning sample as is); off unless asked for.  The fit itself is the reference's own pipeline run on M >= 8
// pairs: _normalize_coords (eight_point.py:308-338), _get_yT_y (:363-393), _compute_f_est (:396-427),
// _enforce_fundamental_mat_constraints (:430-446), T2^T F T1 and the division by [2,2] (:163-166).
//
// One 512-thread block per image pair does the whole loop: three strided reduction passes over the inliers
// (centroids; mean distances; the 45 distinct entries of Y^T Y), the smallest eigenvector of Y^T Y by inverse iteration
// on a Cholesky factor + rank-2 projection on wave 0,
// one scoring pass over all N points, and — if accepted — one pass rewriting the mask.  N <= 50k points x 32 B is
// L2-resident; the kernel is latency-bound and runs once per RANSAC call, off the headline path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_fit.h"
#include "sfm_math.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;

static_assert(sizeof(sfm_refine_info) == 16, "sfm_refine_info layout is part of the ABI");

// Block-wide sums of K doubles per thread in a fixed order: butterfly inside each wave, then the wave
// partials added in wave order by one thread per value.  Result broadcast through `total`.
template <int K>
SFM_DEVICE void block_sum(double (&v)[K], double (*part)[48], double* total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = sfm::wave_sum(v[k]);
        if (lane == 0) part[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double acc = part[0][threadIdx.x];
        for (int w = 1; w < kWaves; ++w) acc += part[w][threadIdx.x];
        total[threadIdx.x] = acc;
    }
    __syncthreads();
}

SFM_DEVICE double aggregate_total(int aggregation, double count, double sum1, double sum2) {
    switch (aggregation) {
        case SFM_AGG_SUM: return sum1;
        case SFM_AGG_SQUARE: return sum2;
        case SFM_AGG_MEAN: return sum1 / count;
        default: return sqrt(sum2 / count);
    }
}

__global__ __launch_bounds__(kThreads) void refine_kernel(
    const Corr* __restrict__ corr, int n, const double* __restrict__ E_in, const uint8_t* __restrict__ mask_in,
    const double* __restrict__ err_in, double thr, int aggregation, int iterations, double* __restrict__ E_out,
    uint8_t* __restrict__ mask_out, sfm_refine_info* __restrict__ info) {
    __shared__ double part[kWaves][48];
    __shared__ double total[48];
    __shared__ double e_new[9];
    __shared__ int flag_new;

    const int64_t b = blockIdx.x;
    const Corr* pts = corr + b * n;
    const uint8_t* min = mask_in + b * n;
    uint8_t* mout = mask_out + b * n;
    const int tid = threadIdx.x;

    // start: the input model and its inliers (sample points, marked 2 by sfm_inlier_mask, count as inliers)
    double v1[1] = {0.0};
    for (int i = tid; i < n; i += kThreads) {
        const uint8_t m = min[i] != 0 ? 1 : 0;
        mout[i] = m;
        v1[0] += (double)m;
    }
    block_sum<1>(v1, part, total);
    double best_cnt = total[0];
    double best_err = err_in[b];
    if (tid < 9) E_out[b * 9 + tid] = E_in[b * 9 + tid];
    int accepted = 0;

    for (int it = 0; it < iterations; ++it) {
        if (!(best_cnt >= 8.0)) break;  // block-uniform
        // pass A: centroids of the inliers in both images
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = tid; i < n; i += kThreads) {
            if (mout[i]) {
                const Corr p = pts[i];
                s[0] += p.xa; s[1] += p.ya; s[2] += p.xb; s[3] += p.yb;
            }
        }
        block_sum<4>(s, part, total);
        sfmfit::Hartley t1, t2;
        t1.cx = total[0] / best_cnt; t1.cy = total[1] / best_cnt;
        t2.cx = total[2] / best_cnt; t2.cy = total[3] / best_cnt;
        // (`total` is next written behind the first barrier of the next block_sum: no barrier needed here)
        // pass B: mean distance from the centroid -> scale sqrt(2) / mean (eight_point.py:323-324)
        double d[2] = {0.0, 0.0};
        for (int i = tid; i < n; i += kThreads) {
            if (mout[i]) {
                const Corr p = pts[i];
                const double ax = p.xa - t1.cx, ay = p.ya - t1.cy, bx = p.xb - t2.cx, by = p.yb - t2.cy;
                d[0] += sqrt(ax * ax + ay * ay);
                d[1] += sqrt(bx * bx + by * by);
            }
        }
        block_sum<2>(d, part, total);
        t1.scale = sqrt(2.0) / (total[0] / best_cnt);
        t2.scale = sqrt(2.0) / (total[1] / best_cnt);
        // pass C: upper triangle of Y^T Y over the normalised inliers (eight_point.py:363-393)
        double a[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) a[k] = 0.0;
        for (int i = tid; i < n; i += kThreads) {
            if (mout[i]) {
                const Corr p = pts[i];
                const double xa = (p.xa - t1.cx) * t1.scale, ya = (p.ya - t1.cy) * t1.scale;
                const double xb = (p.xb - t2.cx) * t2.scale, yb = (p.yb - t2.cy) * t2.scale;
                const double col[9] = {xb * xa, xb * ya, xb, yb * xa, yb * ya, yb, xa, ya, 1.0};
                int idx = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r)
#pragma unroll
                    for (int q = r; q < 9; ++q) a[idx++] += col[r] * col[q];
            }
        }
        block_sum<45>(a, part, total);
        // eigen-solve + rank 2 + un-normalise: every lane of wave 0 runs the same problem (the Jacobi sweeps are
        // wave-uniform loops); lane 0 publishes
        if (tid < kWave) {
            double yty[45], f[9];
#pragma unroll
            for (int k = 0; k < 45; ++k) yty[k] = total[k];
            const int flag = sfmfit::eigenvalues_below9(yty, 1e-10) >= 2 ? SFM_FIT_DEGENERATE : 0;
            sfmfit::smallest_eigenvector_psd9(yty, f);
            double fr[3][3], e[9];
            sfmfit::enforce_rank2(f, fr);
            sfmfit::unnormalise(fr, t1, t2, e);
            const double e22 = e[8];
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) e_new[k] = e[k] / e22;
                flag_new = flag;
            }
        }
        __syncthreads();
        if (flag_new != 0) break;  // degenerate inlier configuration (eight_point.py:415-421): keep what we have
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = e_new[k];
        // pass D: score every correspondence under the refit model (sed.py:7-30; `<=` as ransac.py:74)
        double c[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < n; i += kThreads) {
            const Corr p = pts[i];
            const double sed = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
            if (sed <= thr) {
                c[0] += 1.0;
                c[1] += sed;
                c[2] += sed * sed;
            }
        }
        block_sum<3>(c, part, total);
        const double cnt = total[0];
        const double err = aggregate_total(aggregation, cnt, total[1], total[2]);
        const bool better = cnt > best_cnt || (cnt == best_cnt && err < best_err);  // NaN error never wins
        if (!better) break;
        for (int i = tid; i < n; i += kThreads) {
            const Corr p = pts[i];
            mout[i] = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb) <= thr ? 1 : 0;
        }
        if (tid < 9) E_out[b * 9 + tid] = e[tid];
        best_cnt = cnt;
        best_err = err;
        ++accepted;
    }
    if (tid == 0) {
        info[b].error = best_err;
        info[b].count = (int32_t)best_cnt;
        info[b].accepted = accepted;
    }
}

}  // namespace

extern "C" {

int sfm_refine_inliers(const double* corr, int64_t n, int64_t batch, const double* E_in, const uint8_t* mask_in,
                       const double* err_in, double thr, int aggregation, int iterations, double* E_out,
                       uint8_t* mask_out, sfm_refine_info* info, void* stream) {
    if (n < 0 || batch < 0 || iterations < 0) return fail(SFM_EINVAL, "sfm_refine_inliers: negative size");
    if (n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_refine_inliers: n too large");
    if (aggregation < SFM_AGG_SUM || aggregation > SFM_AGG_RMS)
        return fail(SFM_EINVAL, "sfm_refine_inliers: unknown aggregation");
    if (batch == 0) return SFM_OK;
    if (!E_in || !err_in || !E_out || !info || (n > 0 && (!corr || !mask_in || !mask_out)))
        return fail(SFM_EINVAL, "sfm_refine_inliers: null pointer");
    if (mask_in == mask_out) return fail(SFM_EINVAL, "sfm_refine_inliers: mask_out must not alias mask_in");
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)batch), dim3(kThreads), 0, (hipStream_t)stream,
                       (const Corr*)corr, (int)n, E_in, mask_in, err_in, thr, aggregation, iterations, E_out,
                       mask_out, info);
    return check_launch("refine_kernel");
}

}  // extern "C"
