// Brute-force window matching on the GPU (reference lib/feature_matching/matching.py:36-81 with
// ncc.py:7-54 / ssd.py:7-36 as the score function and util.py:8-27 for the windows).
//
//   patch_extract_kernel   one lane per feature: bounds test, window gather, (NCC) mean removal and sum of
//                          squares.  Patches are stored k-major  P[k][feature]  so that the score kernel's
//                          loads are coalesced across features.
//   pair_scores_kernel     |A| x |B| scores, 64x64 tile per 256-thread block, 4x4 outputs per lane, window
//                          chunks staged through LDS.  The window sum runs over k = 0..K-1 in order with
//                          separate multiply / add roundings (no FMA), so every score is bit-identical to the
//                          oracle's sequential evaluation whatever the tiling.  fp64 VALU bound (2K flop per
//                          pair); MFMA is deliberately not used: its fused accumulation would change the bits.
//   row_summary_kernel     one wave per A-feature scans its row in B order and reproduces what the reference's
//                          per-feature heapq holds at positions 0 and 1 after pushing the scores in order:
//                          heap[0] = first minimum; heap[1] (the LEFT child, which the ratio test divides by —
//                          not necessarily the second best) = min over pushes i landing in the left subtree of
//                          max(score_i, running minimum before i).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_math.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;

constexpr int kTile = 64;     // features per tile side
constexpr int kChunk = 32;    // window elements staged per LDS pass

__global__ void patch_extract_kernel(const double* __restrict__ image, int64_t height, int64_t width,
                                     const double* __restrict__ feats, int64_t n, int64_t stride, int half,
                                     int subtract_mean, double* __restrict__ patches,
                                     double* __restrict__ ssq, uint8_t* __restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = feats[2 * i], y = feats[2 * i + 1];
    const int side = 2 * half + 1;
    const int K = side * side;
    // util.py:8-18 (comparisons on the float coordinates), util.py:21-27 (int() truncation)
    const bool inside = ((double)half <= y) && (y < (double)(height - half)) && ((double)half <= x) &&
                        (x < (double)(width - half));
    if (!inside) {
        for (int k = 0; k < K; ++k) patches[(int64_t)k * stride + i] = 0.0;
        ssq[i] = 0.0;
        ok[i] = 0;
        return;
    }
    const int64_t x0 = (int64_t)x - half, y0 = (int64_t)y - half;
    double mean = 0.0;
    if (subtract_mean) {
        double total = 0.0;
        for (int r = 0; r < side; ++r)
            for (int c = 0; c < side; ++c) total += image[(y0 + r) * width + (x0 + c)];
        mean = total / (double)K;
    }
    double sq = 0.0;
    int k = 0;
    for (int r = 0; r < side; ++r)
        for (int c = 0; c < side; ++c) {
            const double v = image[(y0 + r) * width + (x0 + c)] - mean;
            patches[(int64_t)k * stride + i] = v;
            sq += v * v;
            ++k;
        }
    ssq[i] = sq;
    ok[i] = 1;
}

// MODE 0: NCC (patches mean-removed): score = (num / sqrt(qa qb)) * -1 + 1, 2.0 if a window is out of the image
//         or the denominator is zero (ncc.py:24-54).
// MODE 1: SSD: sum (a-b)^2 / K, +inf if a window is out of the image (ssd.py:24-36).
template <int MODE>
__global__ __launch_bounds__(256) void pair_scores_kernel(
    const double* __restrict__ Pa, int64_t stride_a, const double* __restrict__ Pb, int64_t stride_b,
    const double* __restrict__ qa, const double* __restrict__ qb, const uint8_t* __restrict__ oka,
    const uint8_t* __restrict__ okb, int64_t nA, int64_t nB, int K, double* __restrict__ scores) {
    __shared__ double sA[kChunk][kTile];
    __shared__ double sB[kChunk][kTile];
    const int tid = threadIdx.x;
    const int ty = tid / 16, tx = tid % 16;  // 16 x 16 lanes, 4 x 4 outputs each
    const int64_t a0 = (int64_t)blockIdx.y * kTile, b0 = (int64_t)blockIdx.x * kTile;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

    for (int k0 = 0; k0 < K; k0 += kChunk) {
        const int kc = min(kChunk, K - k0);
        __syncthreads();
        for (int idx = tid; idx < kc * kTile; idx += 256) {
            const int kk = idx / kTile, f = idx % kTile;
            const int64_t ia = a0 + f, ib = b0 + f;
            sA[kk][f] = ia < nA ? Pa[(int64_t)(k0 + kk) * stride_a + ia] : 0.0;
            sB[kk][f] = ib < nB ? Pb[(int64_t)(k0 + kk) * stride_b + ib] : 0.0;
        }
        __syncthreads();
        for (int kk = 0; kk < kc; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = sA[kk][ty * 4 + i];
                bv[i] = sB[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 0) {
                        acc[i][j] += av[i] * bv[j];
                    } else {
                        const double d = av[i] - bv[j];
                        acc[i][j] += d * d;
                    }
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t ia = a0 + ty * 4 + i;
        if (ia >= nA) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ib = b0 + tx * 4 + j;
            if (ib >= nB) continue;
            const bool inside = oka[ia] && okb[ib];
            double s;
            if (MODE == 0) {
                const double den = sqrt(qa[ia] * qb[ib]);
                s = (!inside || den == 0.0) ? 2.0 : (acc[i][j] / den) * -1.0 + 1.0;
            } else {
                s = inside ? acc[i][j] / (double)K : INFINITY;
            }
            scores[ia * nB + ib] = s;
        }
    }
}

SFM_DEVICE bool in_left_subtree(int64_t position_1based) {
    if (position_1based < 2) return false;
    const int k = 63 - __builtin_clzll((unsigned long long)position_1based);
    return position_1based < ((int64_t)1 << k) + ((int64_t)1 << (k - 1));
}

// (value, index) lexicographic minimum == first occurrence of the smallest value
struct MinAt {
    double v;
    int64_t i;
};
SFM_DEVICE MinAt min_at(MinAt a, MinAt b) {
    const bool take_b = (b.v < a.v) || (b.v == a.v && b.i < a.i);
    return take_b ? b : a;
}

__global__ __launch_bounds__(256) void row_summary_kernel(const double* __restrict__ scores, int64_t nA,
                                                          int64_t nB, double* __restrict__ best,
                                                          int32_t* __restrict__ arg,
                                                          double* __restrict__ second) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (row >= nA) return;
    const double* s = scores + row * nB;
    double carry = INFINITY;   // running minimum of everything before this chunk
    MinAt top = {INFINITY, INT64_MAX};
    double sec = INFINITY;
    for (int64_t base = 0; base < nB; base += kWave) {
        const int64_t i = base + lane;
        const bool valid = i < nB;
        const double v = valid ? s[i] : INFINITY;
        // exclusive prefix minimum within the chunk, seeded with the carry
        double incl = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const double other = __shfl_up(incl, off, kWave);
            if (lane >= off) incl = fmin(incl, other);
        }
        double before = __shfl_up(incl, 1, kWave);
        before = (lane == 0) ? carry : fmin(carry, before);
        if (valid && i >= 1 && in_left_subtree(i + 1)) {
            const double cand = (v < before) ? before : v;  // displaced root, or the new item itself
            sec = fmin(sec, cand);
        }
        if (valid) top = min_at(top, MinAt{v, i});
        carry = fmin(carry, __shfl(incl, kWave - 1, kWave));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        MinAt o;
        o.v = __shfl_xor(top.v, off, kWave);
        o.i = __shfl_xor(top.i, off, kWave);
        top = min_at(top, o);
        sec = fmin(sec, __shfl_xor(sec, off, kWave));
    }
    if (lane == 0) {
        best[row] = top.v;
        arg[row] = (int32_t)top.i;
        second[row] = nB > 1 ? sec : NAN;
    }
}

}  // namespace

extern "C" {

int sfm_patch_extract(const double* image, int64_t height, int64_t width, const double* feats, int64_t n,
                      int window_size, int subtract_mean, int64_t stride, double* patches, double* ssq,
                      uint8_t* ok, void* stream) {
    if (n < 0 || height <= 0 || width <= 0 || window_size < 1)
        return fail(SFM_EINVAL, "sfm_patch_extract: bad size");
    if (n == 0) return SFM_OK;
    if (stride < n) return fail(SFM_EINVAL, "sfm_patch_extract: stride < n");
    if (!image || !feats || !patches || !ssq || !ok) return fail(SFM_EINVAL, "sfm_patch_extract: null pointer");
    hipLaunchKernelGGL(patch_extract_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, image,
                       height, width, feats, n, stride, window_size / 2, subtract_mean, patches, ssq, ok);
    return check_launch("patch_extract_kernel");
}

int sfm_pair_scores(int metric, const double* patches_a, int64_t stride_a, const double* patches_b,
                    int64_t stride_b, const double* ssq_a, const double* ssq_b, const uint8_t* ok_a,
                    const uint8_t* ok_b, int64_t n_a, int64_t n_b, int window_elements, double* scores,
                    void* stream) {
    if (n_a < 0 || n_b < 0 || window_elements < 1) return fail(SFM_EINVAL, "sfm_pair_scores: bad size");
    if (metric != SFM_MATCH_NCC && metric != SFM_MATCH_SSD) return fail(SFM_EINVAL, "sfm_pair_scores: unknown metric");
    if (n_a == 0 || n_b == 0) return SFM_OK;
    if (!patches_a || !patches_b || !ssq_a || !ssq_b || !ok_a || !ok_b || !scores)
        return fail(SFM_EINVAL, "sfm_pair_scores: null pointer");
    const dim3 grid(grid_for(n_b, kTile), grid_for(n_a, kTile));
    if (metric == SFM_MATCH_NCC)
        hipLaunchKernelGGL(pair_scores_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, patches_a, stride_a,
                           patches_b, stride_b, ssq_a, ssq_b, ok_a, ok_b, n_a, n_b, window_elements, scores);
    else
        hipLaunchKernelGGL(pair_scores_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, patches_a, stride_a,
                           patches_b, stride_b, ssq_a, ssq_b, ok_a, ok_b, n_a, n_b, window_elements, scores);
    return check_launch("pair_scores_kernel");
}

int sfm_match_row_summary(const double* scores, int64_t n_a, int64_t n_b, double* best, int32_t* arg,
                          double* second, void* stream) {
    if (n_a < 0 || n_b < 0) return fail(SFM_EINVAL, "sfm_match_row_summary: negative size");
    if (n_a == 0) return SFM_OK;
    if (n_b == 0) return fail(SFM_EINVAL, "sfm_match_row_summary: empty rows");
    if (n_b > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_match_row_summary: rows too long");
    if (!scores || !best || !arg || !second) return fail(SFM_EINVAL, "sfm_match_row_summary: null pointer");
    hipLaunchKernelGGL(row_summary_kernel, dim3(grid_for(n_a, 256 / kWave)), dim3(256), 0, (hipStream_t)stream,
                       scores, n_a, n_b, best, arg, second);
    return check_launch("row_summary_kernel");
}

}  // extern "C"
