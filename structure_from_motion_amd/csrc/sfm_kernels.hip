// HIP kernels (gfx950 / MI355X) + C ABI for the RANSAC essential-matrix, cheirality and
// triangulation hot path.  See include/sfm_hip.h for the contract of each entry point and the
// reference lines it replaces; DESIGN.md for the data layout and the roofline of each kernel.
//
// Layout in HBM
//   corr  [batch][n][4]  f64  rows {xa, ya, xb, yb}: one 32-byte record per correspondence, read by a
//                             lane as two 16-byte loads; a wave reads 2 KiB contiguous per step.
//   S     [batch][H][8]  i32  sample table (indices into corr)
//   E     [batch][H][9]  f64  candidate essential matrices, row-major, E[8] == 1
//   cnt/s1/s2/flags [batch][H]
//
// Compiled with -ffp-contract=off (see sfm_math.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "sfm_common.h"
#include "sfm_math.h"
#include "sfm_fit.h"
#include "sfm_score_ws.h"
#include "sfm_matrix_tables.h"
#include "sfm_select.h"

namespace sfmhost {
char* error_buffer() {
    static thread_local char buffer[kErrorBytes] = "";
    return buffer;
}
}  // namespace sfmhost

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;
using sfmhost::grid_stride;
using namespace sfmfit;
using sfmsel::hypothesis_key;
using sfmsel::kNoModelKey;

static_assert(sizeof(sfm_select_result) == 40, "sfm_select_result layout is part of the ABI");

// ------------------------------------------------------------------------------------------------
// K-normalisation pre-pass: one thread per correspondence, 2x16 B in, 32 B out.
// ------------------------------------------------------------------------------------------------
__global__ void normalize_kernel(const double2* __restrict__ pix_a, const double2* __restrict__ pix_b,
                                 int64_t count, double fx, double fy, double cx, double cy,
                                 Corr* __restrict__ corr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const double2 a = pix_a[i];
        const double2 b = pix_b[i];
        Corr c;
        c.xa = (a.x - cx) / fx;
        c.ya = (a.y - cy) / fy;
        c.xb = (b.x - cx) / fx;
        c.yb = (b.y - cy) / fy;
        corr[i] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// Philox sampler: one thread per hypothesis, 32 B out.
// ------------------------------------------------------------------------------------------------
__global__ void sample_philox_kernel(uint64_t seed, uint64_t seed_stride, int64_t h_begin,
                                     int64_t h_count, uint32_t n, int32_t* __restrict__ S) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_count) return;
    const int64_t b = blockIdx.y;
    int32_t idx[8];
    sfm::philox_sample8(seed + (uint64_t)b * seed_stride, (uint64_t)(h_begin + h), n, idx);
    int4* dst = reinterpret_cast<int4*>(S + (b * h_count + h) * 8);
    dst[0] = make_int4(idx[0], idx[1], idx[2], idx[3]);
    dst[1] = make_int4(idx[4], idx[5], idx[6], idx[7]);
}

// The same sampler with the seed read from device memory, so a captured hipGraph of a whole RANSAC pass
// can be replayed with a new seed (kernel arguments are frozen at capture; device memory is not).
__global__ void sample_philox_dev_kernel(const uint64_t* __restrict__ seed_dev, uint64_t seed_stride,
                                         int64_t h_begin, int64_t h_count, uint32_t n,
                                         int32_t* __restrict__ S) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= h_count) return;
    const int64_t b = blockIdx.y;
    int32_t idx[8];
    sfm::philox_sample8(seed_dev[0] + (uint64_t)b * seed_stride, (uint64_t)(h_begin + h), n, idx);
    int4* dst = reinterpret_cast<int4*>(S + (b * h_count + h) * 8);
    dst[0] = make_int4(idx[0], idx[1], idx[2], idx[3]);
    dst[1] = make_int4(idx[4], idx[5], idx[6], idx[7]);
}

// Re-derive the sample of one given hypothesis per batch entry (index read from device memory, so the
// multi-GPU winner can be finalised without a host round trip).  Negative index -> 0..7.
__global__ void sample_philox_at_kernel(uint64_t seed, uint64_t seed_stride,
                                        const int64_t* __restrict__ h_index, uint32_t n, int64_t batch,
                                        int32_t* __restrict__ S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int64_t h = h_index[b];
    int32_t idx[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    if (h >= 0) sfm::philox_sample8(seed + (uint64_t)b * seed_stride, (uint64_t)h, n, idx);
    int4* dst = reinterpret_cast<int4*>(S + b * 8);
    dst[0] = make_int4(idx[0], idx[1], idx[2], idx[3]);
    dst[1] = make_int4(idx[4], idx[5], idx[6], idx[7]);
}

// ------------------------------------------------------------------------------------------------
// Eight-point fit: one lane per hypothesis (all 64 lanes of a wave busy), everything in registers.
// ------------------------------------------------------------------------------------------------
// Trace record written by the traced fit (doubles): normalised coords a [8][2] | b [8][2] | T1 {scale,cx,cy} |
// T2 {scale,cx,cy} | Y^T Y full [9][9] | eigenvalues [9] | f_est [9] | rank-2 F [9]
constexpr int kTraceDoubles = 16 + 16 + 3 + 3 + 81 + 9 + 9 + 9;

// Where the eight sample indices of a hypothesis come from: the table S (enabled == 0), or the Philox sampler run
// inside the fit kernel, which then also fills S (one launch instead of two; a small RANSAC pass is a chain of
// ~4 us kernels).  `seed_dev`, if not NULL, supplies the seed from device memory (graph replay).
struct PhiloxSource {
    const uint64_t* seed_dev;
    uint64_t seed, seed_stride;
    int64_t h_begin;
    int enabled;
};

// Fused small pass (sfm_ransac_pass_small): `blocks` extra blocks behind the fit blocks of the launch prepare the
// scoring workspace of the launch that follows (fp32 points, partial maxima, arrival ticket) — the preparation depends
// on the correspondences only, so it rides along instead of costing a launch of its own.  blocks == 0: plain fit.
struct SmallPrep {
    unsigned char* ws;
    double a_scale;
    int blocks;
    int32_t* order;   // scoring order of the pass that follows (see fit_eight_point_kernel), or NULL
};

// Fused LARGE / batched pass (sfm_ransac_pass_large, sfm_ransac_pass_batch) where the matrix-pipe scoring kernel follows: the fit
// lanes — which hold their hypothesis' E and sample in registers — also write the hypothesis' rows of that kernel's operand table
// and its sample correction (sfm_matrix_tables.h: what matrix_tables_kernel's hypothesis half does from E and S re-read), and
// `step_blocks` extra blocks behind the fit blocks write the point operand table, four steps each (single pair; the chip is
// mostly idle under the fit's one wave of 256 registers per 64 hypotheses).  Both need the data-set maxima of the points:
// every wave folds the partial maxima matrix_setup_kernel left in the launch before.  partial == NULL: plain fit.
struct MatrixPrep {
    const float4* partial;     // [batch][partials] partial coordinate maxima
    int partials;
    double a_scale, thr;
    uint4* hyp_table;          // [batch][h_count][2][3] x 16 bytes
    unsigned char* fix;        // [batch] x sfmws::matrix_fix_bytes(h_count)
    uint4* table;              // point operand table of the pair (step_blocks > 0)
    int step_blocks;
};

template <bool TRACE, bool MATRIX = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void fit_eight_point_kernel(
    const Corr* __restrict__ corr, int64_t n, int32_t* __restrict__ S, int64_t h_count,
    double* __restrict__ E, int32_t* __restrict__ flags, double* __restrict__ lambda2,
    double* __restrict__ trace, PhiloxSource philox, SmallPrep prep, MatrixPrep matrix) {
    const int64_t b = blockIdx.y;
    if (prep.blocks > 0 && blockIdx.x >= gridDim.x - prep.blocks) {  // block-uniform; batch == 1 in this mode
        sfmws::prepare_small_block(corr, (int)n, prep.a_scale, prep.ws, (int)(blockIdx.x - (gridDim.x - prep.blocks)));
        return;
    }
    uint32_t maxima[4] = {0u, 0u, 0u, 0u};
    if constexpr (MATRIX) {   // (an instantiation of its own: the plain fit keeps its register allocation)
        matrixscore::fold_partial_maxima(matrix.partial + b * matrix.partials, matrix.partials, (int)threadIdx.x, maxima);
        if (blockIdx.x >= gridDim.x - matrix.step_blocks) {  // block-uniform: a block of the point operand table (batch == 1)
            const int first = (int)(blockIdx.x - (gridDim.x - matrix.step_blocks)) * 4;
            const int steps = (int)matrixscore::table_steps(n);
            for (int t = first; t < first + 4 && t < steps; ++t)
                matrixscore::prepare_step(corr, (int)n, matrix.a_scale, maxima, matrix.table, t, (int)threadIdx.x);
            return;
        }
    }
    const int64_t h_raw = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool active = h_raw < h_count;   // (a launch with table blocks has h_count covered by the blocks in front of them)
    // inactive tail lanes redo the last hypothesis so the wave-uniform Jacobi loops stay convergent
    const int64_t h = active ? h_raw : h_count - 1;
    const Corr* pts = corr + b * n;
    int32_t sample[8];
    if (philox.enabled) {  // wave-uniform
        const uint64_t seed = (philox.seed_dev != nullptr ? philox.seed_dev[0] : philox.seed) + (uint64_t)b * philox.seed_stride;
        sfm::philox_sample8(seed, (uint64_t)(philox.h_begin + h), (uint32_t)n, sample);
        if (active) {
            int4* dst = reinterpret_cast<int4*>(S + (b * h_count + h) * 8);
            dst[0] = make_int4(sample[0], sample[1], sample[2], sample[3]);
            dst[1] = make_int4(sample[4], sample[5], sample[6], sample[7]);
        }
    } else {
        const int4* src = reinterpret_cast<const int4*>(S + (b * h_count + h) * 8);
        const int4 lo = src[0], hi = src[1];
        sample[0] = lo.x; sample[1] = lo.y; sample[2] = lo.z; sample[3] = lo.w;
        sample[4] = hi.x; sample[5] = hi.y; sample[6] = hi.z; sample[7] = hi.w;
    }

    double xa[8], ya[8], xb[8], yb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const Corr c = pts[sample[k]];
        xa[k] = c.xa; ya[k] = c.ya; xb[k] = c.xb; yb[k] = c.yb;
    }
    const Hartley t1 = hartley8(xa, ya);
    const Hartley t2 = hartley8(xb, yb);

    double* tr = nullptr;
    if constexpr (TRACE) {
        double a[45];
        build_yty(xa, ya, xb, yb, a);
        tr = trace + (b * h_count + h) * kTraceDoubles;
        if (active) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                tr[2 * k] = xa[k]; tr[2 * k + 1] = ya[k];
                tr[16 + 2 * k] = xb[k]; tr[16 + 2 * k + 1] = yb[k];
            }
            tr[32] = t1.scale; tr[33] = t1.cx; tr[34] = t1.cy;
            tr[35] = t2.scale; tr[36] = t2.cx; tr[37] = t2.cy;
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 9; ++p)
#pragma unroll
                for (int q = p; q < 9; ++q) {
                    tr[38 + p * 9 + q] = a[idx];
                    tr[38 + q * 9 + p] = a[idx];
                    ++idx;
                }
        }
    }

    double f[9], sq[8], second;
    const bool need_second = TRACE || (lambda2 != nullptr);
    const int flag = null_vector_of_design(xa, ya, xb, yb, need_second, f, second, sq);
    double fr[3][3];
    double ratio2 = 0.0;
    enforce_rank2(f, fr, &ratio2);
    if (prep.order != nullptr) {  // wave-uniform (small pass)
        // Scoring order for the launch that follows, for free: a sample of eight inliers gives an estimate that is
        // almost rank 2 already, so (sigma_3 / sigma_1)^2 of the unconstrained F predicts which hypotheses will fit the
        // scene — the ones whose scoring takes several times the average (their exact tier handles half the points).
        // On the bench scene the lowest quarter of this ratio holds every such hypothesis.  The scoring launch walks
        // the order from its front, so the expensive hypotheses start with the launch instead of at its tail.
        // Results are written at each hypothesis' own index: the order changes timing only.
        // non-negative floats order like unsigned ints; their top 16 bits (exponent + 7 mantissa bits: 1 % resolution)
        // are plenty for a scheduling hint
        unsigned key = __float_as_uint((float)ratio2) >> 16;
        if (!(ratio2 == ratio2) || !active) key = 0xFFFFu;   // NaN and the padding lanes of the last wave go last
        // rank of this lane's key within the wave (ties by lane), bit by bit from the top: `same` = lanes whose key agrees
        // with mine on the bits seen so far; where my bit is 1, the lanes of `same` with a 0 are smaller than me
        unsigned long long same = ~0ull;
        int rank = 0;
#pragma unroll
        for (int bit = 15; bit >= 0; --bit) {
            const unsigned long long ones = __builtin_amdgcn_ballot_w64(((key >> bit) & 1u) != 0u);
            const bool mine = ((key >> bit) & 1u) != 0u;
            rank += mine ? (int)__popcll(same & ~ones) : 0;
            same &= mine ? ones : ~ones;
        }
        rank += (int)__popcll(same & ((1ull << threadIdx.x) - 1ull));   // equal keys: lower lanes first
        // rank-major layout: first every wave's lowest ratio, then every wave's second lowest, ... — an approximate
        // global sort by quantile with closed-form slots and no atomics; only the last wave can be partial, so layer r
        // holds one entry per wave for r < active_last and one fewer beyond
        const int64_t waves = (h_count + kWave - 1) / kWave;
        const int64_t w = blockIdx.x;
        const int64_t active_last = h_count - (waves - 1) * kWave;
        if (active) {
            const int64_t layer = (int64_t)rank * waves - (rank > active_last ? rank - active_last : 0);
            prep.order[layer + w] = (int32_t)h_raw;
        }
    }
    if constexpr (TRACE) {
        if (active) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                tr[119 + k] = (k < 8) ? sq[k] : 0.0;  // eigenvalues of Y^T Y: sigma_k(Y)^2 and the null one
                tr[128 + k] = f[k];
                tr[137 + k] = fr[k / 3][k % 3];
            }
        }
    }

    // E = T2^T F T1 (eight_point.py:163), then divide by E[2][2] (:166, unguarded)
    double e[9];
    unnormalise(fr, t1, t2, e);
    const double e22 = e[8];
    double en[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) en[k] = e[k] / e22;
    if (active) {
        double* out = E + (b * h_count + h) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = en[k];
        flags[b * h_count + h] = flag;
        if (lambda2 != nullptr) lambda2[b * h_count + h] = second;
    }
    if constexpr (MATRIX) {
        // this hypothesis' operand rows + sample correction, from the E just stored
        matrixscore::prepare_hypothesis_lane(maxima, en, sample, h, (int)h_count, matrix.a_scale, matrix.thr, pts,
                                                 matrix.hyp_table + b * h_count * 2 * matrixscore::kBlocks,
                                             matrix.fix + b * sfmws::matrix_fix_bytes(h_count), active);
    }
}

// Single-problem stages behind the reference's private helpers (one lane does the work; the other lanes
// of the wave run the same problem so the wave-uniform Jacobi loops behave):
//   stage 0: Y^T Y of 8 coordinate pairs taken as they are        in: corr8 [8][4]      out: [81]
//   stage 1: _compute_f_est (eight_point.py:396-427)             in: yty [81]           out: f_est [9] | w [9] | flag
//   stage 2: _enforce_fundamental_mat_constraints (:430-446)     in: f [9]              out: [9]
__global__ __launch_bounds__(kWave) void fit_stage_kernel(int stage, const double* __restrict__ in,
                                                          double* __restrict__ out) {
    if (stage == 0) {
        double xa[8], ya[8], xb[8], yb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xa[k] = in[4 * k]; ya[k] = in[4 * k + 1]; xb[k] = in[4 * k + 2]; yb[k] = in[4 * k + 3];
        }
        double a[45];
        build_yty(xa, ya, xb, yb, a);
        if (threadIdx.x == 0) {
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 9; ++p)
#pragma unroll
                for (int q = p; q < 9; ++q) {
                    out[p * 9 + q] = a[idx];
                    out[q * 9 + p] = a[idx];
                    ++idx;
                }
        }
    } else if (stage == 1) {
        double a[45];
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int q = p; q < 9; ++q) a[idx++] = in[p * 9 + q];
        double f[9], w[9], second;
        const int flag = null_vector_of_yty(a, f, w, second);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                out[k] = f[k];
                out[9 + k] = w[k];
            }
            out[18] = (double)flag;
        }
    } else {
        // a caller's F can have any magnitude: scaled exactly into the unit range and back (pow2_unit_scale)
        const double scale = pow2_unit_scale(in);
        double f[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = in[k] * scale;
        double fr[3][3];
        enforce_rank2(f, fr);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) out[k] = fr[k / 3][k % 3] / scale;
        }
    }
}

// Hartley normalisation of n points (eight_point.py:308-338) for the `_normalize_coords` helper:
// centroid = sequential sum / n, scale = sqrt(2) / mean distance; out: normalised [n][2] then {scale,cx,cy}.
__global__ void hartley_normalize_kernel(const double* __restrict__ coords, int64_t n, double* __restrict__ out) {
    __shared__ double sh[3];
    if (threadIdx.x == 0) {
        double sx = 0.0, sy = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            sx += coords[2 * i];
            sy += coords[2 * i + 1];
        }
        const double cx = sx / (double)n, cy = sy / (double)n;
        double total = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double dx = coords[2 * i] - cx, dy = coords[2 * i + 1] - cy;
            total += sqrt(dx * dx + dy * dy);
        }
        sh[0] = sqrt(2.0) / (total / (double)n);
        sh[1] = cx;
        sh[2] = cy;
        out[2 * n] = sh[0];
        out[2 * n + 1] = cx;
        out[2 * n + 2] = cy;
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        out[2 * i] = (coords[2 * i] - sh[1]) * sh[0];
        out[2 * i + 1] = (coords[2 * i + 1] - sh[2]) * sh[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Selection: one 1024-thread block per batch entry; lexicographic min over (error bits, index).
// ------------------------------------------------------------------------------------------------
// Three passes over the result record with 64-bit atomics, so that many blocks can share the scan:
//   pass 1  key = min over gated hypotheses of the error bits; flag statistics
//   pass 2  best_h = min index among hypotheses whose key equals the minimum (earliest wins)
//   final   fill error / count, apply h_offset, translate "none" sentinels
__global__ void select_init_kernel(sfm_select_result* __restrict__ result) {
    sfm_select_result r;
    r.key = kNoModelKey;
    r.best_h = INT64_MAX;
    r.best_err = INFINITY;
    r.first_flagged = INT64_MAX;
    r.n_flagged = 0;
    r.best_cnt = 0;
    result[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void select_pass1_kernel(
    const int32_t* __restrict__ cnt, const double* __restrict__ s1, const double* __restrict__ s2,
    const int32_t* __restrict__ flags, int64_t h_count, double min_extra, int aggregation,
    sfm_select_result* __restrict__ result) {
    const int64_t b = blockIdx.y;
    cnt += b * h_count; s1 += b * h_count; s2 += b * h_count;
    if (flags != nullptr) flags += b * h_count;
    uint64_t key = kNoModelKey;
    int64_t first_flag = INT64_MAX;
    int n_flag = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < h_count; h += stride) {
        bool flagged;
        const uint64_t k = hypothesis_key(cnt, s1, s2, flags, h, min_extra, aggregation, flagged);
        key = k < key ? k : key;
        if (flagged) {
            first_flag = h < first_flag ? h : first_flag;
            ++n_flag;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t ok = __shfl_xor(key, off, 64);
        key = ok < key ? ok : key;
        const int64_t of = __shfl_xor(first_flag, off, 64);
        first_flag = of < first_flag ? of : first_flag;
        n_flag += __shfl_xor(n_flag, off, 64);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (key != kNoModelKey) atomicMin((unsigned long long*)&result[b].key, (unsigned long long)key);
        if (n_flag) {
            atomicMin((long long*)&result[b].first_flagged, (long long)first_flag);
            atomicAdd(&result[b].n_flagged, n_flag);
        }
    }
}

__global__ __launch_bounds__(256) void select_pass2_kernel(
    const int32_t* __restrict__ cnt, const double* __restrict__ s1, const double* __restrict__ s2,
    const int32_t* __restrict__ flags, int64_t h_count, double min_extra, int aggregation,
    sfm_select_result* __restrict__ result) {
    const int64_t b = blockIdx.y;
    const uint64_t target = result[b].key;
    if (target == kNoModelKey) return;
    cnt += b * h_count; s1 += b * h_count; s2 += b * h_count;
    if (flags != nullptr) flags += b * h_count;
    int64_t best = INT64_MAX;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < h_count; h += stride) {
        bool flagged;
        const uint64_t k = hypothesis_key(cnt, s1, s2, flags, h, min_extra, aggregation, flagged);
        if (k == target && h < best) best = h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int64_t o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && best != INT64_MAX)
        atomicMin((long long*)&result[b].best_h, (long long)best);
}

__global__ void select_final_kernel(const int32_t* __restrict__ cnt, int64_t h_count, int64_t h_offset,
                                    sfm_select_result* __restrict__ result) {
    const int64_t b = blockIdx.x;
    sfm_select_result r = result[b];
    const bool found = r.key != kNoModelKey && r.best_h != INT64_MAX;
    r.best_cnt = found ? cnt[b * h_count + r.best_h] : 0;
    r.best_err = found ? __longlong_as_double((long long)r.key) : INFINITY;
    r.best_h = found ? r.best_h + h_offset : -1;
    if (!found) r.key = kNoModelKey;
    if (r.first_flagged != INT64_MAX) r.first_flagged += h_offset;
    result[b] = r;
}

// The whole selection in ONE launch for moderate hypothesis counts: a 1024-thread block per batch entry takes the
// lexicographic minimum of (error bits, index) directly — same keys, same winner as the three passes above, three
// launches fewer (a small RANSAC pass is a chain of ~4 us kernels; profiles/r01/README.md).
constexpr int kSelectBlock = 1024;
__global__ __launch_bounds__(kSelectBlock) void select_block_kernel(
    const int32_t* __restrict__ cnt, const double* __restrict__ s1, const double* __restrict__ s2,
    const int32_t* __restrict__ flags, int64_t h_count, int64_t h_offset, double min_extra, int aggregation,
    sfm_select_result* __restrict__ result) {
    __shared__ sfmsel::SelectScratch<kSelectBlock> scratch;
    __shared__ int64_t winner;
    const int64_t b = blockIdx.x;
    sfmsel::block_select<kSelectBlock>(cnt + b * h_count, s1 + b * h_count, s2 + b * h_count,
                                       flags != nullptr ? flags + b * h_count : nullptr, h_count, h_offset, min_extra,
                                       aggregation, result + b, scratch, &winner);
}

// Smallest passes (the reference's own workload: a few hundred matches, 2000 iterations): one 1024-thread block selects
// and then writes the winner's mask itself — no cross-block hand-off at all.
__global__ __launch_bounds__(kSelectBlock) void select_block_mask_kernel(
    const int32_t* __restrict__ cnt, const double* __restrict__ s1, const double* __restrict__ s2,
    const int32_t* __restrict__ flags, int64_t h_count, int64_t h_offset, double min_extra, int aggregation,
    sfm_select_result* __restrict__ result, const Corr* __restrict__ corr, int64_t n, const double* __restrict__ E,
    const int32_t* __restrict__ S, double thr, uint8_t* __restrict__ mask) {
    __shared__ sfmsel::SelectScratch<kSelectBlock> scratch;
    __shared__ int64_t winner;
    const int64_t best = sfmsel::block_select<kSelectBlock>(cnt, s1, s2, flags, h_count, h_offset, min_extra, aggregation,
                                                            result, scratch, &winner);
    if (mask != nullptr) sfmsel::write_inlier_mask<4>(corr, n, E, S, h_count, best, thr, mask, threadIdx.x, kSelectBlock);
}

// Selection of a small pass (h_count <= 32768) spread over up to 32 blocks of ONE launch: a single block walking
// every hypothesis is latency-bound (14 us at 10 000, 35 us at 30 000 hypotheses).  Each block folds its slice and
// publishes a partial record write-through (sc1); an agent-scope arrival counter (cdna_hip_programming.md Guideline 16,
// counter form; 32 arrivals, so a single counter does not serialise anything) tells the block that finishes last to
// take one acquire and fold the partial records into the result record.  Counter and records live in the kPointsPad
// bytes behind the fp32 points of the scoring workspace (only ever READ by the scoring loop's over-prefetch); the fit
// launch of the same pass zeroes the counter (sfm_score_ws.h).
struct PartialSelect {
    uint64_t key;
    int64_t best, first_flagged;
    int32_t n_flagged, pad;
};
static_assert(sfmws::kFusedPartialOffset + sfmws::kFusedShards * (int)sizeof(PartialSelect) <= sfmws::kPointsPad,
              "selection state fits the pad");

// With a mask requested the same launch carries ceil(n / 256) further blocks behind the selecting ones: each loads its
// 256 points, waits (one lane, relaxed agent-scope polls with s_sleep, bounded) for the "record published" flag the last
// selecting block raises, and writes its slice of the winner's inlier mask — the mask costs no launch of its own.  All
// blocks of the launch (at most 64) are resident together on any MI355X, so the wait cannot deadlock; should the flag
// not arrive within the bound the slice is filled with 0xFF, a value no mask holds, rather than hanging the GPU.
constexpr unsigned kSelectDoneWord = 8;   // unsigned index into `state`: counter at [0], flag at [8] (same 64-byte line)
constexpr int kMaxFlagPolls = 1 << 22;

__global__ __launch_bounds__(256) void select_sharded_kernel(
    const int32_t* __restrict__ cnt, const double* __restrict__ s1, const double* __restrict__ s2,
    const int32_t* __restrict__ flags, int64_t h_count, int64_t h_offset, double min_extra, int aggregation,
    unsigned char* __restrict__ state, sfm_select_result* __restrict__ result, int select_blocks,
    const Corr* __restrict__ corr, int64_t n, const double* __restrict__ E, const int32_t* __restrict__ S, double thr,
    uint8_t* __restrict__ mask) {
    __shared__ sfmsel::SelectScratch<256> scratch;
    __shared__ int last_block;
    unsigned* counter = reinterpret_cast<unsigned*>(state);
    unsigned* done = counter + kSelectDoneWord;
    PartialSelect* partial = reinterpret_cast<PartialSelect*>(state + sfmws::kFusedPartialOffset);
    if ((int)blockIdx.x >= select_blocks) {
        // ---- mask block: points [256 * m, 256 * m + 256) ----
        const int64_t i = (int64_t)(blockIdx.x - select_blocks) * 256 + threadIdx.x;
        const Corr p = corr[i < n ? i : n - 1];   // in flight while waiting
        if (threadIdx.x == 0) {
            int polls = 0;
            // acquire loads at agent scope pair with the release store of the flag below: once the flag reads 1 the
            // record written before it is visible to this block
            while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u && polls < kMaxFlagPolls) {
                __builtin_amdgcn_s_sleep(8);
                ++polls;
            }
            last_block = polls < kMaxFlagPolls ? 1 : 0;   // reused as "record is there"
        }
        __syncthreads();
        if (i >= n) return;
        if (!last_block) {
            mask[i] = 0xFF;
            return;
        }
        // the record carries the GLOBAL index (local winner + h_offset); E and S are this launch's local arrays
        const int64_t global_h = __hip_atomic_load(&result->best_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t h = global_h - h_offset;
        if (global_h < 0 || h < 0 || h >= h_count) {
            mask[i] = 0;
            return;
        }
        double e[9];
        bool in_sample = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = E[h * 9 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) in_sample |= (S[h * 8 + k] == (int32_t)i);
        const double sed = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
        mask[i] = in_sample ? 2 : ((sed <= thr) ? 1 : 0);
        return;
    }
    uint64_t key = kNoModelKey;
    int64_t best = INT64_MAX, first_flag = INT64_MAX;
    int n_flag = 0;
    // this block's slice: hypotheses blockIdx.x * 256 + t, + select_blocks * 256, ... — all loads of a thread in flight together
    const int64_t stride = (int64_t)select_blocks * 256;
    uint64_t k[4];
    bool flagged[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x + u * stride;
        flagged[u] = false;
        k[u] = h < h_count ? hypothesis_key(cnt, s1, s2, flags, h, min_extra, aggregation, flagged[u]) : kNoModelKey;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x + u * stride;
        if (k[u] < key) {  // increasing h: strict < keeps the earliest
            key = k[u];
            best = h;
        }
        if (flagged[u]) {
            first_flag = h < first_flag ? h : first_flag;
            ++n_flag;
        }
    }
    sfmsel::block_combine<256>(key, best, first_flag, n_flag, scratch);
    if (threadIdx.x == 0) {
        PartialSelect* out = partial + blockIdx.x;
        __hip_atomic_store(&out->key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->first_flagged, first_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->n_flagged, (int32_t)n_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // arrival: release (this block's partial record is visible before the count) + acquire (the block that
        // arrives last sees every other block's record)
        const unsigned arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_block = arrived == (unsigned)select_blocks - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last_block) return;
    key = kNoModelKey;
    best = INT64_MAX;
    first_flag = INT64_MAX;
    n_flag = 0;
    if ((int)threadIdx.x < select_blocks) {
        const PartialSelect p = partial[threadIdx.x];
        key = p.key;
        best = p.best;
        first_flag = p.first_flagged;
        n_flag = p.n_flagged;
    }
    sfmsel::block_combine<256>(key, best, first_flag, n_flag, scratch);
    if (threadIdx.x == 0) {
        const bool found = key != kNoModelKey && best != INT64_MAX;
        // the record is read by the waiting mask blocks of this launch: write-through stores, drained, then the flag
        __hip_atomic_store(&result->key, found ? key : kNoModelKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->best_h, found ? best + h_offset : (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->best_err, found ? __longlong_as_double((long long)key) : (double)INFINITY,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->first_flagged, first_flag != INT64_MAX ? first_flag + h_offset : INT64_MAX,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->n_flagged, (int32_t)n_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->best_cnt, found ? cnt[best] : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // publishes the record above
    }
}

// Selection launch of a fused LARGE pass (sfm_ransac_pass_large): select_sharded_kernel for any number of hypotheses — up to 256
// selecting blocks walk them grid-stride — with the fold of a range-split scoring launch in front: a thread first adds the
// partials of its hypotheses' ranges in range order and the sample correction (what matrix_fold_kernel does as a launch of its
// own), writes cnt / s1 / s2, and selects from those totals.  State (arrival counter, flag, partial records) in the unused head of
// the range-split region of the scoring workspace, zeroed by the pass's first scoring launch.
constexpr int kLargeSelectBlocks = 256;
__global__ __launch_bounds__(256) void select_large_kernel(
    int32_t* __restrict__ cnt, double* __restrict__ s1, double* __restrict__ s2, const int32_t* __restrict__ flags,
    int64_t h_count, int64_t h_offset, double min_extra, int aggregation, unsigned char* __restrict__ state,
    sfm_select_result* __restrict__ result, int select_blocks, int units, const unsigned char* __restrict__ split,
    const unsigned char* __restrict__ fix, const Corr* __restrict__ corr, int64_t n, const double* __restrict__ E,
    const int32_t* __restrict__ S, double thr, uint8_t* __restrict__ mask) {
    __shared__ sfmsel::SelectScratch<256> scratch;
    __shared__ int last_block;
    unsigned* counter = reinterpret_cast<unsigned*>(state);
    unsigned* done = counter + kSelectDoneWord;
    PartialSelect* partial = reinterpret_cast<PartialSelect*>(state + sfmws::kFusedPartialOffset);
    if ((int)blockIdx.x >= select_blocks) {
        // ---- mask block: points [256 * m, 256 * m + 256), as in select_sharded_kernel ----
        const int64_t i = (int64_t)(blockIdx.x - select_blocks) * 256 + threadIdx.x;
        const Corr p = corr[i < n ? i : n - 1];   // in flight while waiting
        if (threadIdx.x == 0) {
            // relaxed polls (each one a load that bypasses the caches, nothing else) and ONE acquire fence once the flag is up:
            // with ~200 blocks polling, an acquire per poll invalidates the XCD's L2 under the selecting blocks' loads
            int polls = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && polls < kMaxFlagPolls) {
                __builtin_amdgcn_s_sleep(16);
                ++polls;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            last_block = polls < kMaxFlagPolls ? 1 : 0;   // reused as "record is there"
        }
        __syncthreads();
        if (i >= n) return;
        if (!last_block) {
            mask[i] = 0xFF;
            return;
        }
        const int64_t global_h = __hip_atomic_load(&result->best_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t h = global_h - h_offset;
        if (global_h < 0 || h < 0 || h >= h_count) {
            mask[i] = 0;
            return;
        }
        double e[9];
        bool in_sample = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = E[h * 9 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) in_sample |= (S[h * 8 + k] == (int32_t)i);
        const double sed = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
        mask[i] = in_sample ? 2 : ((sed <= thr) ? 1 : 0);
        return;
    }
    uint64_t key = kNoModelKey;
    int64_t best = INT64_MAX, first_flag = INT64_MAX;
    int n_flag = 0;
    const int64_t hp = sfmws::split_padded(h_count);
    const int32_t* part_c = units > 1 ? reinterpret_cast<const int32_t*>(split) + hp : nullptr;
    const double* part_a1 = units > 1 ? reinterpret_cast<const double*>(part_c + (int64_t)units * hp) : nullptr;
    const double* part_a2 = units > 1 ? part_a1 + (int64_t)units * hp : nullptr;
    const int32_t* fix_c = reinterpret_cast<const int32_t*>(fix);
    const double* fix_a1 = units > 1 ? reinterpret_cast<const double*>(fix + 4 * hp) : nullptr;
    for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < h_count; h += (int64_t)select_blocks * 256) {
        if (units > 1) {   // the ranges in range order, then the sample correction: matrix_fold_kernel's sums, bit for bit
            // (eight ranges' partials in flight together, then added in range order: one memory latency per eight ranges, not
            // one per range — the launch spent 16 of its 25 us in eight dependent round trips)
            int total = 0;
            double t1 = 0.0, t2 = 0.0;
            for (int u0 = 0; u0 < units; u0 += 8) {
                int c8[8];
                double a8[8], b8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int64_t at = (int64_t)(u0 + k < units ? u0 + k : u0) * hp + h;
                    c8[k] = part_c[at];
                    a8[k] = part_a1[at];
                    b8[k] = part_a2[at];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (u0 + k == 0) {   // (starts FROM range 0's values, as matrix_fold_kernel does: 0.0 + x is x, but -0.0 would not survive)
                        total = c8[k];
                        t1 = a8[k];
                        t2 = b8[k];
                    } else if (u0 + k < units) {
                        total += c8[k];
                        t1 += a8[k];
                        t2 += b8[k];
                    }
                }
            }
            cnt[h] = total + fix_c[h];
            s1[h] = t1 + fix_a1[h];
            s2[h] = t2 + fix_a1[hp + h];
        }
        bool flagged = false;
        const uint64_t k = hypothesis_key(cnt, s1, s2, flags, h, min_extra, aggregation, flagged);
        if (k < key) {  // increasing h: strict < keeps the earliest
            key = k;
            best = h;
        }
        if (flagged) {
            first_flag = h < first_flag ? h : first_flag;
            ++n_flag;
        }
    }
    sfmsel::block_combine<256>(key, best, first_flag, n_flag, scratch);
    if (threadIdx.x == 0) {
        PartialSelect* out = partial + blockIdx.x;
        __hip_atomic_store(&out->key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->first_flagged, first_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out->n_flagged, (int32_t)n_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // arrival: release (this block's totals and partial record are visible before the count); the block that arrives last
        // takes the acquire — one fence, not one per arrival
        const unsigned arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        last_block = arrived == (unsigned)select_blocks - 1 ? 1 : 0;
        if (last_block) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last_block) return;
    key = kNoModelKey;
    best = INT64_MAX;
    first_flag = INT64_MAX;
    n_flag = 0;
    if ((int)threadIdx.x < select_blocks) {
        const PartialSelect p = partial[threadIdx.x];
        key = p.key;
        best = p.best;
        first_flag = p.first_flagged;
        n_flag = p.n_flagged;
    }
    sfmsel::block_combine<256>(key, best, first_flag, n_flag, scratch);
    if (threadIdx.x == 0) {
        const bool found = key != kNoModelKey && best != INT64_MAX;
        __hip_atomic_store(&result->key, found ? key : kNoModelKey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->best_h, found ? best + h_offset : (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->best_err, found ? __longlong_as_double((long long)key) : (double)INFINITY,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->first_flagged, first_flag != INT64_MAX ? first_flag + h_offset : INT64_MAX,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&result->n_flagged, (int32_t)n_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the winner's count was written by another block of this launch: an agent-scope load, behind the acquire above)
        __hip_atomic_store(&result->best_cnt,
                           found ? __hip_atomic_load(cnt + best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // publishes the record above
    }
}

// Last launch of a fused BATCHED pass (sfm_ransac_pass_batch): one 1024-thread block per image pair folds the ranges of its
// pair's scoring launch (matrix_fold_kernel's sums, bit for bit: range order, then the sample correction), selects
// (ransac.py:75-86) and writes the winner's inlier mask — three launches of the separate calls in one, and no cross-block
// hand-off: a pair's hypotheses (a few thousand) and points (~10 000) are one block's work.
__global__ __launch_bounds__(kSelectBlock) void select_fold_mask_batch_kernel(
    int32_t* __restrict__ cnt, double* __restrict__ s1, double* __restrict__ s2, const int32_t* __restrict__ flags, int64_t h_count,
    double min_extra, int aggregation, sfm_select_result* __restrict__ result, int units, const unsigned char* __restrict__ split,
    const unsigned char* __restrict__ fix, const Corr* __restrict__ corr, int64_t n, const double* __restrict__ E,
    const int32_t* __restrict__ S, double thr, uint8_t* __restrict__ mask) {
    __shared__ sfmsel::SelectScratch<kSelectBlock> scratch;
    __shared__ int64_t winner;
    const int64_t b = blockIdx.x;
    cnt += b * h_count;
    s1 += b * h_count;
    s2 += b * h_count;
    if (units > 1) {
        const int64_t hp = sfmws::split_padded(h_count);
        const unsigned char* sp = split + b * sfmws::split_bytes(h_count, units);
        const unsigned char* fx = fix + b * sfmws::matrix_fix_bytes(h_count);
        const int32_t* part_c = reinterpret_cast<const int32_t*>(sp) + hp;
        const double* part_a1 = reinterpret_cast<const double*>(part_c + (int64_t)units * hp);
        const double* part_a2 = part_a1 + (int64_t)units * hp;
        const int32_t* fix_c = reinterpret_cast<const int32_t*>(fx);
        const double* fix_a1 = reinterpret_cast<const double*>(fx + 4 * hp);
        // (thread t folds hypotheses t, t + 1024, ... — the ones block_select has it read back below)
        for (int64_t h = threadIdx.x; h < h_count; h += kSelectBlock) {
            int total = part_c[h];
            double t1 = part_a1[h], t2 = part_a2[h];
            for (int u = 1; u < units; ++u) {
                total += part_c[u * hp + h];
                t1 += part_a1[u * hp + h];
                t2 += part_a2[u * hp + h];
            }
            cnt[h] = total + fix_c[h];
            s1[h] = t1 + fix_a1[h];
            s2[h] = t2 + fix_a1[hp + h];
        }
        __syncthreads();
    }
    const int64_t best = sfmsel::block_select<kSelectBlock>(cnt, s1, s2, flags != nullptr ? flags + b * h_count : nullptr, h_count, 0,
                                                            min_extra, aggregation, result + b, scratch, &winner);
    if (mask != nullptr)
        sfmsel::write_inlier_mask<4>(corr + b * n, n, E + b * h_count * 9, S + b * h_count * 8, h_count, best, thr, mask + b * n,
                                     threadIdx.x, kSelectBlock);
}

// ------------------------------------------------------------------------------------------------
// Inlier mask of the winner (ransac.py:70-76): 1 = surviving non-sample point, 2 = sample point.
// ------------------------------------------------------------------------------------------------
__global__ void inlier_mask_kernel(const Corr* __restrict__ corr, int64_t n,
                                   const double* __restrict__ E, const int32_t* __restrict__ S,
                                   int64_t h_count, const sfm_select_result* __restrict__ result,
                                   double thr, uint8_t* __restrict__ mask) {
    const int64_t b = blockIdx.y;
    sfmsel::write_inlier_mask(corr + b * n, n, E + b * h_count * 9, S + b * h_count * 8, h_count, result[b].best_h, thr,
                              mask + b * n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                              (int64_t)gridDim.x * blockDim.x);
}

__global__ void sed_values_kernel(const Corr* __restrict__ corr, int64_t n, const double* __restrict__ E,
                                  double* __restrict__ out) {
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Corr p = corr[i];
        out[i] = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
    }
}

// ------------------------------------------------------------------------------------------------
// Cheirality (eight_point.py:449-488): one lane per (pose, pair).  P1 = I4, P2 = [R t; 0 0 0 1].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void cheirality_kernel(const Corr* __restrict__ corr, int64_t m,
                                                           const double* __restrict__ pose_rt,
                                                           double distance_threshold,
                                                           uint8_t* __restrict__ pass) {
    const int64_t i_raw = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool active = i_raw < m;
    const int64_t i = active ? i_raw : m - 1;
    const int pose = blockIdx.y;
    const double* rt = pose_rt + pose * 12;
    double P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double P2[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        P2[r * 4 + 0] = rt[r * 3 + 0];
        P2[r * 4 + 1] = rt[r * 3 + 1];
        P2[r * 4 + 2] = rt[r * 3 + 2];
        P2[r * 4 + 3] = rt[9 + r];
    }
    const Corr p = corr[i];
    double X[3];
    sfm::triangulate_dlt(P1, P2, p.xa, p.ya, p.xb, p.yb, X);
    // depth in camera 2: third row of P2 @ [X, 1] (eight_point.py:476), left to right
    const double z2 = ((P2[8] * X[0] + P2[9] * X[1]) + P2[10] * X[2]) + P2[11];
    const double norm = sqrt((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
    const bool ok = (X[2] >= -1e-8) && (z2 >= -1e-8) && (norm <= distance_threshold);
    if (active) pass[(int64_t)pose * m + i] = ok ? 1 : 0;
}

__global__ __launch_bounds__(kWave) void triangulate_kernel(const Corr* __restrict__ corr, int64_t m,
                                                            const double* __restrict__ P1g,
                                                            const double* __restrict__ P2g,
                                                            double* __restrict__ Xout) {
    const int64_t i_raw = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool active = i_raw < m;
    const int64_t i = active ? i_raw : m - 1;
    double P1[12], P2[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        P1[k] = P1g[k];
        P2[k] = P2g[k];
    }
    const Corr p = corr[i];
    double X[3];
    sfm::triangulate_dlt(P1, P2, p.xa, p.ya, p.xb, p.yb, X);
    if (active) {
        Xout[i * 3 + 0] = X[0];
        Xout[i * 3 + 1] = X[1];
        Xout[i * 3 + 2] = X[2];
    }
}

// ------------------------------------------------------------------------------------------------
// Essential-matrix decomposition (eight_point.py:245-280): E = U S V^T, t = vee(U Z U^T) = u3,
// R1 = U W^T V^T, R2 = U W V^T with U, V made proper rotations.  One lane per matrix.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void decompose_essential_kernel(const double* __restrict__ E,
                                                                    int64_t batch,
                                                                    double* __restrict__ pose_rt,
                                                                    int32_t* __restrict__ status) {
    const int64_t b_raw = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool active = b_raw < batch;
    const int64_t b = active ? b_raw : batch - 1;
    const double* e = E + b * 9;
    // E can have any magnitude (the reference's SVD does not care): scaled exactly into the unit range for the Jacobi
    // sweeps, singular values scaled back for the sigma_3 ~ 0 test, which is absolute (pow2_unit_scale)
    const double scale = sfmfit::pow2_unit_scale(e);
    double g[3][3], v[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) g[c][r] = e[r * 3 + c] * scale;
    sfm::hestenes_svd<3>(g, v);
    double sig[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) sig[c] = sqrt(g[c][0] * g[c][0] + g[c][1] * g[c][1] + g[c][2] * g[c][2]);  // of the scaled E
    // order columns by decreasing singular value: (i0, i1, i2)
    int i0 = 0, i1 = 1, i2 = 2;
    if (sig[i0] < sig[i1]) { int t = i0; i0 = i1; i1 = t; }
    if (sig[i1] < sig[i2]) { int t = i1; i1 = i2; i2 = t; }
    if (sig[i0] < sig[i1]) { int t = i0; i0 = i1; i1 = t; }
    double u[3][3], vt[3][3];  // u[col][row], vt[col][row] = V columns
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double g0 = 0, g1 = 0, v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g0 = (c == i0) ? g[c][r] : g0;
            g1 = (c == i1) ? g[c][r] : g1;
            v0 = (c == i0) ? v[c][r] : v0;
            v1 = (c == i1) ? v[c][r] : v1;
            v2 = (c == i2) ? v[c][r] : v2;
        }
        u[0][r] = g0;
        u[1][r] = g1;
        vt[0][r] = v0;
        vt[1][r] = v1;
        vt[2][r] = v2;
    }
    double s0 = 0, s1v = 0, s2v = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s0 = (c == i0) ? sig[c] : s0;
        s1v = (c == i1) ? sig[c] : s1v;
        s2v = (c == i2) ? sig[c] : s2v;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u[0][r] /= s0;
        u[1][r] /= s1v;
    }
    // third left singular vector from the cross product: U is a proper rotation by construction,
    // which is what the det(U) == -1 -> U *= -1 branch (eight_point.py:263-264) establishes.
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    // make V proper as well: flip v3 if det(V) < 0 (the reference flips all of V^T; R1/R2 below are
    // then the same pair of rotations)
    const double cx = vt[0][1] * vt[1][2] - vt[0][2] * vt[1][1];
    const double cy = vt[0][2] * vt[1][0] - vt[0][0] * vt[1][2];
    const double cz = vt[0][0] * vt[1][1] - vt[0][1] * vt[1][0];
    const double detv = cx * vt[2][0] + cy * vt[2][1] + cz * vt[2][2];
    if (detv < 0.0) {
        vt[2][0] = -vt[2][0];
        vt[2][1] = -vt[2][1];
        vt[2][2] = -vt[2][2];
    }
    // np.isclose(0, s[-1]) with atol 1e-8 (eight_point.py:268)
    const double s2_true = s2v / scale;   // exact: scale is a power of two
    const int st = (s2_true <= 1e-8 + 1e-5 * s2_true) ? 0 : 1;
    // R1 = U W^T V^T, R2 = U W V^T with W = [[0,-1,0],[1,0,0],[0,0,1]]:
    //   U W^T = [-u2, u1, u3] columns -> R1 = -u2 v1^T + u1 v2^T + u3 v3^T
    //   U W   = [ u2,-u1, u3]         -> R2 =  u2 v1^T - u1 v2^T + u3 v3^T
    if (active) {
        double* out = pose_rt + b * 48;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double r1 = (-(u[1][r] * vt[0][c]) + u[0][r] * vt[1][c]) + u[2][r] * vt[2][c];
                const double r2 = (u[1][r] * vt[0][c] - u[0][r] * vt[1][c]) + u[2][r] * vt[2][c];
                out[0 * 12 + r * 3 + c] = r1;
                out[1 * 12 + r * 3 + c] = r1;
                out[2 * 12 + r * 3 + c] = r2;
                out[3 * 12 + r * 3 + c] = r2;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // t = vee(U Z U^T) = u1 x u2 = u3
            out[0 * 12 + 9 + r] = u[2][r];
            out[1 * 12 + 9 + r] = -u[2][r];
            out[2 * 12 + 9 + r] = u[2][r];
            out[3 * 12 + 9 + r] = -u[2][r];
        }
        status[b] = st;
    }
}

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================
extern "C" {

const char* sfm_last_error(void) { return sfmhost::error_buffer(); }
int sfm_abi_version(void) { return SFM_ABI_VERSION; }

int sfm_normalize_correspondences(const double* pix_a, const double* pix_b, int64_t count, double fx,
                                  double fy, double cx, double cy, double* corr, void* stream) {
    if (count < 0) return fail(SFM_EINVAL, "sfm_normalize_correspondences: negative count");
    if (count == 0) return SFM_OK;
    if (!pix_a || !pix_b || !corr) return fail(SFM_EINVAL, "sfm_normalize_correspondences: null pointer");
    hipLaunchKernelGGL(normalize_kernel, dim3(grid_stride(count, 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, (const double2*)pix_a, (const double2*)pix_b, count, fx, fy,
                       cx, cy, (Corr*)corr);
    return check_launch("normalize_kernel");
}

int sfm_sample_philox(uint64_t seed, uint64_t seed_stride, int64_t h_begin, int64_t h_count, int64_t n,
                      int64_t batch, int32_t* S, void* stream) {
    if (h_count < 0 || batch < 0 || h_begin < 0) return fail(SFM_EINVAL, "sfm_sample_philox: negative size");
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_sample_philox: need 8 <= n < 2^31");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!S) return fail(SFM_EINVAL, "sfm_sample_philox: null pointer");
    SFM_REQUIRE_GRID("sfm_sample_philox", h_count, 256, 256, batch);
    hipLaunchKernelGGL(sample_philox_kernel, dim3(grid_for(h_count, 256), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, seed, seed_stride, h_begin, h_count, (uint32_t)n, S);
    return check_launch("sample_philox_kernel");
}

int sfm_sample_philox_dev(const uint64_t* seed_dev, uint64_t seed_stride, int64_t h_begin, int64_t h_count,
                          int64_t n, int64_t batch, int32_t* S, void* stream) {
    if (h_count < 0 || batch < 0 || h_begin < 0) return fail(SFM_EINVAL, "sfm_sample_philox_dev: negative size");
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_sample_philox_dev: need 8 <= n < 2^31");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!S || !seed_dev) return fail(SFM_EINVAL, "sfm_sample_philox_dev: null pointer");
    SFM_REQUIRE_GRID("sfm_sample_philox_dev", h_count, 256, 256, batch);
    hipLaunchKernelGGL(sample_philox_dev_kernel, dim3(grid_for(h_count, 256), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, seed_dev, seed_stride, h_begin, h_count, (uint32_t)n, S);
    return check_launch("sample_philox_dev_kernel");
}

int sfm_sample_philox_at(uint64_t seed, uint64_t seed_stride, const int64_t* h_index, int64_t n,
                         int64_t batch, int32_t* S, void* stream) {
    if (batch < 0) return fail(SFM_EINVAL, "sfm_sample_philox_at: negative size");
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_sample_philox_at: need 8 <= n < 2^31");
    if (batch == 0) return SFM_OK;
    if (!S || !h_index) return fail(SFM_EINVAL, "sfm_sample_philox_at: null pointer");
    SFM_REQUIRE_GRID("sfm_sample_philox_at", batch, 64, 64);
    hipLaunchKernelGGL(sample_philox_at_kernel, dim3(grid_for(batch, 64)), dim3(64), 0, (hipStream_t)stream,
                       seed, seed_stride, h_index, (uint32_t)n, batch, S);
    return check_launch("sample_philox_at_kernel");
}

int sfm_fit_eight_point(const double* corr, int64_t n, const int32_t* S, int64_t h_count, int64_t batch,
                        double* E, int32_t* flags, double* lambda2, void* stream) {
    if (h_count < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_fit_eight_point: negative size");
    if (n < 8) return fail(SFM_EINVAL, "sfm_fit_eight_point: need at least 8 correspondences");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!corr || !S || !E || !flags) return fail(SFM_EINVAL, "sfm_fit_eight_point: null pointer");
    SFM_REQUIRE_GRID("sfm_fit_eight_point", h_count, kWave, kWave, batch);
    hipLaunchKernelGGL(fit_eight_point_kernel<false>, dim3(grid_for(h_count, kWave), (unsigned)batch),
                       dim3(kWave), 0, (hipStream_t)stream, (const Corr*)corr, n, const_cast<int32_t*>(S), h_count, E,
                       flags, lambda2, (double*)nullptr, PhiloxSource{nullptr, 0, 0, 0, 0}, SmallPrep{nullptr, 0.0, 0, nullptr}, MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0});
    return check_launch("fit_eight_point_kernel");
}

int sfm_sample_fit_philox(uint64_t seed, const uint64_t* seed_dev, uint64_t seed_stride, int64_t h_begin,
                          const double* corr, int64_t n, int64_t h_count, int64_t batch, int32_t* S, double* E,
                          int32_t* flags, void* stream) {
    if (h_count < 0 || batch < 0 || h_begin < 0) return fail(SFM_EINVAL, "sfm_sample_fit_philox: negative size");
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_sample_fit_philox: need 8 <= n < 2^31");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!corr || !S || !E || !flags) return fail(SFM_EINVAL, "sfm_sample_fit_philox: null pointer");
    SFM_REQUIRE_GRID("sfm_fit_eight_point", h_count, kWave, kWave, batch);
    hipLaunchKernelGGL(fit_eight_point_kernel<false>, dim3(grid_for(h_count, kWave), (unsigned)batch),
                       dim3(kWave), 0, (hipStream_t)stream, (const Corr*)corr, n, S, h_count, E, flags,
                       (double*)nullptr, (double*)nullptr, PhiloxSource{seed_dev, seed, seed_stride, h_begin, 1},
                       SmallPrep{nullptr, 0.0, 0, nullptr}, MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0});
    return check_launch("fit_eight_point_kernel (philox)");
}

int sfm_ransac_pass_small(uint64_t seed, const uint64_t* seed_dev, int use_philox, int64_t h_begin, const double* corr,
                          int64_t n, int64_t h_count, double thr, double min_extra, int aggregation, int64_t h_offset,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2,
                          sfm_select_result* result, uint8_t* mask, void* workspace, int64_t workspace_bytes,
                          void* stream, const sfm_score_options* options) {
    if (n < 8 || n > sfmws::kSmallMaxPoints)
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: need 8 <= n <= 8192 (larger point sets: the separate calls)");
    if (h_count < 1 || h_count > 32 * 1024)
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: need 1 <= h_count <= 32768 (more hypotheses: the separate calls)");
    if (h_begin < 0) return fail(SFM_EINVAL, "sfm_ransac_pass_small: negative h_begin");
    if (aggregation < SFM_AGG_SUM || aggregation > SFM_AGG_RMS)
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: unknown aggregation");
    if (!corr || !S || !E || !flags || !cnt || !s1 || !s2 || !result || !workspace)
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: null pointer");
    if (!sfmhost::score_options_valid(options)) return fail(SFM_EINVAL, "sfm_ransac_pass_small: an option is out of range");
    if (workspace_bytes < sfm_score_workspace_bytes_ex(n, h_count, 1, options))
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: workspace smaller than sfm_score_workspace_bytes_ex(n, h_count, 1, options)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_ransac_pass_small: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // launch 1: eight-point fits (Philox samples drawn in the kernel, or the caller's table in S) + workspace preparation
    const int prep_blocks = (int)((n + sfmws::kPrepPoints - 1) / sfmws::kPrepPoints);
    const unsigned fit_blocks = grid_for(h_count, kWave);
    hipLaunchKernelGGL(fit_eight_point_kernel<false>, dim3(fit_blocks + (unsigned)prep_blocks, 1u), dim3(kWave), 0, st,
                       (const Corr*)corr, n, S, h_count, E, flags, (double*)nullptr, (double*)nullptr,
                       PhiloxSource{seed_dev, seed, 0, h_begin, use_philox ? 1 : 0},
                       SmallPrep{static_cast<unsigned char*>(workspace), sfmhost::small_pass_a_scale(thr), prep_blocks,
                                 sfmhost::small_pass_order(static_cast<unsigned char*>(workspace), n, h_count)},
                       MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0});
    const int rc = check_launch("fit_eight_point_kernel (fused small pass)");
    if (rc != SFM_OK) return rc;
    // launch 2: SED scoring
    const int rc2 = sfmhost::launch_small_score(sfmhost::SmallPass{corr, n, E, S, flags, h_count, thr, min_extra, aggregation,
                                                                   h_offset, cnt, s1, s2, result, mask,
                                                                   static_cast<unsigned char*>(workspace), st, options});
    if (rc2 != SFM_OK) return rc2;
    // launch 3: selection spread over up to 32 blocks x 256 threads x 4 hypotheses, folded by the block that arrives last
    // (a single block needs 14 us at 10 000 and 35 us at 30 000 hypotheses), and — behind them in the same launch — the
    // blocks that write the winner's inlier mask once the record is published
    if (h_count <= 4096 && n <= 4096) {  // one block does both: 7 us instead of 10 at 300 x 2000
        hipLaunchKernelGGL(select_block_mask_kernel, dim3(1), dim3(kSelectBlock), 0, st, (const int32_t*)cnt,
                           (const double*)s1, (const double*)s2, (const int32_t*)flags, h_count, h_offset, min_extra,
                           aggregation, result, (const Corr*)corr, n, (const double*)E, (const int32_t*)S, thr, mask);
        return check_launch("select_block_mask_kernel");
    }
    const int select_blocks = (int)((h_count + 1023) / 1024);
    const int mask_blocks = mask != nullptr ? (int)((n + 255) / 256) : 0;
    unsigned char* state = static_cast<unsigned char*>(workspace) + sfmws::ws_points_offset(1) + 16 * n;
    hipLaunchKernelGGL(select_sharded_kernel, dim3((unsigned)(select_blocks + mask_blocks)), dim3(256), 0, st,
                       (const int32_t*)cnt, (const double*)s1, (const double*)s2, (const int32_t*)flags, h_count, h_offset,
                       min_extra, aggregation, state, result, select_blocks, (const Corr*)corr, n, (const double*)E,
                       (const int32_t*)S, thr, mask);
    return check_launch("select_sharded_kernel");
}

int sfm_ransac_pass_large(uint64_t seed, const uint64_t* seed_dev, int use_philox, int64_t h_begin, const double* corr,
                          int64_t n, int64_t h_count, double thr, double min_extra, int aggregation, int64_t h_offset,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2,
                          sfm_select_result* result, uint8_t* mask, void* workspace, int64_t workspace_bytes,
                          void* stream, const sfm_score_options* options) {
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_ransac_pass_large: need 8 <= n < 2^31");
    if (h_count < 1 || h_count > 0x3FFFFFFF) return fail(SFM_EINVAL, "sfm_ransac_pass_large: need 1 <= h_count < 2^30");
    if (h_begin < 0) return fail(SFM_EINVAL, "sfm_ransac_pass_large: negative h_begin");
    if (aggregation < SFM_AGG_SUM || aggregation > SFM_AGG_RMS)
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: unknown aggregation");
    if (!corr || !S || !E || !flags || !cnt || !s1 || !s2 || !result || !workspace)
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: null pointer");
    if (!sfmhost::score_options_valid(options)) return fail(SFM_EINVAL, "sfm_ransac_pass_large: an option is out of range");
    if (workspace_bytes < sfm_score_workspace_bytes_ex(n, h_count, 1, options))
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: workspace smaller than sfm_score_workspace_bytes_ex(n, h_count, 1, options)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: workspace must be 16-byte aligned");
    SFM_REQUIRE_GRID("sfm_ransac_pass_large", h_count, kWave, kWave, 1);
    SFM_REQUIRE_GRID("sfm_ransac_pass_large (mask)", n, 256, 256);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    unsigned char* state = ws + sfmws::ws_tail_offset(n, h_count, 1);   // the unused head of the range-split region
    const bool state_fits = 4 * sfmws::split_padded(h_count) >= sfmws::kFusedPartialOffset + kLargeSelectBlocks * (int64_t)sizeof(PartialSelect);
    sfmhost::LargePass pass{corr, n, E, S, h_count, thr, cnt, s1, s2, ws, workspace_bytes,
                            state_fits ? reinterpret_cast<unsigned*>(state) : nullptr, st, options};
    // launch 1 (where the matrix-pipe scoring kernel will run): partial maxima of the points + every zeroing the pass needs
    sfmhost::MatrixTables tables;
    int rc = sfmhost::launch_large_setup(pass, &tables);
    if (rc != SFM_OK) return rc;
    // launch 2: the eight-point fits (Philox samples drawn in the kernel, or the caller's table in S); with the matrix-pipe
    // kernel ahead, every fit lane also writes its hypothesis' operand rows and sample correction, and blocks behind the fit
    // blocks the point operand table (MatrixPrep)
    const MatrixPrep prep = tables.matrix ? MatrixPrep{tables.partial, tables.partials, tables.a_scale, thr, tables.hyp_table, tables.fix,
                                                       tables.table, tables.step_blocks}
                                          : MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0};
    if (!sfmhost::grid_fits((int64_t)grid_for(h_count, kWave) + prep.step_blocks, 1, kWave))
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: size exceeds what one launch covers");
    if (tables.matrix)
        hipLaunchKernelGGL((fit_eight_point_kernel<false, true>), dim3(grid_for(h_count, kWave) + (unsigned)prep.step_blocks, 1u), dim3(kWave),
                           0, st, (const Corr*)corr, n, S, h_count, E, flags, (double*)nullptr, (double*)nullptr,
                           PhiloxSource{seed_dev, seed, 0, h_begin, use_philox ? 1 : 0}, SmallPrep{nullptr, 0.0, 0, nullptr}, prep);
    else
        hipLaunchKernelGGL((fit_eight_point_kernel<false, false>), dim3(grid_for(h_count, kWave), 1u), dim3(kWave), 0, st, (const Corr*)corr,
                           n, S, h_count, E, flags, (double*)nullptr, (double*)nullptr,
                           PhiloxSource{seed_dev, seed, 0, h_begin, use_philox ? 1 : 0}, SmallPrep{nullptr, 0.0, 0, nullptr}, prep);
    rc = check_launch("fit_eight_point_kernel (fused large pass)");
    if (rc != SFM_OK) return rc;
    pass.tables_ready = tables.matrix;
    // then sfm_score_sed's launches — cost pre-pass, class count, scan + scatter, the scoring kernel (other sizes: that call's
    // own preparation first) — with the ranges' partials left unfolded
    sfmhost::LargeScore folded_later{1, nullptr, nullptr};
    // (without room for the selection state the scoring call folds its ranges itself — folded_later = NULL —: the separate
    // selection below reads cnt / s1 / s2.  Round 4 deferred the fold in that case too, and selected from unfolded partials.)
    rc = sfmhost::launch_large_score(pass, state_fits ? &folded_later : nullptr);
    if (rc != SFM_OK) return rc;
    if (!state_fits) {   // a few hundred hypotheses: the separate selection and mask launches
        rc = sfm_select_best(cnt, s1, s2, flags, h_count, 1, min_extra, aggregation, h_offset, result, stream);
        if (rc != SFM_OK || mask == nullptr) return rc;
        return sfm_inlier_mask(corr, n, E, S, h_count, 1, result, thr, mask, stream);
    }
    // launch 8: fold of the ranges + selection over up to 256 blocks + (behind them) the blocks that write the winner's mask
    const int select_blocks = (int)std::min<int64_t>(kLargeSelectBlocks, (h_count + 511) / 512);
    const int mask_blocks = mask != nullptr ? (int)((n + 255) / 256) : 0;
    hipLaunchKernelGGL(select_large_kernel, dim3((unsigned)(select_blocks + mask_blocks)), dim3(256), 0, st, cnt, s1, s2,
                       (const int32_t*)flags, h_count, h_offset, min_extra, aggregation, state, result, select_blocks,
                       folded_later.units, (const unsigned char*)folded_later.split, folded_later.fix, (const Corr*)corr, n,
                       (const double*)E, (const int32_t*)S, thr, mask);
    return check_launch("select_large_kernel");
}

int sfm_ransac_pass_batch(uint64_t seed, const uint64_t* seed_dev, uint64_t seed_stride, int use_philox, int64_t h_begin,
                          const double* corr, int64_t n, int64_t h_count, int64_t batch, double thr, double min_extra, int aggregation,
                          int32_t* S, double* E, int32_t* flags, int32_t* cnt, double* s1, double* s2, sfm_select_result* result,
                          uint8_t* mask, void* workspace, int64_t workspace_bytes, void* stream, const sfm_score_options* options) {
    if (n < 8 || n > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: need 8 <= n < 2^31");
    if (h_count < 1 || h_count > 0x3FFFFFFF) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: need 1 <= h_count < 2^30");
    if (batch < 0 || batch > 65535) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: need 0 <= batch <= 65535");
    if (h_begin < 0) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: negative h_begin");
    if (aggregation < SFM_AGG_SUM || aggregation > SFM_AGG_RMS) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: unknown aggregation");
    if (batch == 0) return SFM_OK;
    if (!corr || !S || !E || !flags || !cnt || !s1 || !s2 || !result || !workspace)
        return fail(SFM_EINVAL, "sfm_ransac_pass_batch: null pointer");
    if (!sfmhost::score_options_valid(options)) return fail(SFM_EINVAL, "sfm_ransac_pass_batch: an option is out of range");
    if (workspace_bytes < sfm_score_workspace_bytes_ex(n, h_count, batch, options))
        return fail(SFM_EINVAL, "sfm_ransac_pass_batch: workspace smaller than sfm_score_workspace_bytes_ex(n, h_count, batch, options)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_ransac_pass_batch: workspace must be 16-byte aligned");
    SFM_REQUIRE_GRID("sfm_ransac_pass_batch", h_count, kWave, kWave, batch);
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    sfmhost::LargePass pass{corr, n, E, S, h_count, thr, cnt, s1, s2, ws, workspace_bytes, nullptr, st, options, batch};
    // launches 1-2 (where the matrix-pipe scoring kernel will run): partial maxima + zeroing, the pairs' point operand tables
    sfmhost::MatrixTables tables;
    int rc = sfmhost::launch_large_setup(pass, &tables);
    if (rc != SFM_OK) return rc;
    // launch 3: the eight-point fits; with the matrix-pipe kernel ahead every lane also writes its hypothesis' operand rows and
    // sample correction
    const PhiloxSource source{seed_dev, seed, seed_stride, h_begin, use_philox ? 1 : 0};
    const dim3 fit_grid(grid_for(h_count, kWave), (unsigned)batch);
    if (tables.matrix)
        hipLaunchKernelGGL((fit_eight_point_kernel<false, true>), fit_grid, dim3(kWave), 0, st, (const Corr*)corr, n, S, h_count, E, flags,
                           (double*)nullptr, (double*)nullptr, source, SmallPrep{nullptr, 0.0, 0, nullptr},
                           MatrixPrep{tables.partial, tables.partials, tables.a_scale, thr, tables.hyp_table, tables.fix, nullptr, 0});
    else
        hipLaunchKernelGGL((fit_eight_point_kernel<false, false>), fit_grid, dim3(kWave), 0, st, (const Corr*)corr, n, S, h_count, E, flags,
                           (double*)nullptr, (double*)nullptr, source, SmallPrep{nullptr, 0.0, 0, nullptr},
                           MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0});
    rc = check_launch("fit_eight_point_kernel (batched pass)");
    if (rc != SFM_OK) return rc;
    pass.tables_ready = tables.matrix;
    // sfm_score_sed's launches (matrix-pipe kernel: cost pre-pass, class count, scan + scatter, scoring — the ranges left unfolded)
    sfmhost::LargeScore folded_later{1, nullptr, nullptr};
    rc = sfmhost::launch_large_score(pass, &folded_later);
    if (rc != SFM_OK) return rc;
    // last launch: fold of the ranges + selection + mask, one block per pair
    hipLaunchKernelGGL(select_fold_mask_batch_kernel, dim3((unsigned)batch), dim3(kSelectBlock), 0, st, cnt, s1, s2, (const int32_t*)flags,
                       h_count, min_extra, aggregation, result, folded_later.units, (const unsigned char*)folded_later.split,
                       folded_later.fix, (const Corr*)corr, n, (const double*)E, (const int32_t*)S, thr, mask);
    return check_launch("select_fold_mask_batch_kernel");
}

int sfm_fit_trace_doubles(void) { return kTraceDoubles; }

int sfm_fit_eight_point_traced(const double* corr, int64_t n, const int32_t* S, int64_t h_count, int64_t batch,
                               double* E, int32_t* flags, double* trace, void* stream) {
    if (h_count < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_fit_eight_point_traced: negative size");
    if (n < 8) return fail(SFM_EINVAL, "sfm_fit_eight_point_traced: need at least 8 correspondences");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!corr || !S || !E || !flags || !trace) return fail(SFM_EINVAL, "sfm_fit_eight_point_traced: null pointer");
    SFM_REQUIRE_GRID("sfm_fit_eight_point_traced", h_count, kWave, kWave, batch);
    hipLaunchKernelGGL(fit_eight_point_kernel<true>, dim3(grid_for(h_count, kWave), (unsigned)batch),
                       dim3(kWave), 0, (hipStream_t)stream, (const Corr*)corr, n, const_cast<int32_t*>(S), h_count, E,
                       flags, (double*)nullptr, trace, PhiloxSource{nullptr, 0, 0, 0, 0}, SmallPrep{nullptr, 0.0, 0, nullptr}, MatrixPrep{nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, 0});
    return check_launch("fit_eight_point_kernel<trace>");
}

int sfm_fit_stage(int stage, const double* in, double* out, void* stream) {
    if (stage < 0 || stage > 2) return fail(SFM_EINVAL, "sfm_fit_stage: stage must be 0, 1 or 2");
    if (!in || !out) return fail(SFM_EINVAL, "sfm_fit_stage: null pointer");
    hipLaunchKernelGGL(fit_stage_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, stage, in, out);
    return check_launch("fit_stage_kernel");
}

int sfm_hartley_normalize(const double* coords, int64_t n, double* out, void* stream) {
    if (n <= 0) return fail(SFM_EINVAL, "sfm_hartley_normalize: need at least one point");
    if (!coords || !out) return fail(SFM_EINVAL, "sfm_hartley_normalize: null pointer");
    hipLaunchKernelGGL(hartley_normalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, coords, n, out);
    return check_launch("hartley_normalize_kernel");
}

int sfm_select_best(const int32_t* cnt, const double* s1, const double* s2, const int32_t* flags,
                    int64_t h_count, int64_t batch, double min_extra, int aggregation, int64_t h_offset,
                    sfm_select_result* result, void* stream) {
    if (h_count < 0 || batch < 0) return fail(SFM_EINVAL, "sfm_select_best: negative size");
    if (aggregation < SFM_AGG_SUM || aggregation > SFM_AGG_RMS)
        return fail(SFM_EINVAL, "sfm_select_best: unknown aggregation");
    if (batch == 0) return SFM_OK;
    if (batch > 65535) return fail(SFM_EINVAL, "sfm_select_best: batch > 65535");
    if (!result || (h_count > 0 && (!cnt || !s1 || !s2)))
        return fail(SFM_EINVAL, "sfm_select_best: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (h_count <= 32 * kSelectBlock) {  // <= 32 hypotheses per thread: one block per batch entry, one launch
        hipLaunchKernelGGL(select_block_kernel, dim3((unsigned)batch), dim3(kSelectBlock), 0, st, cnt, s1, s2, flags,
                           h_count, h_offset, min_extra, aggregation, result);
        return check_launch("select_block_kernel");
    }
    hipLaunchKernelGGL(select_init_kernel, dim3((unsigned)batch), dim3(1), 0, st, result);
    if (h_count > 0) {
        const dim3 grid(grid_stride(h_count, 256, 64), (unsigned)batch);
        hipLaunchKernelGGL(select_pass1_kernel, grid, dim3(256), 0, st, cnt, s1, s2, flags, h_count, min_extra,
                           aggregation, result);
        hipLaunchKernelGGL(select_pass2_kernel, grid, dim3(256), 0, st, cnt, s1, s2, flags, h_count, min_extra,
                           aggregation, result);
    }
    hipLaunchKernelGGL(select_final_kernel, dim3((unsigned)batch), dim3(1), 0, st, cnt, h_count, h_offset, result);
    return check_launch("select_best_kernel");
}

int sfm_inlier_mask(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                    int64_t batch, const sfm_select_result* result, double thr, uint8_t* mask,
                    void* stream) {
    if (h_count < 0 || batch < 0 || n < 0) return fail(SFM_EINVAL, "sfm_inlier_mask: negative size");
    if (n == 0 || batch == 0) return SFM_OK;
    if (!corr || !E || !S || !result || !mask) return fail(SFM_EINVAL, "sfm_inlier_mask: null pointer");
    SFM_REQUIRE_GRID("sfm_inlier_mask", 1, 1, 256, batch);
    hipLaunchKernelGGL(inlier_mask_kernel, dim3(grid_stride(n, 256, 1024), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, (const Corr*)corr, n, E, S, h_count, result, thr, mask);
    return check_launch("inlier_mask_kernel");
}

int sfm_sed_values(const double* corr, int64_t n, const double* E, double* out, void* stream) {
    if (n < 0) return fail(SFM_EINVAL, "sfm_sed_values: negative size");
    if (n == 0) return SFM_OK;
    if (!corr || !E || !out) return fail(SFM_EINVAL, "sfm_sed_values: null pointer");
    hipLaunchKernelGGL(sed_values_kernel, dim3(grid_stride(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const Corr*)corr, n, E, out);
    return check_launch("sed_values_kernel");
}

int sfm_cheirality(const double* corr, int64_t m, const double* pose_rt, int64_t poses,
                   double distance_threshold, uint8_t* pass, void* stream) {
    if (m < 0 || poses < 0) return fail(SFM_EINVAL, "sfm_cheirality: negative size");
    if (m == 0 || poses == 0) return SFM_OK;
    if (!corr || !pose_rt || !pass) return fail(SFM_EINVAL, "sfm_cheirality: null pointer");
    SFM_REQUIRE_GRID("sfm_cheirality", m, kWave, kWave, poses);
    hipLaunchKernelGGL(cheirality_kernel, dim3(grid_for(m, kWave), (unsigned)poses), dim3(kWave), 0,
                       (hipStream_t)stream, (const Corr*)corr, m, pose_rt, distance_threshold, pass);
    return check_launch("cheirality_kernel");
}

int sfm_triangulate(const double* corr, int64_t m, const double* P1, const double* P2, double* X,
                    void* stream) {
    if (m < 0) return fail(SFM_EINVAL, "sfm_triangulate: negative size");
    if (m == 0) return SFM_OK;
    if (!corr || !P1 || !P2 || !X) return fail(SFM_EINVAL, "sfm_triangulate: null pointer");
    SFM_REQUIRE_GRID("sfm_triangulate", m, kWave, kWave);
    hipLaunchKernelGGL(triangulate_kernel, dim3(grid_for(m, kWave)), dim3(kWave), 0, (hipStream_t)stream,
                       (const Corr*)corr, m, P1, P2, X);
    return check_launch("triangulate_kernel");
}

int sfm_decompose_essential(const double* E, int64_t batch, double* pose_rt, int32_t* status,
                            void* stream) {
    if (batch < 0) return fail(SFM_EINVAL, "sfm_decompose_essential: negative size");
    if (batch == 0) return SFM_OK;
    if (!E || !pose_rt || !status) return fail(SFM_EINVAL, "sfm_decompose_essential: null pointer");
    SFM_REQUIRE_GRID("sfm_decompose_essential", batch, kWave, kWave);
    hipLaunchKernelGGL(decompose_essential_kernel, dim3(grid_for(batch, kWave)), dim3(kWave), 0,
                       (hipStream_t)stream, E, batch, pose_rt, status);
    return check_launch("decompose_essential_kernel");
}

}  // extern "C"
