// Layout of the scoring workspace (sfm_score_workspace_bytes) and the device routines that fill it, shared by the
// scoring kernels (sfm_score.hip) and by the fit kernel (sfm_kernels.hip), which prepares the workspace of a small
// fused pass in spare blocks of its own launch.
//
//   [batch x 4 uint32 maxima][batch x n float4 points][kPointsPad bytes][batch x kBuckets int32][batch x h_count int32]
//   then — sized by what the call launches (WsPlan, below) — the range-split region and the matrix-pipe kernel's operand tables
//
// maxima: data-set maxima of |xa'|, |ya'|, |xb|, |yb| of the fp32 points as bit patterns (non-negative floats order
// like unsigned ints).  buckets: per pair 256 ints — class counters of the longest-first ordering in large launches;
// in a fused small pass (which never orders) the same words carry the per-block partial maxima and the arrival ticket.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfm_common.h"

namespace sfmws {

constexpr int kEstimatePoints = 1024;        // points scanned by the cost pre-pass
constexpr int kClasses = 16 * 15 + 1;        // cost classes: sixteen per power of two of the estimate (heaviest first), 0 survivors last
constexpr int kClassStride = 16;             // ints between class counters: one 64-byte line each
constexpr int kBuckets = 256 * kClassStride; // ints reserved per batch entry (16 KiB)
constexpr int64_t kPointsPad = 4096;         // bytes after the fp32 points: the scoring loop prefetches up to 3 KiB past a pair's last point

// fused small pass (n <= kSmallMaxPoints): the fit launch prepares the points in blocks of kPrepPoints
constexpr int kPrepPoints = 512;             // 64 lanes x 8 points, all loads of a lane in flight together
constexpr int kSmallMaxPoints = 8192;
constexpr int kMaxPrepBlocks = kSmallMaxPoints / kPrepPoints;  // 16 partial maxima x 4 words = buckets[0, 64)
static_assert(4 * kMaxPrepBlocks <= kBuckets, "the partial maxima live in the class-counter words");
// state of a small pass's sharded selection (select_sharded_kernel, sfm_kernels.hip) in the kPointsPad bytes behind
// the fp32 points: one 64-byte line with the arrival counter and the "record published" flag, then one 32-byte partial
// record per selecting block
constexpr int kFusedShards = 32;             // selecting blocks: 32 x 256 threads x 4 hypotheses = 32768
constexpr int kFusedPartialOffset = 64;      // bytes: partial records behind the counter's line

__host__ __device__ inline int64_t ws_points_offset(int64_t batch) { return 16 * batch; }
__host__ __device__ inline int64_t ws_buckets_offset(int64_t n, int64_t batch) {
    return 16 * batch + 16 * n * batch + kPointsPad;
}
__host__ __device__ inline int64_t ws_order_offset(int64_t n, int64_t batch) {
    return ws_buckets_offset(n, batch) + ((4 * (int64_t)kBuckets * batch + 15) / 16) * 16;
}
// Behind the scoring order the workspace is sized by what the call will actually launch (WsPlan: round 5; until then every
// region was reserved whether or not a launch could pick it — 521 MB for 256 pairs x 10 000 x 2 000).
struct WsPlan {
    bool matrix;   // the matrix-pipe kernel runs: its operand tables and sample corrections are reserved
    int units;     // ranges of the points the scoring launch is cut into (1: none)
    bool record;   // one pair, matrix-pipe kernel: the cost pre-pass hands its reject words to the scoring launch
};
// Range split (sfm_score.hip): a wave's hypotheses are scored over `units` ranges of the points by different waves, which leave
// their partial counts and sums at [range][hypothesis]:  [h_pad int32 head][units x h_pad int32][units x h_pad f64][units x h_pad f64]
// per pair.  The head is unused by the scoring kernels; for one pair it holds the state of a fused pass's selection launch.
constexpr int kSplitMaxUnits = 16;
__host__ __device__ inline int64_t split_padded(int64_t h_count) { return (h_count + 3) & ~(int64_t)3; }
__host__ __device__ inline int64_t ws_tail_offset(int64_t n, int64_t h_count, int64_t batch) {   // behind the scoring order(s)
    return ((ws_order_offset(n, batch) + 4 * h_count * batch + 15) / 16) * 16;
}
__host__ __device__ inline int64_t split_bytes(int64_t h_count, int units) {   // one pair
    return split_padded(h_count) * (4 + (int64_t)(units > 1 ? units : 0) * (4 + 8 + 8));
}
__host__ __device__ inline int64_t split_region_bytes(int64_t h_count, int64_t batch, int units) {
    return batch == 1 ? split_bytes(h_count, units) : (units > 1 ? batch * split_bytes(h_count, units) : 0);
}
// The partials are written with plain stores by the scoring kernels and added in range order by matrixscore::matrix_fold_kernel
// (or the selection launch of a fused pass).
// Operand tables of the matrix-pipe kernel (sfm_score_matrix.h; at most kMatrixMaxPoints points per pair), behind the split
// region: per pair and step of 32 points three blocks of 64 lanes x 16 bytes (96 bytes per point), then per pair and hypothesis
// 2 halves x 3 blocks x 16 bytes, then per pair the sample corrections of the hypotheses ([h_pad] int32 | [h_pad] f64 | [h_pad] f64).
constexpr int64_t kMatrixMaxPoints = 1 << 22;   // 4 M points per pair (a 400 MB operand table); until round 4: 65 536 (absolute steps in 16 bits)
__host__ __device__ inline int64_t matrix_table_steps(int64_t n) { return (((n + 31) / 32) + 3) & ~(int64_t)3; }   // (with pad steps: sfm_score_matrix.h)
__host__ __device__ inline int64_t matrix_table_bytes(int64_t n) { return matrix_table_steps(n) * 3 * 64 * 16; }   // one pair
__host__ __device__ inline int64_t matrix_hyp_table_bytes(int64_t h_count) { return h_count * 96; }                // one pair
__host__ __device__ inline int64_t matrix_fix_bytes(int64_t h_count) { return split_padded(h_count) * (4 + 8 + 8); }
__host__ __device__ inline int64_t ws_matrix_offset(int64_t n, int64_t h_count, int64_t batch, const WsPlan& plan) {
    return ((ws_tail_offset(n, h_count, batch) + split_region_bytes(h_count, batch, plan.units) + 255) / 256) * 256;
}
__host__ __device__ inline int64_t ws_matrix_hyp_offset(int64_t n, int64_t h_count, int64_t batch, const WsPlan& plan) {
    return ((ws_matrix_offset(n, h_count, batch, plan) + (plan.matrix ? batch * matrix_table_bytes(n) : 0) + 255) / 256) * 256;
}
__host__ __device__ inline int64_t ws_matrix_fix_offset(int64_t n, int64_t h_count, int64_t batch, const WsPlan& plan) {
    return ((ws_matrix_hyp_offset(n, h_count, batch, plan) + (plan.matrix ? batch * matrix_hyp_table_bytes(h_count) : 0) + 255) / 256) * 256;
}
// ... and, for a single pair, the reject words the cost pre-pass records for the scoring launch (sfm_score_matrix.h, MatrixPair::record):
// [range][2 chunks of 8 steps][h_pad][2 halves] x 16 bytes (the first 16 steps of every range; the launcher records only when the
// ranges x 16 steps are exactly the pre-pass' steps: at most SFM_MATRIX_ESTIMATE_STEPS / 16 = 8 ranges)
#ifndef SFM_MATRIX_ESTIMATE_STEPS
#define SFM_MATRIX_ESTIMATE_STEPS 128   // steps of 32 points the matrix-pipe kernel's cost pre-pass scans at most (4096 points)
#endif
constexpr int kMatrixReplaySteps = 16;
__host__ __device__ inline int64_t matrix_record_bytes(int64_t h_count) {   // one pair
    return split_padded(h_count) * (int64_t)((SFM_MATRIX_ESTIMATE_STEPS / kMatrixReplaySteps) * (kMatrixReplaySteps / 8) * 2 * 16);
}
__host__ __device__ inline int64_t ws_matrix_record_offset(int64_t n, int64_t h_count, int64_t batch, const WsPlan& plan) {
    return ((ws_matrix_fix_offset(n, h_count, batch, plan) + (plan.matrix ? batch * matrix_fix_bytes(h_count) : 0) + 255) / 256) * 256;
}
__host__ __device__ inline int64_t workspace_bytes_for(int64_t n, int64_t h_count, int64_t batch, const WsPlan& plan) {
    return ws_matrix_record_offset(n, h_count, batch, plan) + (plan.record ? matrix_record_bytes(h_count) : 0);
}
// Work counters of the matrix-pipe kernel's persistent waves (one per XCD, a 64-byte line each), in the words of the class-counter
// block that no class uses (classes end at int 16 * 240 = 3840); score_reset_kernel zeroes them with the class counters.
constexpr int kWideFromWord = 3844;         // int: entry of the scoring order where the waves of 64 hypotheses begin (sfm_score_matrix.h: WIDE_WAVES), written by the sort
constexpr int kTicketWords = 3968;          // ints [3968, 3968 + 8 * 16): the scoring launch
constexpr int kTicketWordsPrepass = 3848;   // ints [3848 + 8 x], x < 8: the cost pre-pass (eight 32-byte slots)
static_assert(16 * (kClasses - 1) < kWideFromWord && kWideFromWord < kTicketWordsPrepass, "a free word behind the class counters");
static_assert(16 * (kClasses - 1) < kTicketWordsPrepass && kTicketWordsPrepass + 8 * 8 <= kTicketWords &&
              kTicketWords + 8 * 16 <= kBuckets, "the work counters sit between the class counters and the end of the block");

// fp32 record of one correspondence as tier 1 reads it; a_scale: 1 for the two-sided test, c ~ 1/sqrt(T) for the
// one-sided one (reject_mask_one_sided in sfm_score.hip)
__device__ __forceinline__ float4 to_filter_point(const Corr& p, double a_scale) {
    return make_float4((float)(p.xa * a_scale), (float)(p.ya * a_scale), (float)p.xb, (float)p.yb);
}

// One 64-lane block of a fused small pass prepares points [block * 512, ...): fp32 copies and this block's partial
// maxima; block 0 also zeroes the arrival counter of the pass's selection launch (kernel boundaries order them).
// NaN coordinates: fmaxf ignores them; such points fail every filter comparison and are decided by the exact tier.
__device__ __forceinline__ void prepare_small_block(const Corr* __restrict__ pts, int n, double a_scale,
                                                    unsigned char* __restrict__ ws, int block) {
    const int lane = threadIdx.x & (kWave - 1);
    float4* __restrict__ out = reinterpret_cast<float4*>(ws + ws_points_offset(1));
    int32_t* buckets = reinterpret_cast<int32_t*>(ws + ws_buckets_offset(n, 1));
    Corr p[8];
    const int base = block * kPrepPoints + lane;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = pts[min(base + j * kWave, n - 1)];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = base + j * kWave;
        if (i < n) {
            const float4 q = to_filter_point(p[j], a_scale);
            out[i] = q;
            m0 = fmaxf(m0, fabsf(q.x));
            m1 = fmaxf(m1, fabsf(q.y));
            m2 = fmaxf(m2, fabsf(q.z));
            m3 = fmaxf(m3, fabsf(q.w));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
        m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
        m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
    }
    if (lane == 0) {
        uint32_t* partial = reinterpret_cast<uint32_t*>(buckets) + 4 * block;
        partial[0] = __float_as_uint(m0);
        partial[1] = __float_as_uint(m1);
        partial[2] = __float_as_uint(m2);
        partial[3] = __float_as_uint(m3);
    }
    if (block == 0 && lane < 16)   // the line holding the arrival counter and the "record published" flag
        reinterpret_cast<unsigned*>(ws + ws_points_offset(1) + 16 * (int64_t)n)[lane] = 0u;
}

}  // namespace sfmws
