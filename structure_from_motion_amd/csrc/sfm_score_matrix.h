// Scoring kernel with tier 1 on the matrix pipe (included by sfm_score.hip; same results as its two other kernels).
//
// Tier 1 of score_sed_filtered_kernel is two small contractions per (point, hypothesis):
//   r'[i,h] = sum_t m_t(i) E_t(h),  m = (xb xa', xb ya', xb c, yb xa', yb ya', yb c, xa', ya', c)          (9 terms)
//   dS[i,h] = sum_k q_k(i) g_k(h),  the quadratic forms of dB = lb0^2 + lb1^2 (lb = E^T b) in (xb^2, xb yb, yb^2, xb, yb, 1) and of
//             dA = la0^2 + la1^2 (la = E a) in (xa^2, xa ya, ya^2, xa, ya, 1), summed: by the harmonic-mean inequality
//             sed = r^2 (1 / dA + 1 / dB) >= 4 r^2 / (dA + dB) — tight where dA = dB, never weaker than 2 r^2 / max(dA, dB); it lets
//             1.05 x the true inliers through where the one-sided r^2 / dB of the VALU filter lets 1.42 x (12 terms)
// On v_mfma_f32_32x32x2_f32 they cost more than the 12 VALU instructions they replace (round 2: fp32 MFMA runs at the VALU
// rate).  The 16-bit MFMAs run at 16 x that rate, and a value is the sum of two fp16 values to 2^-22:
//   r'  ~ sum_t  m_hi E_hi + m_hi E_mid + m_mid E_hi                                   27 products, K = 32: 2 x v_mfma_f32_32x32x16_f16
//   dS / 4 <= sum_k bf(q_k) bf(g_k / 4) + slack(h) + eps sum_k |g_k / 4| Q_k            13 products, K = 16: 1 x v_mfma_f32_32x32x16_bf16
// per 32 points x 32 hypotheses: 96 matrix cycles and 2 vector instructions per accumulator register (dS - r^2 in one fma, its
// sign shifted in with an alignbit) + 4 for the push — 36 a step — instead of 12 vector instructions per 64 evaluations.  (Matrix
// and vector work of a SIMD add up on this part, they do not overlap at full clock: profiles/r04/README.md item 9.)  Points are the A operand (rows), hypotheses
// the B operand (columns): lane l holds the results of hypothesis (l & 31) for 16 of a step's 32 points (rows
// (j & 3) + 8 (j >> 2) + 4 (l >> 5), j = accumulator register).  The operand table puts point 16 (l >> 5) + j of the step into
// that row (point_of_row below), so a lane's sixteen results are sixteen CONSECUTIVE points: the exact tier turns a survivor bit
// into an address with one shift-and-add, and a lane's gathers fall into one 512-byte stretch of the correspondences.
//
// Because a lane belongs to ONE hypothesis, everything behind tier 1 is lane-parallel: each lane pushes its survivors onto its
// own LDS queue (one word per step with survivors: the step — relative to the item's range — and its 16 survivor bits;
// slot-major, so the 64 lanes of a push hit 64 banks), and a round of the exact tier has every lane pop kPops points of its own
// queue and evaluate them under its own hypothesis into its own (count, sum, sum of
// squares): no wave reduction anywhere, the two lanes of a hypothesis are added once at the end.  The evaluation is
// sfm::sed_inlier (sfm_math.h): the reference's r, dA, dB bit for bit, one refined reciprocal instead of two IEEE divisions, the
// division sequence itself wherever the value is not clear of the threshold — the decision is always the reference's.  Rounds start when a queue is
// nearly full (looked at once per group of steps, outside the hot loop) and go on until every queue is down to kLow, so with
// hypotheses of similar load in a wave (the heaviest-first order is dealt row-major here) nearly all lanes are busy in every round.
//
// Waves of 64 hypotheses (round 5; one pair in cost order from 8192 hypotheses on, behind the heaviest classes): the three matrix
// instructions of a step run twice — operand rows of columns 0-31, then of columns 32-63 — and one v_permlane32_swap of the two
// reject words gives every lane both halves of ONE hypothesis (lower lanes: first group, upper lanes: second).  Everything
// behind tier 1 is unchanged — one queue, one E, one (count, sum, sum of squares) per lane; entries carry 2 x step + half — and
// the operands of a step are loaded once for 2048 evaluations: that stream (3 KiB per wave and step) keeps a CU's vector L1
// ~90 % busy in waves of 32.  See WIDE_WAVES at the kernel.
//
// The cost pre-pass (the <true> instantiation: tier 1 alone, survivors counted) scans, for a launch of eight ranges, the first 16
// steps of each range and leaves its reject words for the scoring waves, which replay them instead of computing those steps
// again (MatrixPair::record).
//
// Work items and their results (round 4).  An item is (group of 32 hypotheses, range of the points): one wave, placed by block
// index (consecutive blocks = the ranges of one group; a multiple of eight ranges puts range u on XCD u mod 8) or taken from a
// per-XCD counter by persistent waves (sfm_score_options.persistent).  A range leaves its partial (count, sum, sum of squares) at
// [range][hypothesis] with plain stores; matrix_fold_kernel — or the selection launch of a fused pass — adds the ranges in range
// order and then the hypothesis' sample correction (computed once by matrix_hypothesis_kernel).  No atomics: device-scope
// atomics were two thirds of every wave's life in rounds 2-3 (profiles/r04/README.md).
//
// Ranges.  fp16 holds 2^-14 .. 2^16 at full precision, so both sides of the r' chain are scaled by exact powers of two: the
// point terms by s_p (data set: the largest term maximum lands in [2^13, 2^14)), the hypothesis entries by s_h (per
// hypothesis: the largest entry in [2^10, 2^11)); the dB chain (bf16: fp32's range) carries s_p^2 and s_h^2, so the compare
// r''^2 > dB'' is the unscaled one.
//
// Bound (in scaled units).  x = hi + mid + res with hi = fp16(x), mid = fp16(x - hi), |x - hi| <= 2^-11 |x|.  A normal mid is
// rounded relatively, |res| <= 2^-11 |x - hi| <= 2^-22 |x|; a mid below 2^-14 lies on fp16's SUBNORMAL grid of 2^-24 and is
// rounded absolutely, |res| <= 2^-25 — which is within 2^-22 |x| only for |x| >= 2^-3.  Smaller entries exist: they are the small
// terms of a pair whose large terms set the power-of-two scale (pixel-unit coordinates spread the nine products over 2^20), and
// below 2^-14 hi is subnormal too.  So in general
//   |res| <= max(2^-22 |x|, 2^-25),     |mid| <= 2^-11 |x| + 2^-24
// (fp16 subnormals are not flushed by the matrix unit — tools/micro/mfma_bf16_filter.hip probes it).  Dropped products per term
// (x = m_t s_p, y = E_t s_h): mid_x mid_y + res_x y + (hi_x + mid_x) res_y
//   <= 3.1 * 2^-22 |x y|  +  (2^-25 + 2^-35) (|x| + |y|)  +  2^-48.
// Accumulation: every product of two fp16 (subnormal or not) is exact in fp32; the matrix unit truncates each aligned addend
// at the unit of the largest one (probe: <= 3.9 * 2^-23 sum |products| observed), bounded here by one ulp of the largest term per
// addend: 17 * 2^-23 sum |products| per instruction, 34 * 2^-23 for the chain of two.  Together
//   |r''_mfma - r''| <= delta'' = 5.2e-6 sum_t |E_t| s_h M_t s_p  +  1.01 * 2^-25 sum_t (M_t s_p + |E_t| s_h)
// (M_t: data-set maximum of |m_t|).  The second, absolute term was missing until round 4 (round 3's advisor: with coordinates
// of +-10 000 an emulation of the split saw |error| up to 33 delta'' and true inliers rejected); with it the same emulation stays
// below 0.6 delta''.  On K-normalised coordinates it adds ~1e-3 to a delta'' of ~20.  tests/test_gpu_parity.py measures the
// margin per (point, hypothesis) through sfm_debug_matrix_filter.
// dS: bf16 has 8 significant bits, |bf(x) - x| <= 2^-8 |x|, so |bf(q) bf(g) - q g| <= (2^-7 + 2^-16) |q g|, accumulation 17 * 2^-23:
// eps = 0.008 times sum |g_k| Q_k (Q_k: data-set maximum of |q_k|) rides with the slack, rounded up, so the accumulated value is an
// upper bound of the exact (dA + dB) / 4.  (Per-point absolute terms |q_k| eps |g_k| in twelve more slots are tighter — 3.77 instead
// of 3.87 % of the evaluations survive — but need a second operand block, 128 bytes per point, and lose: 1.59 vs 1.49 ms.)
// With (x - d)^2 >= x^2 / (1 + k) - d^2 / k, k = 2^-5:   r''^2 > (dS'' / 4)_up + delta''^2 (1 + k) / k  =>  c^2 r^2 >= (dS / 4) / (1 + k)  =>
// 4 r^2 / dS >= T  when  c^2 (1 + k) T <= 1, and sed_fl >= (1 - 5 * 2^-53) r_fl^2 (1 / da_fl + 1 / db_fl) >= (1 - 5 * 2^-53) 4 r_fl^2 /
// (da_fl + db_fl) (T carries the factor 1 + 1e-5 for the fp64 roundings and the one rounding of the test).  The slack
// rides in the constant slot of the dB chain.  A hypothesis whose magnitudes fall outside the scaled ranges, or with a NaN,
// gets an infinite slack: nothing is rejected and the exact tier decides everything.
#pragma once

#ifndef SFM_MATRIX_WIDE
#define SFM_MATRIX_WIDE 1    // large single-pair launches: waves of 64 hypotheses behind the heaviest groups (matrix_item's WIDE; 0: waves of 32 everywhere)
#endif
#ifndef SFM_MATRIX_OCC
#define SFM_MATRIX_OCC 4     // blocks per CU the matrix-pipe kernel is compiled for (32 KiB of queues each); the kernel with wide waves: 3 (163 VGPRs)
#endif
#ifndef SFM_MATRIX_POPS
#define SFM_MATRIX_POPS 4    // points a lane pops per round of the exact tier (2: the same at 50 000 x 100 000, 10-13 % slower at 20 000 x 40 000 and 50 000 x 20 000 where the final drains dominate; 6: a wave less per SIMD; 8: spills)
#endif
#ifndef SFM_MATRIX_ABLATE
#define SFM_MATRIX_ABLATE 0  // measurement builds only (WRONG results; tools/r04/ablate.sh): bit 0 no operand refills, bit 1 one matrix
#endif                       // instruction instead of three, bit 2 one sign test instead of sixteen, bit 3 no queue push, bit 4 no fp16 subnormal operands
#ifndef SFM_MATRIX_MASK_GROUP_SINGLE
#define SFM_MATRIX_MASK_GROUP_SINGLE 4   // pops of an exact-tier round that share one execution-mask region, launches of one pair (0: no masking; A/B builds)
#endif
#ifndef SFM_MATRIX_MASK_GROUP_WIDE
#define SFM_MATRIX_MASK_GROUP_WIDE 2     // ... launches of one pair with wide waves (-0.2 ... -0.9 % against 4 at 60 000 ... 250 000 hypotheses)
#endif
#ifndef SFM_MATRIX_MASK_GROUP_BATCH
#define SFM_MATRIX_MASK_GROUP_BATCH 2    // ... launches over a batch of pairs
#endif
#ifndef SFM_MATRIX_E_IN_REGISTERS
#define SFM_MATRIX_E_IN_REGISTERS 1   // E held in 18 VGPRs through the tier-1 loop (0: loaded where a burst of rounds starts — better while the
                                      // loop's registers were the constraint, 2.8 % slower at the bench size on the final kernel)
#endif
#ifndef SFM_MATRIX_STATS
#define SFM_MATRIX_STATS 0   // diagnostic build: rounds of the exact tier, points popped, push-loop iterations (sfm_debug_matrix_stats)
#endif

#include "sfm_matrix_tables.h"

namespace matrixscore {

#if SFM_MATRIX_STATS
__device__ unsigned long long g_matrix_stats[4];
#endif
#ifndef SFM_MATRIX_STAMPS
#define SFM_MATRIX_STAMPS 0   // diagnostic build (tools/r04/matrix_timeline.py): s_memrealtime stamps of every wave's phases
#endif
#if SFM_MATRIX_STAMPS
__device__ unsigned long long g_matrix_stamps[10 * 65536];   // begin, operands in, loop done, drained, samples fixed, handed off, hw id, wave, exact-tier lane slots | points popped, shader clocks
#define SFM_STAMP(k)                                                                     \
    do {                                                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
        stamp[k] = __builtin_amdgcn_s_memrealtime();                                     \
    } while (0)
#else
#define SFM_STAMP(k) do {} while (0)
#endif

#ifndef SFM_MATRIX_CAP
#define SFM_MATRIX_CAP 32
#endif
constexpr bool kWideWaves = SFM_MATRIX_WIDE != 0;   // the launcher may pick the kernel with wide waves (score_sed_matrix_kernel<.., .., true>)
constexpr int kWideOcc = 3;
#ifndef SFM_MATRIX_WIDE_MIN_HYPOTHESES
#define SFM_MATRIX_WIDE_MIN_HYPOTHESES 8192    // ... for one pair with at least this many hypotheses, in cost order (measured down to 200 000 x 10 000: -4 %)
#endif
#ifndef SFM_MATRIX_WIDE_MIN_HYPOTHESES_BATCH
#define SFM_MATRIX_WIDE_MIN_HYPOTHESES_BATCH (1 << 30)   // ... per pair of a batch (measurement builds: 1024)
#endif
#ifndef SFM_MATRIX_WIDE_FROM
#define SFM_MATRIX_WIDE_FROM 4096              // entries of the heaviest-first order that stay in waves of 32 (a multiple of 128: whole blocks)
#endif
constexpr int kWideFromMax = 4 * SFM_MATRIX_WIDE_FROM;   // the wide waves begin at the first class boundary in [WIDE_FROM, kWideFromMax] (none: at entry 0)
static_assert(SFM_MATRIX_WIDE_FROM % 128 == 0, "blocks of four waves of 32 hypotheses");
constexpr int kCap = SFM_MATRIX_CAP;   // entries per lane queue (a power of two: the queue is a ring); an entry is one step's survivors
constexpr int kHigh = kCap - 4;  // a step pushes at most one entry per lane: rounds start when a queue holds this many (checked once per group of steps) ...
constexpr int kLow = 8;          // ... and stop when every queue is down to this
constexpr int kPops = SFM_MATRIX_POPS;
#ifndef SFM_MATRIX_AHEAD
#define SFM_MATRIX_AHEAD 1
#endif
static_assert(SFM_MATRIX_AHEAD >= 1 && kHigh + 2 * (SFM_MATRIX_AHEAD + 1) - 1 <= kCap, "a group of kAhead + 1 steps (two entries each in a wide wave) must fit behind kHigh - 1 entries");
static_assert(SFM_MATRIX_AHEAD + 1 <= 4, "the point operand table is padded to a multiple of four steps");
constexpr int kAhead = SFM_MATRIX_AHEAD;   // steps of operand loads in flight behind the one being processed (register stages: kAhead + 1)
#ifndef SFM_MATRIX_ESTIMATE_AHEAD
#define SFM_MATRIX_ESTIMATE_AHEAD 1   // the same for the cost pre-pass (tier 1 alone: no exact tier competes for its registers); 1 or 3
#endif
static_assert(SFM_MATRIX_ESTIMATE_AHEAD == 1 || SFM_MATRIX_ESTIMATE_AHEAD == 3, "the recording pre-pass packs steps in pairs; the table is padded to four steps");
#ifndef SFM_MATRIX_BUFFER_LOADS
#define SFM_MATRIX_BUFFER_LOADS 1   // operand refills of the step loop as buffer loads (0: global loads with a 64-bit vector add per step)
#endif
#ifndef SFM_MATRIX_MASKED_SUMS
#define SFM_MATRIX_MASKED_SUMS 0    // 1: the exact tier's count and sums under the execution mask instead of three selects — measured: equal at
                                    // 50 000 x 100 000, 3-4 % slower at 20 000 x 40 000 and 50 000 x 20 000 (the branch in light waves' drains)
#endif
#ifndef SFM_MATRIX_REPLAY
#define SFM_MATRIX_REPLAY 1         // the scoring launch replays the tier-1 results of the cost pre-pass (MatrixPair::record); 0: computes them again
#endif
#ifndef SFM_MATRIX_ESTIMATE_STEPS
#define SFM_MATRIX_ESTIMATE_STEPS 128
#endif
constexpr int kEstimateSteps = SFM_MATRIX_ESTIMATE_STEPS;   // steps of 32 points the cost pre-pass scans at most (4096 points)
constexpr int64_t kMaxPoints = sfmws::kMatrixMaxPoints;
constexpr int kMaxRangeSteps = kWideWaves ? 32768 : 65536;   // a queue entry keeps the step RELATIVE TO ITS RANGE (wide waves: 2 x step + half) in 16 bits: at most 2^15 steps (1 M points) per range
static_assert(kMaxPoints <= (int64_t)sfmws::kSplitMaxUnits * kMaxRangeSteps * kTile, "sixteen ranges of 2^16 steps cover the largest pair");

// ... and an eighth of a smaller point set, but no fewer than 1024 points: the pre-pass is tier 1 over that share of the points
__host__ __device__ inline int estimate_steps(int64_t n) {
    const int64_t eighth = steps_of(n) / 8;
    return (int)(eighth < 32 ? 32 : eighth > kEstimateSteps ? kEstimateSteps : eighth);
}

__global__ __launch_bounds__(256) void matrix_prepare_kernel(const Corr* __restrict__ corr, int n, double c,
                                                            const unsigned char* __restrict__ ws, uint4* __restrict__ table) {
    const int64_t pair = blockIdx.y;   // (`ws`: the pair's maxima as score_prepare_kernel left them); four steps per block
    const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x / kWave);
    if (t >= (int)table_steps(n)) return;
    prepare_step(corr + pair * (int64_t)n, n, c, reinterpret_cast<const uint32_t*>(ws + 16 * pair),
                 table + pair * table_steps(n) * kBlocks * 64, t, (int)(threadIdx.x & (kWave - 1)));
}

__global__ __launch_bounds__(256) void matrix_hypothesis_kernel(const unsigned char* __restrict__ ws, const double* __restrict__ E,
                                                                int h_count, double a_scale, uint4* __restrict__ hyp_table,
                                                                float* __restrict__ bound_out, const Corr* __restrict__ pts, int n,
                                                                const int32_t* __restrict__ S, double thr,
                                                                unsigned char* __restrict__ fix) {
    const int64_t pair = blockIdx.y;   // (`ws`: the pair's maxima)
    prepare_hypothesis(reinterpret_cast<const uint32_t*>(ws + 16 * pair), E + pair * (int64_t)h_count * 9, h_count, a_scale,
                       hyp_table + pair * (int64_t)h_count * 2 * kBlocks,
                       bound_out != nullptr ? bound_out + pair * (int64_t)h_count * 8 : nullptr, pts + pair * (int64_t)n,
                       S != nullptr ? S + pair * (int64_t)h_count * 8 : nullptr, thr,
                       fix != nullptr ? fix + pair * sfmws::matrix_fix_bytes(h_count) : nullptr,
                       (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- one pair: the preparation in TWO launches instead of five (a pass is a chain of dependent launches of 4-15 us each) ----
// Launch 1 (matrix_setup_kernel): per-block partial maxima of the scaled coordinates — plain stores, folded by the blocks of
// launch 2, so no atomics and nothing to zero first — and the zeroing the later launches need: class counters and their
// cursors, the cost estimates (`cnt`: the pre-pass adds its ranges' counts), the state words of the pass's selection launch.
constexpr int kSetupBlocks = 256;   // at most this many partial maxima per pair (one per block; blocks walk the points grid-stride)
// (grid: x = blocks of a pair, y = pair: every array with a leading pair dimension; `state` belongs to a single pair)
__global__ __launch_bounds__(256) void matrix_setup_kernel(const Corr* __restrict__ corr, int n, double a_scale,
                                                           float4* __restrict__ partial, int32_t* __restrict__ buckets,
                                                           int32_t* __restrict__ cnt, int h_count, unsigned* __restrict__ state) {
    const int64_t pair = blockIdx.y;
    corr += pair * (int64_t)n;
    partial += pair * (int64_t)gridDim.x;
    buckets += pair * (int64_t)sfmws::kBuckets;
    if (cnt != nullptr) cnt += pair * (int64_t)h_count;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    const int stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = first; i < n; i += stride) {
        const float4 q = sfmws::to_filter_point(corr[i], a_scale);   // NaN coordinates: fmaxf ignores them (such points fail the filter's test)
        m0 = fmaxf(m0, fabsf(q.x));
        m1 = fmaxf(m1, fabsf(q.y));
        m2 = fmaxf(m2, fabsf(q.z));
        m3 = fmaxf(m3, fabsf(q.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
        m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
        m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
    }
    __shared__ float part[4][4];
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        part[w][0] = m0; part[w][1] = m1; part[w][2] = m2; part[w][3] = m3;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = make_float4(fmaxf(fmaxf(part[0][0], part[1][0]), fmaxf(part[2][0], part[3][0])),
                                          fmaxf(fmaxf(part[0][1], part[1][1]), fmaxf(part[2][1], part[3][1])),
                                          fmaxf(fmaxf(part[0][2], part[1][2]), fmaxf(part[2][2], part[3][2])),
                                          fmaxf(fmaxf(part[0][3], part[1][3]), fmaxf(part[2][3], part[3][3])));
    for (int i = first; i < sfmws::kBuckets; i += stride) buckets[i] = 0;
    if (cnt != nullptr)
        for (int i = first; i < h_count; i += stride) cnt[i] = 0;
    if (state != nullptr && pair == 0 && first < 16) state[first] = 0u;
}

// Launch 2 (matrix_tables_kernel): every block folds its pair's partial maxima (<= 256 x 16 bytes) and then writes its share of
// the point operand table (four steps per block) or of the hypothesis operand table and the sample corrections.  (In a fused
// pass the lanes of the fit launch write the hypotheses' half themselves — and, for one pair, blocks of that launch the point
// table: this kernel then runs for the point tables of a batch only, or not at all.)
__global__ __launch_bounds__(256) void matrix_tables_kernel(const Corr* __restrict__ corr, int n, double a_scale,
                                                            const float4* __restrict__ partial, int partials, uint4* __restrict__ table,
                                                            int step_blocks, const double* __restrict__ E, int h_count,
                                                            uint4* __restrict__ hyp_table, const int32_t* __restrict__ S, double thr,
                                                            unsigned char* __restrict__ fix) {
    __shared__ float part[4][4];
    __shared__ uint32_t maxima[4];
    const int64_t pair = blockIdx.y;
    corr += pair * (int64_t)n;
    partial += pair * (int64_t)partials;
    {
        const float4 m = (int)threadIdx.x < partials ? partial[threadIdx.x] : make_float4(0.f, 0.f, 0.f, 0.f);
        float m0 = m.x, m1 = m.y, m2 = m.z, m3 = m.w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
            m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
            m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
        }
        const int w = threadIdx.x / kWave;
        if ((threadIdx.x & (kWave - 1)) == 0) {
            part[w][0] = m0; part[w][1] = m1; part[w][2] = m2; part[w][3] = m3;
        }
        __syncthreads();
        if (threadIdx.x < 4)
            maxima[threadIdx.x] = __float_as_uint(fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]),
                                                        fmaxf(part[2][threadIdx.x], part[3][threadIdx.x])));
        __syncthreads();
    }
    if ((int)blockIdx.x < step_blocks) {
        const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x / kWave);
        if (t < (int)table_steps(n))
            prepare_step(corr, n, a_scale, maxima, table + pair * table_steps(n) * kBlocks * 64, t, (int)(threadIdx.x & (kWave - 1)));
        return;
    }
    prepare_hypothesis(maxima, E + pair * (int64_t)h_count * 9, h_count, a_scale, hyp_table + pair * (int64_t)h_count * 2 * kBlocks, nullptr,
                       corr, S + pair * (int64_t)h_count * 8, thr, fix + pair * sfmws::matrix_fix_bytes(h_count),
                       (int64_t)((int)blockIdx.x - step_blocks) * 256 + threadIdx.x);
}

// Diagnostic (sfm_debug_matrix_filter; tests/test_gpu_parity.py measures the bound's margin with it): tier 1 of ONE 32 x 32 tile
// per wave — the three MFMAs of the scoring kernel on the same operand tables, in the same order — with the raw accumulators
// written out: r_out / d_out [h_count][32 steps] = r'' and the upper bound of dS'' / 4 (slack included) of every (hypothesis, point).
__global__ __launch_bounds__(64) void matrix_filter_dump_kernel(const uint4* __restrict__ hyp_table, const uint4* __restrict__ table,
                                                                int steps, int h_count, float* __restrict__ r_out,
                                                                float* __restrict__ d_out) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int col = lane & 31, half = lane >> 5;
    const int h = min((int)blockIdx.y * kHyps + col, h_count - 1);
    const uint4* __restrict__ operands = hyp_table + ((int64_t)h * 2 + half) * kBlocks;
    const f16x8 B0 = __builtin_bit_cast(f16x8, operands[0]);
    const f16x8 B1 = __builtin_bit_cast(f16x8, operands[1]);
    const bf16x8 B2 = __builtin_bit_cast(bf16x8, operands[2]);
    const uint4* __restrict__ src = table + (size_t)t * kBlocks * 64 + lane;
    const uint4 A0 = src[0], A1 = src[64], A2 = src[128];
    float16v r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d = r;
    r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A0), B0, r, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2), B2, d, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A1), B1, r, 0, 0, 0);
    if ((int)blockIdx.y * kHyps + col >= h_count) return;
    const size_t point0 = (size_t)h * ((size_t)steps * kTile) + (size_t)t * kTile + 16 * half;   // (register j: point 16 half + j)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        r_out[point0 + j] = r[j];
        d_out[point0 + j] = d[j];
    }
}

// One work item: 32 hypotheses (entries [32 wave, 32 wave + 32) of the processing order) over the steps of one range of the
// points, done by one wave.
// ESTIMATE: the cost pre-pass of this kernel — tier 1 alone over the first `units` x `steps_per_unit` steps (a single pair: in
// `units` ranges by different waves, a pre-pass of 3125 waves over 128 steps each is latency-bound: 59 us, 4 x 32 steps: see
// profiles/r03), cnt[h] = the survivors per 1024 points in sixteenths (what the counting sort's classes are defined on).  With 1024 points the estimate is 52 +- 7 for a typical hypothesis
// and the 32 hypotheses a wave runs in lock step differ by that noise (lane utilisation 0.77); tier 1 on the matrix pipe makes
// 4096 points as cheap as 1024 were on the VALU.
struct MatrixPair {   // the arrays of one image pair
    const Corr* __restrict__ pts;
    const uint4* __restrict__ hyp_table;
    const uint4* __restrict__ table;
    const double* __restrict__ E;
    const int32_t* __restrict__ order;
    int32_t* __restrict__ cnt;
    double* __restrict__ s1;
    double* __restrict__ s2;
    unsigned char* __restrict__ split;
    const unsigned char* __restrict__ fix;   // sample corrections of matrix_hypothesis_kernel: [h] int32 | [h] double | [h] double
    // Tier-1 results the cost pre-pass leaves for the scoring launch (nullptr: none): the reject word of every (hypothesis, lane
    // half, range, step) it scanned — 32 bytes per (hypothesis, half, range), laid out [range][8 steps][hypothesis][half] x 16 bytes, per pair —
    // when the pre-pass scans the FIRST kReplaySteps steps of each of the scoring launch's ranges instead of the first steps of
    // the points: the scoring wave of a range then replays those words (five vector instructions a step) instead of computing
    // them again (three matrix and 36 vector instructions): the pre-pass' 8 % of tier 1 is no longer done twice.
    uint16_t* __restrict__ record;
    int range_stride;   // steps between the starts of the pre-pass' ranges (= the scoring launch's steps per range) when recording
};
constexpr int kReplaySteps = sfmws::kMatrixReplaySteps;   // 16 steps of every range: 8 ranges = the 128 steps (4096 points) of the bench workload's pre-pass
static_assert(kReplaySteps % 8 == 0 && kReplaySteps <= kCap - 4 && 2 * kReplaySteps <= kCap, "whole 16-byte stores; the replayed entries fit the empty ring");
__host__ __device__ inline int64_t record_bytes(int64_t h_count) { return sfmws::matrix_record_bytes(h_count); }

// One wave-uniform ticket from an agent-scope counter, in straight-line assembly with the exec mask set by hand: written as
// `if (lane == 0) t = atomicAdd(..); t = readfirstlane(t)` inside a loop the compiler may thread the inactive lanes past the
// atomic into the next iteration, where readfirstlane then reads one of THEM (round 3, profiles/r03/README.md).
SFM_DEVICE int take_ticket(int32_t* counter) {
    int ticket, returned;
    unsigned long long saved;
    const int one = 1;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %[returned], %[address], %[one], off sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 %[ticket], %[returned]\n\t"
        "s_mov_b64 exec, %[saved]"
        : [ticket] "=s"(ticket), [returned] "=&v"(returned), [saved] "=&s"(saved)
        : [address] "v"(counter), [one] "v"(one)
        : "memory");
    return ticket;
}

template <bool ESTIMATE, int MASK_GROUP, bool WIDE>
SFM_DEVICE void matrix_item(const MatrixPair& a, int n, int h_count, double thr, int units, int steps_per_unit, int h0, int h_limit,
                            int unit, uint32_t* const my_queue, int lane, unsigned item_id) {
    const Corr* __restrict__ pts = a.pts;
    const uint4* __restrict__ hyp_table = a.hyp_table;
    const uint4* __restrict__ table = a.table;
    const double* __restrict__ E = a.E;
    const int32_t* __restrict__ order = a.order;
    int32_t* __restrict__ cnt = a.cnt;
    double* __restrict__ s1 = a.s1;
    double* __restrict__ s2 = a.s2;
#if SFM_MATRIX_STAMPS
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    const unsigned long long clock_begin = __builtin_amdgcn_s_memtime();
#endif
    if (h0 >= h_limit) return;   // (no block-level synchronisation in this kernel); h_limit: the end of this kind of wave's entries of the order
    const int col = lane & 31, half = lane >> 5;
    // A lane OWNS one hypothesis — its queue, its exact entries, its sums.  Waves of 32: entry h0 + col of the order, shared by the
    // two lanes of a column (16 of a step's 32 points each).  Wide waves (64): entry h0 + lane — the lower lanes the columns of the
    // first operand group, the upper lanes those of the second —, all 32 points of a step; for tier 1 a lane also supplies its
    // half of the operand rows of BOTH groups' column `col`.
    const bool valid = h0 + (WIDE ? lane : col) < h_limit;
    const int slot = min(h0 + (WIDE ? lane : col), h_limit - 1);
    const int h = order != nullptr ? order[slot] : slot;
    int h_first = h, h_second = h;   // wide waves: the hypotheses of column `col` in the two operand groups
    if (WIDE) {
        const int slot_first = min(h0 + col, h_limit - 1), slot_second = min(h0 + kHyps + col, h_limit - 1);
        h_first = order != nullptr ? order[slot_first] : slot_first;
        h_second = order != nullptr ? order[slot_second] : slot_second;
    }

    // ---- this lane's hypothesis: exact entries for the exact tier; the B operands of tier 1 come from the table
    // matrix_hypothesis_kernel wrote once per launch (a hypothesis is scored by up to 16 range waves and the cost pre-pass)
    // (the exact entries are loaded where the exact tier starts — in front of a burst of rounds, and of the final drain — not
    // kept in 18 VGPRs through the tier-1 loop, which never reads them: the loop's occupancy is what they would cost)
    double e[9];
    auto load_e = [&]() __attribute__((always_inline)) {
        int64_t first = (int64_t)h * 9;
        asm volatile("" : "+v"(first));   // (an address the compiler cannot prove loop-invariant: the loads stay where they are written)
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = E[first + j];
    };
#if !SFM_MATRIX_E_IN_REGISTERS
    if (ESTIMATE)
#endif
        load_e();
    const uint4* __restrict__ operands = hyp_table + ((int64_t)h_first * 2 + half) * kBlocks;
    const f16x8 B0 = __builtin_bit_cast(f16x8, operands[0]);
    const f16x8 B1 = __builtin_bit_cast(f16x8, operands[1]);
    const bf16x8 B2 = __builtin_bit_cast(bf16x8, operands[2]);
    const uint4* __restrict__ operands_second = hyp_table + ((int64_t)h_second * 2 + half) * kBlocks;   // (wide waves; else the same rows, unused)
    const f16x8 C0 = __builtin_bit_cast(f16x8, operands_second[0]);
    const f16x8 C1 = __builtin_bit_cast(f16x8, operands_second[1]);
    const bf16x8 C2 = __builtin_bit_cast(bf16x8, operands_second[2]);
    SFM_STAMP(1);

    // The lane's queue is a ring of entries (step << 16 | the step's 16 survivor bits), pushed at `tail`, popped at `head`; the
    // entry being consumed lives in registers (`cur` = its remaining bits, `cur_base` = index of its row 0).  Points of a
    // hypothesis are therefore scored — and their errors summed — in the order of the points, whatever the other lanes of the
    // wave do (when rounds happen depends on the fullest queue of the wave, and which hypotheses share a wave on the arrival
    // order of the counting sort's atomics: a stack would make the summation order, i.e. the last bits of the sums, vary from
    // run to run).  One entry per step and lane instead of one per survivor: the push is a single predicated store.
    int c = 0;
    // (head and tail count BYTES of the lane's column, 256 per entry: the slot's address is one and-or away)
    unsigned head = 0, tail = 0, cur = 0;
    constexpr unsigned kSlotBytes = kWave * 4, kRingMask = (kCap - 1) * kSlotBytes;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t queue_base = (uint32_t)(uintptr_t)(lds_u32*)my_queue;   // LDS byte address of slot 0: a wave's ring is 8 KiB-aligned
    auto ring_slot = [&](unsigned counter) __attribute__((always_inline)) {
        return (lds_u32*)(uintptr_t)(queue_base | (counter & kRingMask));
    };
    unsigned cur_off = 0;   // byte offset of the entry's point "-16" in the correspondences (bit 15 - j <-> leading zeros 16 + j)
    double a1 = 0.0, a2 = 0.0;
    // (my_queue: this lane's column of the wave's queue block in LDS — slot k at my_queue[k * kWave])
#if SFM_MATRIX_STATS || SFM_MATRIX_STAMPS
    unsigned stat_rounds = 0, stat_pops = 0, stat_push_iterations = 0;
#endif

    const int steps_total = (int)steps_of(n);
    const bool recording = ESTIMATE && a.record != nullptr;    // the pre-pass of a launch that replays (see MatrixPair::record)
    const bool replaying = !ESTIMATE && a.record != nullptr;
    // (steps_per_unit is a multiple of kStages: ranges start on group boundaries)
    const int step_begin = units > 1 ? unit * (recording ? a.range_stride : steps_per_unit) : 0;
    // this lane's words of this range: 16-byte chunks of eight steps, [range][chunk][hypothesis][half] — the pre-pass, whose lanes
    // are consecutive hypotheses, stores 2 KiB per wave and chunk in one piece; the scoring wave gathers its chunks by hypothesis
    // either way.  (The first layout kept a lane's chunks together, 512 bytes from the next lane's: the pre-pass took 73 us
    // instead of 55 for its 50 000 store instructions of 64 separate lines each.)
    const int64_t record_stride = sfmws::split_padded(h_count) * 2;   // chunks between a lane's consecutive chunks
    uint4* const my_record = a.record == nullptr ? nullptr
        : reinterpret_cast<uint4*>(a.record) + (int64_t)unit * (kReplaySteps / 8) * record_stride + (int64_t)h * 2 + (WIDE ? 0 : half);   // (wide waves: both halves, [0] and [1])
    const int step_end = units > 1 || ESTIMATE ? min(step_begin + steps_per_unit, steps_total) : steps_total;
    const int last_loadable = (int)table_steps(n) - 1;

    // one round of the exact tier: every lane takes up to kPops of its queued points (their gathers in flight together) and
    // scores them, in queue order, under its own hypothesis
    const sfm::SedGate gate = sfm::sed_gate(thr);
    // bytes: point (step_begin x 32 + 16 half - 16) of the correspondences (kMaxPoints x 32 bytes = 128 MB: 32 bits are plenty)
    const unsigned lane_off = (unsigned)((step_begin * kTile + (WIDE ? 0 : 16 * half) - 16) * (int)sizeof(Corr));
    auto round = [&]() __attribute__((always_inline)) {
        Corr p[kPops];
        bool active[kPops];
#pragma unroll
        for (int k = 0; k < kPops; ++k) {
            if ((cur == 0u) & (head != tail)) {   // (one masked region, not two nested ones)
                const unsigned entry = *ring_slot(head);
                head += kSlotBytes;
                cur = entry & 0xffffu;
                // (the entry keeps the step relative to its range: 32 points x 32 bytes; wide waves: 2 x step + half: 16 points x 32 bytes)
                cur_off = lane_off + ((entry >> 16) << (WIDE ? 9 : 10));
            }
            active[k] = cur != 0u;
            const unsigned lz = (unsigned)__builtin_clz(cur | 1u);   // 16..31 for a live entry: bit 15 - j is register j = point j of the lane's sixteen
            cur &= ~(0x80000000u >> lz);
            // a 32-bit byte offset from the (uniform) base: one shift-and-add, and the load takes base + offset as it is
            const unsigned off = active[k] ? cur_off + (lz << 5) : 0u;
            p[k] = *reinterpret_cast<const Corr*>(reinterpret_cast<const unsigned char*>(pts) + off);
        }
#if SFM_MATRIX_STATS || SFM_MATRIX_STAMPS
        stat_rounds += kPops;
#pragma unroll
        for (int k = 0; k < kPops; ++k) stat_pops += active[k] ? 1u : 0u;
#endif
        // `from` .. `to`: evaluate the pops of this stretch in straight-line code (the scheduler interleaves their dependency chains)
        auto evaluate = [&](int from, int to) __attribute__((always_inline)) {
#pragma unroll
            for (int k = from; k < to; ++k) {
                double sed;
                const bool ok = sfm::sed_inlier(e, p[k].xa, p[k].ya, p[k].xb, p[k].yb, gate, sed) && active[k];
#if SFM_MATRIX_MASKED_SUMS
                if (ok) {   // under the execution mask (three instructions for the inlier lanes) instead of three selects + three instructions for all
                    asm volatile("" : "+v"(c));
                    c += 1;
                    a1 += sed;
                    a2 = fma(sed, sed, a2);
                }
#else
                c += ok ? 1 : 0;
                const double kept = ok ? sed : 0.0;   // masked once; its square is the masked square
                a1 += kept;
                a2 = fma(kept, kept, a2);
#endif
            }
        };
        if (MASK_GROUP > 0) {
            // Under the execution mask of the lanes that popped a point in this stretch of MASK_GROUP pops (a lane pops in order, so
            // active[k] implies active[k - 1]: the first pop of a stretch decides).  The same instructions are issued, but the idle
            // lanes — a fifth of the slots at the bench size — no longer compute on a dummy point: these launches run at the card's
            // power cap, and the energy comes back as clock.  One region per pop saves the most energy and lets the scheduler
            // interleave nothing (-1.8 % at 50 000 x 100 000, +3 ... +6 % on launches of few hypotheses over many points, whose
            // lanes are rarely idle); one region per round of four keeps the four dependency chains interleaved and never lost
            // (-1.0 ... -1.9 % on every shape measured); batches of small pairs, whose lanes often hold one or two points, do best
            // with pairs of pops (256 x 10 000 x 2 000: -2.4 % against -1.5 %).  profiles/r05/README.md item 10.
#pragma unroll
            for (int g = 0; g < kPops; g += (MASK_GROUP > 0 ? MASK_GROUP : kPops))
                if (active[g]) evaluate(g, g + MASK_GROUP < kPops ? g + MASK_GROUP : kPops);
        } else {
            evaluate(0, kPops);
        }
    };

    unsigned survivors = 0;   // ESTIMATE
    unsigned rec[4] = {0u, 0u, 0u, 0u}, rec_low = 0u;   // ESTIMATE, recording: the reject words of the last eight steps
    unsigned rec_upper[4] = {0u, 0u, 0u, 0u}, rec_upper_low = 0u;   // ... wide waves: of the lane's other half
    static_assert((kAhead + 1) % 2 == 0 || !SFM_MATRIX_REPLAY, "the recording pre-pass packs the steps of a loop group two to a dword");
    const int first_step = step_begin + (replaying ? kReplaySteps : 0);   // (the host replays only when every range has that many steps)
    if (step_begin < step_end) {
        const uint4* __restrict__ src = table + lane;
        const unsigned lane_bytes = (unsigned)lane * 16u;
        // the point operand table as a buffer (gfx9 descriptor: raw, 32-bit data format; range = the table, pad steps included)
        const __amdgpu_buffer_rsrc_t table_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint4*>(table), (short)0, (int)(table_steps(n) * (int64_t)(kBlocks * 64 * 16)), 0x00020000);
        // Operand loads run kStages steps ahead of the step being processed.  The stages ROTATE — the step loop is unrolled
        // kStages times and stage s is refilled (with the operands of step t + kStages) right behind the three matrix
        // instructions that consumed it — so no register is ever copied: rounds 3's form shifted the stages down once per
        // step, and a copy out of a register a load is still writing makes the wave wait for ALL its loads (s_waitcnt
        // vmcnt(0) at the top of every step), i.e. the loop ran one memory latency per step whatever the depth — tier 1 of a
        // light wave took ~450 cycles per step and SIMD against ~220 of issue for that reason.
        constexpr int kStages = (ESTIMATE ? SFM_MATRIX_ESTIMATE_AHEAD : kAhead) + 1;
        uint4 A[kStages][kBlocks];
        // (the first fill in stage order, oldest first, like every refill: the wait in front of stage 0 at the loop's head is ONE
        // instruction for the entry and the back edge — entered with stage 0's loads as the youngest it would be vmcnt(0) forever)
#pragma unroll
        for (int a = 0; a < kStages; ++a) {
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) A[a][b] = src[((size_t)min(first_step + a, last_loadable) * kBlocks + b) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (replaying) {
            // the first kReplaySteps steps of the range: the pre-pass' reject words instead of tier 1 (at most 16 pushes into an
            // empty ring of kCap = 32: no round can be due)
            unsigned words[kReplaySteps / 2], words_upper[kReplaySteps / 2];   // (wide waves: the lane's two halves)
#pragma unroll
            for (int q = 0; q < kReplaySteps / 8; ++q) {
                const uint4 w = my_record[q * record_stride];
                words[4 * q] = w.x, words[4 * q + 1] = w.y, words[4 * q + 2] = w.z, words[4 * q + 3] = w.w;
                if (WIDE) {
                    const uint4 u = my_record[q * record_stride + 1];
                    words_upper[4 * q] = u.x, words_upper[4 * q + 1] = u.y, words_upper[4 * q + 2] = u.z, words_upper[4 * q + 3] = u.w;
                }
            }
#pragma unroll
            for (int s = 0; s < kReplaySteps; ++s) {
                const unsigned rejected = (s & 1) ? words[s >> 1] >> 16 : words[s >> 1] & 0xffffu;
                const unsigned field = WIDE ? 2u * (unsigned)s : (unsigned)s;
                if (rejected != 0xffffu) {
                    *ring_slot(tail) = ~(rejected ^ ((field << 16) ^ 0xffff0000u));
                    tail += kSlotBytes;
                }
                if (WIDE) {
                    const unsigned upper = (s & 1) ? words_upper[s >> 1] >> 16 : words_upper[s >> 1] & 0xffffu;
                    if (upper != 0xffffu) {
                        *ring_slot(tail) = ~(upper ^ (((field + 1u) << 16) ^ 0xffff0000u));
                        tail += kSlotBytes;
                    }
                }
            }
            // (wide waves push up to 2 x kReplaySteps = kCap entries: the ring may be full now, and the step loop below looks
            // at the queues only BEHIND a group of steps)
            if (WIDE && __builtin_amdgcn_ballot_w64((int)(tail - head) >= kHigh * (int)kSlotBytes) != 0ull) {
                __builtin_amdgcn_wave_barrier();
                do round(); while (__builtin_amdgcn_ballot_w64((int)(tail - head) > kLow * (int)kSlotBytes) != 0ull);
            }
        }
        int t0 = first_step;
        while (t0 < step_end) {
         bool queue_full = false;
         do {   // the hot loop: groups of kStages steps until a queue is full (or the range ends)
#pragma unroll
          for (int stage = 0; stage < kStages; ++stage) {
            const int t = t0 + stage;   // (past the end of the points — at most kStages - 1 steps of the last range — these are the table's pad steps, which keep nothing)
            float16v r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d = r;
            r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[stage][0]), B0, r, 0, 0, 0);
#if !(SFM_MATRIX_ABLATE & 2)
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[stage][2]), B2, d, 0, 0, 0);
            r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[stage][1]), B1, r, 0, 0, 0);
#endif
            // rejected bits, register 0 ending up in bit 15: the sign of dB - r^2 (one rounding: the sign is exact, and zero
            // — equality — keeps the point) shifted in with an alignbit.  A NaN with its sign set counts as rejected, which is
            // what the exact tier would decide for it (sed = NaN is not <= thr).
            unsigned rejected = 0;
#pragma unroll
            for (int j = 0; j < ((SFM_MATRIX_ABLATE & 4) ? 1 : 16); ++j)
                rejected = __builtin_amdgcn_alignbit(rejected, __float_as_uint(__builtin_fmaf(-r[j], r[j], d[j])), 31);
            unsigned rejected_upper = 0xffffu;   // wide waves: the other half's word of this lane's hypothesis (below)
            if (WIDE) {
                // the same step against the operand rows of columns 32 .. 63, then the lane halves exchange words: v_permlane32_swap
                // swaps vdst[32..63] with src0[0..31] — lower lanes keep their first-group word (points 0 .. 15 of hypothesis `col`)
                // and receive the upper lanes' (points 16 .. 31); upper lanes receive the lower lanes' second-group word and keep theirs
                float16v r2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d2 = r2;
                r2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[stage][0]), C0, r2, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[stage][2]), C2, d2, 0, 0, 0);
                r2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[stage][1]), C1, r2, 0, 0, 0);
                unsigned second = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    second = __builtin_amdgcn_alignbit(second, __float_as_uint(__builtin_fmaf(-r2[j], r2[j], d2[j])), 31);
                const auto swapped = __builtin_amdgcn_permlane32_swap(rejected, second, false, false);
                rejected = swapped[0];         // half 0 of this lane's hypothesis
                rejected_upper = swapped[1];   // half 1
            }
            // (sixteen shifts: the upper half of `rejected` is zero.)  keep = ~rejected & 0xffff — bit 15 - j: register j = point
            // 16 half + j of this step — is never materialised on the scoring path: "any survivor" is rejected != 0xffff and the
            // queue entry (step << 16) | keep is one exclusive-nor of `rejected` with a wave-uniform word
#define SFM_KEEP_OF(REJECTED) (~(REJECTED) & 0xffffu)
            {
                // refill this stage.  The loads must stay BEHIND the three matrix instructions that read the stage: hoisted
                // above the second r' instruction (where instruction selection likes to put them) the load of block 1 needs a
                // copy of the old block 1, and a copy waits for the loads.  The empty asm makes the address depend on `rejected`.
                // (the OFFSET goes through the asm, not the pointer: a pointer coming out of an asm has lost its address space and
                // the loads become flat_load, which the compiler can only wait for with vmcnt(0))
                // (a buffer load: descriptor + scalar step offset + the lane's constant 32-bit offset + immediate are the load's own
                // addressing mode — no vector instruction computes an address in this loop; the scalar offset goes through the asm)
                unsigned step_bytes = (unsigned)min(t + kStages, last_loadable) * (unsigned)(kBlocks * 64 * 16);   // (wave-uniform: a scalar register)
                asm volatile("" : "+s"(step_bytes), "+v"(rejected), "+v"(rejected_upper));
#if !(SFM_MATRIX_ABLATE & 1)
#if SFM_MATRIX_BUFFER_LOADS
#pragma unroll
                for (int b = 0; b < kBlocks; ++b)
                    A[stage][b] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(table_rsrc, (int)lane_bytes + b * 1024, (int)step_bytes, 0));
#else
                const uint4* nxt = src + step_bytes / 16;
#pragma unroll
                for (int b = 0; b < kBlocks; ++b) A[stage][b] = nxt[b * 64];
#endif
#endif
            }
            if (ESTIMATE) {
                survivors += (unsigned)__builtin_popcount(SFM_KEEP_OF(rejected));
                if (WIDE) survivors += (unsigned)__builtin_popcount(SFM_KEEP_OF(rejected_upper));
                if (recording) {   // two steps make a dword, four dwords a 16-byte store (below): not sixteen scattered 2-byte stores a wave
                    if (stage & 1) {
                        rec[0] = rec[1];
                        rec[1] = rec[2];
                        rec[2] = rec[3];
                        rec[3] = rec_low | (rejected << 16);
                        if (WIDE) {
                            rec_upper[0] = rec_upper[1];
                            rec_upper[1] = rec_upper[2];
                            rec_upper[2] = rec_upper[3];
                            rec_upper[3] = rec_upper_low | (rejected_upper << 16);
                        }
                    } else {
                        rec_low = rejected;
                        rec_upper_low = rejected_upper;
                    }
                }
            } else {
#if SFM_MATRIX_ABLATE & 8
                survivors += (unsigned)__builtin_popcount(SFM_KEEP_OF(rejected));
#else
                const unsigned field = WIDE ? 2u * (unsigned)(t - step_begin) : (unsigned)(t - step_begin);   // (wide waves: 2 x step + half)
                if (rejected != 0xffffu) {   // push: one entry with this step's survivors
                    // (step << 16) | keep  ==  ~(rejected ^ k),  k = (step << 16) ^ 0xffff0000 (wave-uniform): one v_xnor
                    *ring_slot(tail) = ~(rejected ^ ((field << 16) ^ 0xffff0000u));
                    tail += kSlotBytes;
                }
                if (WIDE && rejected_upper != 0xffffu) {
                    *ring_slot(tail) = ~(rejected_upper ^ (((field + 1u) << 16) ^ 0xffff0000u));
                    tail += kSlotBytes;
                }
#endif
            }
          }
          t0 += kStages;
          if (recording && ((t0 - step_begin) & 7) == 0)   // (wave-uniform) the last eight steps' words: [t0 - 8, t0)
          {
              my_record[((t0 - step_begin) / 8 - 1) * record_stride] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
              if (WIDE) my_record[((t0 - step_begin) / 8 - 1) * record_stride + 1] = make_uint4(rec_upper[0], rec_upper[1], rec_upper[2], rec_upper[3]);
          }
          queue_full = !ESTIMATE && __builtin_amdgcn_ballot_w64((int)(tail - head) >= kHigh * (int)kSlotBytes) != 0ull;
         } while (t0 < step_end && !queue_full);
         // Rounds of the exact tier are looked at once per group of kStages steps, not between its stages, and run OUTSIDE the
         // hot loop: that loop is then straight-line code whose operand loads the compiler can count (s_waitcnt
         // vmcnt(3 (kStages - 1)) in front of a stage: the younger stages stay in flight) — with the gathers of a round
         // between two stages it falls back to vmcnt(0) at the top of every group.  A group pushes at most kStages entries per
         // lane: kHigh + kStages - 1 <= kCap.  (When rounds happen does not change any sum: a lane's queue is first-in first-out.)
         if (queue_full) {
#if !SFM_MATRIX_E_IN_REGISTERS
             load_e();
#endif
             __builtin_amdgcn_wave_barrier();
             do round(); while (__builtin_amdgcn_ballot_w64((int)(tail - head) > kLow * (int)kSlotBytes) != 0ull);
         }
        }
    }
    if (ESTIMATE) {   // survivors per 1024 points of this hypothesis (both lanes) in sixteenths, at least 1 when there was any
        const unsigned both = WIDE ? survivors : survivors + (unsigned)__shfl_xor((int)survivors, 32, 64);
        const unsigned scanned = (unsigned)min(units * steps_per_unit * kTile, n);   // by all ranges of the pre-pass together
        unsigned sixteenths = (unsigned)(((unsigned long long)both * 16384ull) / (scanned > 0u ? scanned : 1u));
        sixteenths = both > 0u && sixteenths == 0u ? 1u : sixteenths;
        if ((WIDE || half == 0) && valid) {
            // integer sums: any order (the launcher zeroed cnt).  (Plain stores instead — a timing experiment — change nothing: 70 vs 73 us.)
            if (units > 1) atomicAdd(cnt + h, (int32_t)sixteenths);
            else cnt[h] = (int32_t)sixteenths;
        }
        return;
    }
    SFM_STAMP(2);
#if !SFM_MATRIX_E_IN_REGISTERS
    load_e();
#endif
    __builtin_amdgcn_wave_barrier();
    while (__builtin_amdgcn_ballot_w64(tail != head || cur != 0u) != 0ull) round();
    SFM_STAMP(3);
#if SFM_MATRIX_ABLATE & 8
    c += (int)survivors;
#endif
#if SFM_MATRIX_STATS
    if (lane == 0) {
        atomicAdd(&g_matrix_stats[0], (unsigned long long)stat_rounds);
        atomicAdd(&g_matrix_stats[2], (unsigned long long)stat_push_iterations);
    }
    atomicAdd(&g_matrix_stats[1], (unsigned long long)stat_pops);
#endif

    // (The eight sample points — never counted, always summed: ransac.py:70-79 — were scanned like any other point; their
    // correction is a constant of the hypothesis, computed by matrix_hypothesis_kernel and added where the totals are written.
    // Until round 4 the wave of range 0 re-scored them here: eight dependent gather + fp64 evaluations, 20 us in a kernel whose
    // throughput is slots / wave lifetime, and — consecutive blocks being the ranges of one group, dealt round-robin over the
    // XCDs — all of them on XCD 0, which finished a quarter of the launch after the other seven.)
    SFM_STAMP(4);
    // the two lanes of a hypothesis: first half + second half
    // (wide waves: a lane has scored all of its hypothesis' points of this range itself, in point order)
    const int c_other = WIDE ? 0 : __shfl_xor(c, 32, 64);
    const double a1_other = WIDE ? 0.0 : __shfl_xor(a1, 32, 64), a2_other = WIDE ? 0.0 : __shfl_xor(a2, 32, 64);
    const int ck = c + c_other;
    const double s1k = WIDE ? a1 : (half == 0 ? a1 + a1_other : a1_other + a1);
    const double s2k = WIDE ? a2 : (half == 0 ? a2 + a2_other : a2_other + a2);
    if ((WIDE || half == 0) && valid) {
        const int64_t hp = sfmws::split_padded(h_count);
        if (units <= 1) {   // the totals, with the correction for the sample points behind them
            const int32_t* fix_c = reinterpret_cast<const int32_t*>(a.fix);
            const double* fix_a1 = reinterpret_cast<const double*>(a.fix + 4 * hp);
            cnt[h] = ck + fix_c[h];
            s1[h] = s1k + fix_a1[h];
            s2[h] = s2k + fix_a1[hp + h];
        } else {
            // Range split: this range's partial goes to [range][hypothesis] with PLAIN stores; matrix_fold_kernel, launched
            // behind this kernel, adds the ranges in range order (a fixed order: identical sums from run to run).  Until round 4
            // the ranges met on a per-hypothesis arrival counter — three write-through stores, an agent-scope read-modify-write
            // and, in the last arriver, 3 x ranges agent-scope loads per (hypothesis, range): 800 000 device-scope atomics and
            // 2.4 M write-through stores per launch, which the memory side of eight non-coherent L2s serves at ~1.5 M / ms — with
            // the step loop stubbed out the launch still took 0.6 ms, two thirds of every wave's life spent in that hand-off
            // (profiles/r04/README.md); a kernel boundary orders the same data for nothing.
            int32_t* part_c = reinterpret_cast<int32_t*>(a.split) + hp;
            double* part_a1 = reinterpret_cast<double*>(part_c + (int64_t)units * hp);
            double* part_a2 = part_a1 + (int64_t)units * hp;
            part_c[unit * hp + h] = ck;
            part_a1[unit * hp + h] = s1k;
            part_a2[unit * hp + h] = s2k;
        }
    }
#if SFM_MATRIX_STAMPS
    SFM_STAMP(5);
    const unsigned wave_id = item_id;
    if (lane == 0 && wave_id < 65536u) {
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
#pragma unroll
        for (int k = 0; k < 6; ++k) g_matrix_stamps[10 * wave_id + k] = stamp[k];
        g_matrix_stamps[10 * wave_id + 6] = ((unsigned long long)xcc_id << 32) | hw_id;
        g_matrix_stamps[10 * wave_id + 7] = ((unsigned long long)unit << 32) | (unsigned)h0;
        g_matrix_stamps[10 * wave_id + 9] = __builtin_amdgcn_s_memtime() - clock_begin;
    }
    {   // exact-tier lane slots (rounds x pops, the same in every lane) and points popped, summed over the wave
        unsigned long long pops = stat_pops;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pops += __shfl_xor((unsigned)pops, off, 64);
        if (lane == 0 && wave_id < 65536u) g_matrix_stamps[10 * wave_id + 8] = ((unsigned long long)stat_rounds << 32) | (unsigned)pops;
    }
#endif
}

// Totals of a range-split launch: partials [range][hypothesis] added in range order, then the sample correction.
__global__ __launch_bounds__(256) void matrix_fold_kernel(const unsigned char* __restrict__ split, const unsigned char* __restrict__ fix,
                                                          int units, int h_count, int32_t* __restrict__ cnt,
                                                          double* __restrict__ s1, double* __restrict__ s2) {
    const int64_t pair = blockIdx.y;
    const int64_t hp = sfmws::split_padded(h_count);
    split += pair * sfmws::split_bytes(h_count, units);
    if (fix != nullptr) fix += pair * sfmws::matrix_fix_bytes(h_count);
    cnt += pair * (int64_t)h_count;
    s1 += pair * (int64_t)h_count;
    s2 += pair * (int64_t)h_count;
    const int32_t* part_c = reinterpret_cast<const int32_t*>(split) + hp;
    const double* part_a1 = reinterpret_cast<const double*>(part_c + (int64_t)units * hp);
    const double* part_a2 = part_a1 + (int64_t)units * hp;
    const int32_t* fix_c = reinterpret_cast<const int32_t*>(fix);
    const double* fix_a1 = reinterpret_cast<const double*>(fix + 4 * hp);
    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < h_count; h += (int64_t)gridDim.x * blockDim.x) {
        int total = part_c[h];
        double t1 = part_a1[h], t2 = part_a2[h];
        for (int u = 1; u < units; ++u) {
            total += part_c[u * hp + h];
            t1 += part_a1[u * hp + h];
            t2 += part_a2[u * hp + h];
        }
        if (fix != nullptr) {   // (the VALU-filter kernel corrects for its sample points itself)
            total += fix_c[h];
            t1 += fix_a1[h];
            t2 += fix_a1[hp + h];
        }
        cnt[h] = total;
        s1[h] = t1;
        s2[h] = t2;
    }
}

// The launch.  tickets == nullptr: one item per wave, placed by block index (batches of pairs with their XCD-aware block map).
// tickets != nullptr (a single pair): PERSISTENT waves — the grid is what the chip holds at once (CUs x SFM_MATRIX_OCC blocks),
// and every wave takes items from a counter until they run out.  Why: the hardware's workgroup dispatcher does not keep this
// kernel's slots full.  With one block per 4 items (round 3) the stamps of every wave (tools/r04/matrix_timeline.py,
// profiles/r04) showed, per XCD, ONE shader engine at its 128 waves and the other three at 30-45 for most of the launch while
// thousands of blocks were waiting — blocks are handed to the engines in turn, and the turn waits for the full one — 2600 of
// 4096 slots occupied on average, the launch ending when the slowest XCD did.  Waves that fetch their own work do not depend on
// any of that: a slot is busy until the items are gone, and the items can be as fine as the balance wants.
// Items of XCD x (tickets[16 x]): the ranges u = x (mod 8) of every group when the ranges are a multiple of eight — an XCD then
// streams only its own eighth of the point operand table through its L2, as the block order of round 3 did; otherwise one
// counter serves all.  The counters are zeroed by score_reset_kernel (they live behind the class counters).
// WIDE_WAVES (one pair, in cost order, many hypotheses: launch_matrix): the entries [0, wide_from) of the order — the heaviest
// hypotheses, whose items are the longest of the launch — go in waves of 32 as everywhere else, the entries behind them in waves
// of 64 (matrix_item's WIDE): the step's operands are loaded once for 2048 evaluations instead of 1024 — the operand stream keeps
// a CU's vector L1 ~90 % busy otherwise (profiles/r05/README.md) — and a lane scores all 32 points of a step under its
// hypothesis.  All of them wide, the heaviest items would take as long as the whole launch (50 000 x 100 000: 888 us median for
// the first 2048 entries, 1.27 ms the longest, of a 1.27 ms launch).  wide_from is a multiple of 128: whole blocks of either kind.
// The kernel with wide waves needs 163 VGPRs: three waves per SIMD.
template <bool ESTIMATE, int MASK_GROUP = 0, bool WIDE_WAVES = false>
__global__ __launch_bounds__(256, WIDE_WAVES ? kWideOcc : SFM_MATRIX_OCC) void score_sed_matrix_kernel(
    const Corr* __restrict__ pts, const uint4* __restrict__ hyp_table, const uint4* __restrict__ table, int n,
    const double* __restrict__ E, int h_count, double thr, const int32_t* __restrict__ order, int32_t* __restrict__ cnt,
    double* __restrict__ s1, double* __restrict__ s2, int units, int steps_per_unit, unsigned char* __restrict__ split,
    const unsigned char* __restrict__ fix, int batch, int blocks_per_pair, int32_t* __restrict__ tickets,
    uint16_t* __restrict__ record, int range_stride, int record_ranges, const int32_t* __restrict__ wide_word, int wide_max) {
    __shared__ alignas(kCap * kWave * 4) uint32_t queues[ESTIMATE ? 1 : 256 / kWave][ESTIMATE ? 1 : kCap][kWave];   // (a wave's ring: 8 KiB, aligned: ring_slot() in matrix_item)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    uint32_t* const my_queue = &queues[ESTIMATE ? 0 : wave_in_block][0][lane];
    MatrixPair a{pts, hyp_table, table, E, order, cnt, s1, s2, split, fix, record, range_stride};
    // wave index of the launch -> first entry of the order it takes, and the kind of wave: indices below narrow_waves are waves of
    // 32 over the entries [0, wide_from), the rest waves of 64 behind them
    // (a pair's wave indices [0, wide_max / 32) are waves of 32 — those at or behind wide_from have nothing to do —, the rest waves
    // of 64 from wide_from on; wide_from: what the sort left in the pair's word, anywhere in [0, wide_max])
    // (wide_word == nullptr: every wave wide — the cost pre-pass of a measurement build)
    int wide_from = WIDE_WAVES ? (wide_word != nullptr ? min(*wide_word, h_count) : 0) : h_count;   // (wave-uniform: a scalar load; a batch: set again below, per pair)
    const int narrow_waves = wide_max / kHyps;
    auto item = [&](int wave, int unit, unsigned item_id, int ranges) __attribute__((always_inline)) {
        if (WIDE_WAVES && wave >= narrow_waves)
            matrix_item<ESTIMATE, MASK_GROUP, true>(a, n, h_count, thr, ranges, steps_per_unit, wide_from + (wave - narrow_waves) * 2 * kHyps, h_count,
                                                    unit, my_queue, lane, item_id);
        else
            matrix_item<ESTIMATE, MASK_GROUP, false>(a, n, h_count, thr, ranges, steps_per_unit, wave * kHyps, wide_from, unit, my_queue, lane,
                                                     item_id);
    };
    if (tickets != nullptr) {
        const int waves32 = WIDE_WAVES ? narrow_waves + (h_count - wide_from + 2 * kHyps - 1) / (2 * kHyps) : (h_count + kHyps - 1) / kHyps;
        const bool by_xcc = units % 8 == 0;
        unsigned xcc = 0;
        if (by_xcc) {
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
        }
        const int mine = by_xcc ? units / 8 : units;   // ranges of a group that one counter hands out
        const int items = waves32 * mine;
        // A wave empties its own XCD's counter first, then walks the other seven: every item is taken whatever the placement
        // of the launch's waves — on a device that exposes fewer than eight XCDs (a CPX / DPX / QPX partition, a CU mask) the
        // counters of the absent ones would otherwise never be served and their ranges' partials never written (round 4's
        // advisor).  On the full chip the seven extra tickets per wave are taken when the launch is draining anyway.
        for (unsigned k = 0; k < (by_xcc ? 8u : 1u); ++k) {
            const unsigned x = (xcc + k) & 7u;
            for (;;) {
                const int t = take_ticket(tickets + 16 * x);
                if (t >= items) break;   // (every wave gets here: the counter only grows)
                const int wave = t / mine;
                const int unit = by_xcc ? (int)x + 8 * (t % mine) : t % mine;
                item(wave, unit, (unsigned)(wave * units + unit), units);
            }
        }
        return;
    }
    int block_of_range = blockIdx.x, unit = 0;
    if (batch > 1) {
        // XCD-aware block -> (pair, block of the pair) map, as in score_sed_filtered_kernel: workgroups are dealt round-robin
        // over the 8 XCDs by linear id, so all blocks of a pair get ids of one residue class mod 8 and the pair's operand
        // table (96 bytes per point) and fp64 points stay in ONE L2 — and the pair's points are cut into ranges until its waves
        // fill an XCD by themselves: with eight small pairs resident per XCD their tables (1.3 MB each at 10 000 points) thrash
        // the 4 MB L2 (C5 without ranges: 4.6 ms against 3.2 with the VALU kernel, whose points take 16 bytes each).
        const int label = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int64_t pair = (int64_t)(j / blocks_per_pair) * 8 + label;
        block_of_range = j % blocks_per_pair;
        if (pair >= batch) return;   // padding of the last group of eight
        if (units > 1) {
            unit = block_of_range % units;
            block_of_range /= units;
            if (a.split != nullptr) a.split += pair * sfmws::split_bytes(h_count, units);   // (the pre-pass has ranges and no partials)
        }
        a.pts += pair * (int64_t)n;
        a.table += pair * table_steps(n) * kBlocks * 64;
        a.hyp_table += pair * (int64_t)h_count * 2 * kBlocks;
        a.E += pair * (int64_t)h_count * 9;
        if (a.order != nullptr) a.order += pair * (int64_t)h_count;
        a.cnt += pair * (int64_t)h_count;
        a.s1 += pair * (int64_t)h_count;
        a.s2 += pair * (int64_t)h_count;
        if (a.fix != nullptr) a.fix += pair * sfmws::matrix_fix_bytes(h_count);
        if (a.record != nullptr) a.record += pair * (sfmws::matrix_record_bytes(h_count) / 2);
        if (WIDE_WAVES && wide_word != nullptr) wide_from = min(wide_word[pair * sfmws::kBuckets], h_count);
    } else if (units > 1) {   // `units` consecutive blocks take the same hypotheses over one range of the points each
        unit = block_of_range % units;
        block_of_range /= units;
    }
    const int wave = block_of_range * (256 / kWave) + wave_in_block;
    if (ESTIMATE && record != nullptr) {
        // the recording pre-pass: a wave scans the first steps of SEVERAL of the scoring launch's `record_ranges` ranges — two for a
        // single pair (the launch has half as many units), all of them for a pair of a batch (one unit: 145 000 waves of 16 steps
        // for the 256 pairs of C5 cost more in starting up than the replay saves)
        const int per_wave = record_ranges / units;
        for (int p = 0; p < per_wave; ++p)
            item(wave, unit * per_wave + p, 0u, record_ranges);
        return;
    }
    item(wave, unit, blockIdx.x * (256 / kWave) + wave_in_block, units);
}

}  // namespace matrixscore
