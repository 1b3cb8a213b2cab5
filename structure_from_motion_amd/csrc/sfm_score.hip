// SED scoring of N correspondences under H hypotheses — the H x N loop of the RANSAC driver
// (reference lib/ransac/ransac.py:66-82, scorer lib/epipolar/sed.py:7-30).
//
// Two kernels produce the same (count, sum, sum of squares) per hypothesis:
//
//   score_sed_exact_kernel     every evaluation in fp64 (68 VALU instructions each).  Used when the
//                              caller passes no workspace, and as the cross-check of the other one.
//
//   score_sed_filtered_kernel  two tiers.  Tier 1 evaluates a CONSERVATIVE fp32 lower bound of the
//                              SED for every (hypothesis, point) and rejects the pair only when the
//                              bound proves sed > thr.  Survivors (about the inlier fraction, a few
//                              percent for a typical hypothesis) are compacted onto a per-wave LDS
//                              stack and tier 2 evaluates them with exactly the fp64 routine of the
//                              exact kernel.  Every point whose fp64 SED could be <= thr reaches
//                              tier 2, so counts and inlier decisions are identical to the exact
//                              kernel bit for bit; only the summation order of the two sums differs.
//
// Tier-1 bound (u = 2^-24; hats = values rounded to fp32; a = (xa, ya, 1), b = (xb, yb, 1)):
//   r  = b^T E a has 9 terms; r32 is the fp32 FMA evaluation of the same bilinear form.  Each term
//        carries <= 3 input roundings and <= 4 FMA roundings, so |r32 - r| <= gamma_7 * R with
//        R = sum |E_jk||b_j||a_k| <= Emax * M,  M >= (|xb|+|yb|+1)(|xa|+|ya|+1) for every point of the
//        data set.  The fp64 value r_fl of the exact path differs from r by <= 6 * 2^-53 * R.  With
//        delta = 10 u Emax M (inflated for its own roundings):   |r_fl| >= s := |r32| - delta.
//   la_j = (E a)_j : |la_j32 - la_j,fl| <= 5 u A_j,  A_j = |E_j0| Xa + |E_j1| Ya + |E_j2|, with Xa, Ya the
//        data-set maxima of |xa|, |ya|; eta_j := 6 u A_j.  With (|x| + eta)^2 <= (1+k) x^2 + (1 + 1/k) eta^2:
//        da_fl <= (1+k) (la_0,32^2 + la_1,32^2 + ca),  ca := (eta_a0^2 + eta_a1^2) / k,  k = 2^-10.
//        Same for lb = E^T b and db.  Write dA := la_0,32^2 + la_1,32^2 + ca, dB likewise.
//   sed_fl >= (1 - 5*2^-53) r_fl^2 (1/da_fl + 1/db_fl) >= s^2 (dA + dB) / ((1+k) dA dB).
//   Reject  <=>  s > 0  and  s^2 (dA + dB) > T dA dB  and  T dA dB > 1e-30,  T = thr (1+k) (1 + 1e-5).
//   The factor 1e-5 covers every rounding of the fp32 evaluation of both sides (< 30 u = 1.8e-6).
//   NaN / inf / overflow / underflow anywhere make a comparison false (or the guard fail) and the
//   pair goes to tier 2.  thr < 0 or NaN switches the filter off.
//   The steady-state loop uses the ONE-SIDED form of this test (drop the 1/da term: s^2 > T dB), which needs
//   neither la nor dA — reject_mask_one_sided below, which also folds T into the prepared coordinates, delta into
//   the dB slack and the underflow guard into a per-hypothesis check (12 VALU per evaluation);
//   SFM_SCORE_ONE_SIDED=0 restores the two-sided test everywhere (ablation).
//
// Compiled with -ffp-contract=off; the fp32 tier spells its FMAs explicitly.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "sfm_common.h"
#include "sfm_math.h"
#include "sfm_score_ws.h"
#include "sfm_score_matrix.h"

namespace {

using namespace sfmws;  // workspace layout: kBuckets, kPointsPad, ws_*_offset, workspace_bytes_for, ...
using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;
using sfmhost::grid_stride;

constexpr int kHypPerWave = 4;
#ifndef SFM_SCORE_PACKED
#define SFM_SCORE_PACKED 0   // tier 1 of two hypotheses per instruction in packed fp32 (experiment: profiles/r03/README.md)
#endif
#ifndef SFM_WAVE_STAMPS
#define SFM_WAVE_STAMPS 0   // diagnostic build (tools/wave_timeline.py): per-wave start / end stamps of the filtered kernel
#endif
#if SFM_WAVE_STAMPS
__device__ unsigned long long g_wave_stamps[4 * 65536];  // begin, end, first hypothesis, exact-tier batches; read by nothing but sfm_debug_read_wave_stamps
#endif
constexpr int kStack = 256;  // entries per (wave, hypothesis) survivor stack; <= 63 left + 128 pushed per step; popped in groups of 64

// ------------------------------------------------------------------------------------------------
// Epilogue shared by both kernels: fixed-order wave reduction + sample fix-up + store.
// The 8 sample points are never counted and always summed (ransac.py:70-79): the main loop treats them
// like any other point, lanes 0..7 then re-score them exactly and patch the totals.
// ------------------------------------------------------------------------------------------------
SFM_DEVICE void finish_hypothesis(const Corr* __restrict__ pts, const int32_t* __restrict__ sample,
                                  const double (&e)[9], double thr, int lane, int c, double a1, double a2,
                                  int32_t* cnt_out, double* s1_out, double* s2_out) {
    // the sample fix-up goes into the per-lane partials of lanes 0..7 first: one set of reductions, not two
    if (lane < 8) {
        const Corr p = pts[sample[lane]];
        const double sed = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
        const bool counted = sed <= thr;  // already in (c, a1, a2)
        c += counted ? -1 : 0;
        const double extra = counted ? 0.0 : sed;  // NaN / inf propagate: such a model never wins
        a1 += extra;
        a2 += extra * extra;   // the square of the masked value: one select instead of two, same bits
    }
    const int ck = sfm::wave_sum(c);
    const double s1k = sfm::wave_sum(a1);
    const double s2k = sfm::wave_sum(a2);
    if (lane == 0) {
        *cnt_out = ck;
        *s1_out = s1k;
        *s2_out = s2k;
    }
}

// ------------------------------------------------------------------------------------------------
// Exact kernel: one wave owns HPW hypotheses (E in scalar registers) and streams all n points.
// ------------------------------------------------------------------------------------------------
template <int HPW>
__global__ __launch_bounds__(256) void score_sed_exact_kernel(
    const Corr* __restrict__ corr, int n, const double* __restrict__ E, const int32_t* __restrict__ S,
    int h_count, double thr, int32_t* __restrict__ cnt, double* __restrict__ s1,
    double* __restrict__ s2) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave =
        __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave));
    const int h0 = wave * HPW;
    if (h0 >= h_count) return;
    const int64_t b = blockIdx.y;
    const Corr* __restrict__ pts = corr + b * (int64_t)n;
    const double* __restrict__ Eb = E + b * (int64_t)h_count * 9;
    const int32_t* __restrict__ Sb = S + b * (int64_t)h_count * 8;

    double e[HPW][9];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = min(h0 + k, h_count - 1);
#pragma unroll
        for (int j = 0; j < 9; ++j) e[k][j] = Eb[(int64_t)h * 9 + j];
    }
    int c[HPW];
    double a1[HPW], a2[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        c[k] = 0;
        a1[k] = 0.0;
        a2[k] = 0.0;
    }
    for (int i = lane; i < n; i += kWave) {
        const Corr p = pts[i];
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const double sed = sfm::sed_value(e[k], p.xa, p.ya, p.xb, p.yb);
            const bool ok = sed <= thr;
            c[k] += ok ? 1 : 0;
            const double kept = ok ? sed : 0.0;   // masked once; its square is the masked square (same bits, one select less)
            a1[k] += kept;
            a2[k] += kept * kept;
        }
    }
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = h0 + k;
        if (h < h_count) {  // wave-uniform
            const int64_t o = b * (int64_t)h_count + h;
            finish_hypothesis(pts, Sb + (int64_t)h * 8, e[k], thr, lane, c[k], a1[k], a2[k], cnt + o, s1 + o,
                              s2 + o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Workspace preparation for the filtered kernel: fp32 copy of the correspondences and the data-set
// maxima of |xa|, |ya|, |xb|, |yb| (as fp32 bit patterns: non-negative floats order like unsigned ints).
// Workspace layout: sfm_score_ws.h.
// ------------------------------------------------------------------------------------------------
// Zero the per-batch maxima and class counters.  A kernel rather than hipMemsetAsync: one launch instead of
// two, and a captured hipGraph of the pass then holds kernel nodes only (with hipMemsetAsync nodes in it, replaying
// the graph while a second captured graph was alive faulted on ROCm 7.2 — profiles/r01/README.md).
__global__ __launch_bounds__(256) void score_reset_kernel(unsigned char* __restrict__ ws, int32_t* __restrict__ buckets) {
    const int64_t b = blockIdx.x;
    static_assert(kBuckets % 256 == 0, "256 threads clear the counter words");
#pragma unroll
    for (int k = 0; k < kBuckets / 256; ++k) buckets[b * kBuckets + k * 256 + threadIdx.x] = 0;
    if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(ws + 16 * b)[threadIdx.x] = 0u;
}

// arrival counters of a range-split launch (one per hypothesis): re-armed before every scoring launch
__global__ __launch_bounds__(256) void score_split_reset_kernel(int32_t* __restrict__ arrivals, int64_t h_count,
                                                                int64_t ints_per_pair = 0) {
    arrivals += (int64_t)blockIdx.y * ints_per_pair;   // a batch keeps one region per pair
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < h_count; i += (int64_t)gridDim.x * blockDim.x)
        arrivals[i] = 0;
}

__global__ void score_prepare_kernel(const Corr* __restrict__ corr, int64_t n, double a_scale,
                                     unsigned char* __restrict__ ws) {
    const int64_t b = blockIdx.y;
    const int64_t batch = gridDim.y;
    const Corr* pts = corr + b * n;
    uint32_t* maxima = reinterpret_cast<uint32_t*>(ws + 16 * b);
    float4* out = reinterpret_cast<float4*>(ws + ws_points_offset(batch)) + b * n;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Corr p = pts[i];
        // a_scale: 1 for the two-sided test; c ~ 1/sqrt(T) for the one-sided one (see reject_mask_one_sided)
        const float4 q = to_filter_point(p, a_scale);
        out[i] = q;
        // NaN coordinates: fmaxf ignores them; such points always fail the filter's comparisons and are
        // decided by the exact tier.
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
    // one atomic per block and coordinate (all blocks hit the same cache line: keep the count low)
    __shared__ float part[4][4];
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        part[wave][0] = m0; part[wave][1] = m1; part[wave][2] = m2; part[wave][3] = m3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float m = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]),
                              fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
        // a single block per pair owns the result: plain store, and the launcher skips the zeroing kernel
        if (gridDim.x == 1) maxima[threadIdx.x] = __float_as_uint(m);
        else atomicMax(maxima + threadIdx.x, __float_as_uint(m));
    }
}

// Per-hypothesis constants of the fp32 filter (all wave-uniform).
struct FilterConsts {
    float e[9];        // E rounded to fp32
    float delta;       // >= |r32 - r_fl|
    float ca, cb;      // additive slack of the dA / dB upper bounds (see the bound above)
};

// `a_scale` is the factor the prepared a-side coordinates carry (Xa, Ya are maxima of the scaled values): the third
// homogeneous coordinate of a is a_scale instead of 1, i.e. the entries E_j2 that multiply it are scaled here.
SFM_DEVICE FilterConsts make_filter_consts(const double (&E)[9], float Xa, float Ya, float Xb, float Yb,
                                           double a_scale = 1.0) {
    constexpr float u = 5.9604644775390625e-08f;  // 2^-24
    constexpr float up = 1.0f + 1e-5f;            // absorbs the roundings of these bound computations
    FilterConsts f;
    float emax = 0.f;
    float poison = 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float ej = (float)E[j];
        emax = fmaxf(emax, fabsf(ej));
        // NaN anywhere in E must poison the bounds (fmaxf would drop it): add the entries' NaN-ness back.
        poison += ej * 0.0f;
        f.e[j] = (j % 3 == 2) ? (float)(E[j] * a_scale) : ej;
    }
    const float w = (float)a_scale * (1.0f + 1e-6f);  // >= the third coordinate of the scaled a
    const float M = (Xa + Ya + w) * (Xb + Yb + 1.0f);
    f.delta = (10.0f * u) * emax * M * up + poison;
    const float a0 = fabsf(f.e[0]) * Xa + fabsf(f.e[1]) * Ya + fabsf(f.e[2]);
    const float a1 = fabsf(f.e[3]) * Xa + fabsf(f.e[4]) * Ya + fabsf(f.e[5]);
    const float b0 = fabsf(f.e[0]) * Xb + fabsf(f.e[3]) * Yb + fabsf(f.e[6]);
    const float b1 = fabsf(f.e[1]) * Xb + fabsf(f.e[4]) * Yb + fabsf(f.e[7]);
    constexpr float inv_k = 1024.0f;
    const float ea0 = (6.0f * u) * a0 * up, ea1 = (6.0f * u) * a1 * up;
    const float eb0 = (6.0f * u) * b0 * up, eb1 = (6.0f * u) * b1 * up;
    f.ca = (ea0 * ea0 + ea1 * ea1) * (inv_k * up) + poison;
    f.cb = (eb0 * eb0 + eb1 * eb1) * (inv_k * up) + poison;
    return f;
}

SFM_DEVICE float uniform(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }

// true => the pair provably has sed > thr in the exact fp64 evaluation
SFM_DEVICE bool filter_rejects(const FilterConsts& f, float T, float xa, float ya, float xb, float yb) {
    const float lb0 = fmaf(xb, f.e[0], fmaf(yb, f.e[3], f.e[6]));
    const float lb1 = fmaf(xb, f.e[1], fmaf(yb, f.e[4], f.e[7]));
    const float lb2 = fmaf(xb, f.e[2], fmaf(yb, f.e[5], f.e[8]));
    const float r = fmaf(lb0, xa, fmaf(lb1, ya, lb2));
    const float la0 = fmaf(f.e[0], xa, fmaf(f.e[1], ya, f.e[2]));
    const float la1 = fmaf(f.e[3], xa, fmaf(f.e[4], ya, f.e[5]));
    const float s = fabsf(r) - f.delta;
    const float dA = fmaf(la0, la0, fmaf(la1, la1, f.ca));
    const float dB = fmaf(lb0, lb0, fmaf(lb1, lb1, f.cb));
    const float lhs = (s * s) * (dA + dB);
    const float rhs = T * (dA * dB);
    // bitwise '&': three compares and two s_and, no control flow
    return (s > 0.0f) & (lhs > rhs) & (rhs > 1e-30f);
}

// The one-sided form of the same test: sed = r^2 / da + r^2 / db >= r^2 / db, so  s > 0  and  s^2 > T dB  already
// proves sed_fl > thr — the bound derivation above cut short after the db term.  It needs neither la = E a nor dA.
// It is weaker: pairs with thr < sed <~ 2 thr (where da ~ db, the usual case for an essential matrix) slip through
// to tier 2 — about 40 % more tier-2 work, still a net gain (3.33 -> 3.17 ms on 50k x 100k when first introduced).
// Choosing per hypothesis between the db and the da form (sed is symmetric under E -> E^T, a <-> b) would guard
// against lopsided matrices, but the second code path costs a wave of occupancy (109 VGPRs, SGPR spills) — more than
// the filter saves; a lopsided E only makes tier 1 weaker, never wrong.
// The loop's form of the one-sided test, as a wave mask (bit = lane rejects): 12 VALU instructions per evaluation —
// 8 FMA for r', 2 FMA for dB, one square, one compare.
//  * No threshold multiply: the prepared a-side coordinates (and E_j2, the entries that meet a's third coordinate)
//    carry a factor c, so the bilinear form evaluates to r' = c r with the same rounding budget (every input still
//    rounded once) and delta' is computed from the scaled data-set maxima: |c r_fl| >= |r'| - delta'.
//  * No subtraction of delta' and no sign test: with k = 2^-10 and 2 x d <= k' x^2 + d^2 / k' (k' = k / (1 + k)),
//        (x - d)^2 >= x^2 / (1 + k) - d^2 / k,
//    so  r'^2 > dB + delta'^2 (1 + k) / k  implies  (|r'| - delta')^2 >= dB / (1 + k)  (and |r'| > delta'), hence
//    r_fl^2 >= dB / (c^2 (1 + k)) >= T dB  when  c^2 (1 + k) T <= 1:  c = (1 - 1e-6) / sqrt(T (1 + k)).  The delta term is
//    folded into cb once per hypothesis (arm_one_sided); it is ~1e-9 of a typical dB.
//    c = 0 switches the test off (thr negative, NaN, or T outside [1e-30, 1e30]): r' = 0 never exceeds dB >= 0.
//  * No underflow guard in the loop: dB >= cb (rounding is monotone), and arm_one_sided makes cb +inf where it
//    would be too small to trust (or NaN).
//  * The compare writes its lane mask straight to a scalar register pair; going through a per-lane bool and a
//    ballot makes the compiler materialise 0/1 in a VGPR and compare it again.
SFM_DEVICE unsigned long long reject_mask_one_sided(const FilterConsts& f, float xa_scaled, float ya_scaled, float xb,
                                                    float yb) {
    const float lb0 = fmaf(xb, f.e[0], fmaf(yb, f.e[3], f.e[6]));
    const float lb1 = fmaf(xb, f.e[1], fmaf(yb, f.e[4], f.e[7]));
    const float lb2 = fmaf(xb, f.e[2], fmaf(yb, f.e[5], f.e[8]));
    const float r = fmaf(lb0, xa_scaled, fmaf(lb1, ya_scaled, lb2));
    const float dB = fmaf(lb0, lb0, fmaf(lb1, lb1, f.cb));
    return __builtin_amdgcn_ballot_w64(r * r > dB);
}

// The same test for TWO hypotheses at once in packed fp32 (v_pk_fma_f32: both halves of a 64-bit register pair per
// instruction): the halves carry hypothesis A and hypothesis B, the point's coordinates are broadcast to both (op_sel
// picks the dword of the (xa', ya') / (xb, yb) register pair) and the multipliers / addends are packed (A, B) pairs — the
// four multipliers in SGPR pairs as before.  10 v_pk_fma_f32 + 1 v_pk_mul_f32 + 2 compares per point for two hypotheses
// instead of 24 instructions; every half is the IEEE fma of the plain form, so the masks are bit for bit the same.
typedef float float2v __attribute__((ext_vector_type(2)));
struct PackedConsts {
    float2v e0, e1, e3, e4;          // multipliers (SGPR pairs)
    float2v e2, e5, e6, e7, e8, cb;  // the other multipliers and the addends (wave-uniform VGPR pairs)
};
SFM_DEVICE float2v sgpr_pair(float a, float b) { return float2v{uniform(a), uniform(b)}; }
SFM_DEVICE PackedConsts pack_consts(const FilterConsts& a, const FilterConsts& b) {
    PackedConsts p;
    p.e0 = sgpr_pair(a.e[0], b.e[0]);
    p.e1 = sgpr_pair(a.e[1], b.e[1]);
    p.e3 = sgpr_pair(a.e[3], b.e[3]);
    p.e4 = sgpr_pair(a.e[4], b.e[4]);
    p.e2 = float2v{a.e[2], b.e[2]};
    p.e5 = float2v{a.e[5], b.e[5]};
    p.e6 = float2v{a.e[6], b.e[6]};
    p.e7 = float2v{a.e[7], b.e[7]};
    p.e8 = float2v{a.e[8], b.e[8]};
    p.cb = float2v{a.cb, b.cb};
    return p;
}
// d = broadcast(coord[SEL]) * mult + add, mult in an SGPR pair / in a VGPR pair
template <int SEL>
SFM_DEVICE float2v pk_fma_coord_s(float2v coord, float2v mult, float2v add) {
    float2v d;
    if (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(coord), "s"(mult), "v"(add));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(coord), "s"(mult), "v"(add));
    return d;
}
template <int SEL>
SFM_DEVICE float2v pk_fma_coord_v(float2v coord, float2v mult, float2v add) {
    float2v d;
    if (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(coord), "v"(mult), "v"(add));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(coord), "v"(mult), "v"(add));
    return d;
}
// d = value * broadcast(coord[SEL]) + add
template <int SEL>
SFM_DEVICE float2v pk_fma_by_coord(float2v value, float2v coord, float2v add) {
    float2v d;
    if (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(value), "v"(coord), "v"(add));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(value), "v"(coord), "v"(add));
    return d;
}
SFM_DEVICE float2v pk_fma(float2v a, float2v b, float2v c) {
    float2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
SFM_DEVICE float2v pk_mul(float2v a, float2v b) {
    float2v d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// lane masks of the pairs hypothesis A / hypothesis B reject; a_xy = (xa', ya'), b_xy = (xb, yb) of the lane's point
SFM_DEVICE void reject_masks_packed(const PackedConsts& f, float2v a_xy, float2v b_xy, unsigned long long& reject_a,
                                    unsigned long long& reject_b) {
    const float2v lb0 = pk_fma_coord_s<0>(b_xy, f.e0, pk_fma_coord_s<1>(b_xy, f.e3, f.e6));
    const float2v lb1 = pk_fma_coord_s<0>(b_xy, f.e1, pk_fma_coord_s<1>(b_xy, f.e4, f.e7));
    const float2v lb2 = pk_fma_coord_v<0>(b_xy, f.e2, pk_fma_coord_v<1>(b_xy, f.e5, f.e8));
    const float2v r = pk_fma_by_coord<0>(lb0, a_xy, pk_fma_by_coord<1>(lb1, a_xy, lb2));
    const float2v dB = pk_fma(lb0, lb0, pk_fma(lb1, lb1, f.cb));
    const float2v rr = pk_mul(r, r);
    reject_a = __builtin_amdgcn_ballot_w64(rr.x > dB.x);
    reject_b = __builtin_amdgcn_ballot_w64(rr.y > dB.y);
}

// Turn the constants of the two-sided test into those of reject_mask_one_sided: fold delta into cb (see there), and
// switch the filter off for this hypothesis (cb = +inf: nothing exceeds dB) when cb is too small to keep dB out of
// the denormal range, or NaN.
SFM_DEVICE void arm_one_sided(FilterConsts& f) {
    constexpr float kappa = 1.0f / 1024.0f;
    const float folded = f.cb + (f.delta * f.delta) * ((1.0f + kappa) / kappa * (1.0f + 1e-5f));
    f.cb = (f.cb > 1e-36f) ? folded : INFINITY;  // NaN cb, or NaN / inf delta -> NaN or inf: never rejects
    if (!(f.cb == f.cb)) f.cb = INFINITY;
}

// Factor carried by the prepared a-side coordinates (host side).
inline double one_sided_scale(double thr) {
    const double T = thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5);
    return (T > 1e-30 && T < 1e30) ? (1.0 - 1e-6) / sqrt(T * (1.0 + 1.0 / 1024.0)) : 0.0;  // NaN compares false
}

// ------------------------------------------------------------------------------------------------
// Load balancing.  A hypothesis that fits the scene keeps ~half of the points in tier 2 and costs several
// times the average; with only a few generations of waves per launch, such waves starting late leave the chip
// idle at the end.  A pre-pass estimates every hypothesis' tier-2 load (filter survivors among the first 1024
// points), a counting sort orders hypotheses by decreasing estimate, and the scoring kernel walks that order
// (longest first).  Results are still written at each hypothesis' own index: the order only affects speed.
// ------------------------------------------------------------------------------------------------
template <int HPW, bool ONE_SIDED>
__global__ __launch_bounds__(256) void score_estimate_kernel(const unsigned char* __restrict__ ws, int n,
                                                             const double* __restrict__ E, int h_count, double thr,
                                                             double a_scale, int32_t* __restrict__ estimate) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = blockIdx.x * (256 / kWave) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int h0 = wave * HPW;
    if (h0 >= h_count) return;
    const int64_t b = blockIdx.y;
    const double* __restrict__ Eb = E + b * (int64_t)h_count * 9;
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws + 16 * b);
    const float4* __restrict__ pts32 =
        reinterpret_cast<const float4*>(ws + ws_points_offset(gridDim.y)) + b * (int64_t)n;
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f);
    const float Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f);
    const float Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    const float T = (thr >= 0.0) ? (float)(thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5)) * (1.0f + 2e-7f) : INFINITY;
    FilterConsts f[HPW];
    int c[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = min(h0 + k, h_count - 1);
        double e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = Eb[(int64_t)h * 9 + j];
        f[k] = make_filter_consts(e, Xa, Ya, Xb, Yb, a_scale);
        if (ONE_SIDED) arm_one_sided(f[k]);
        c[k] = 0;
    }
    const int limit = min(n, kEstimatePoints);
    for (int i = lane; i < limit; i += kWave) {
        const float4 p = pts32[i];
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            if (ONE_SIDED) {  // lanes beyond `limit` are inactive and contribute no bits
                const unsigned long long rej = reject_mask_one_sided(f[k], p.x, p.y, p.z, p.w);
                c[k] += (int)((~rej >> lane) & 1ull);
            } else {
                c[k] += filter_rejects(f[k], T, p.x, p.y, p.z, p.w) ? 0 : 1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int total = sfm::wave_sum(c[k]);
        const int h = h0 + k;
        if (h < h_count && lane == 0) estimate[b * (int64_t)h_count + h] = total * 16;   // survivors per 1024 points, in sixteenths
    }
}

// Cost class of a hypothesis from its estimate = filter survivors per 1024 points in units of 1/16 (0 .. 16384): sixteen classes
// per power of two, heaviest first, kClasses - 1 = no survivors at all.  The matrix-pipe kernel (sfm_score_matrix.h) runs the 32
// hypotheses of a wave in lock step through its exact tier, so a wave's hypotheses should differ by per cents, not by 2 x
// (lane utilisation 0.66 with octave classes, 0.77 with quarter octaves, 0.8x with these and its 4096-point estimate).
SFM_DEVICE int cost_class(int estimate) {
    if (estimate <= 0) return kClasses - 1;
    const unsigned e = (unsigned)min(estimate, 16384);
    const int lg = 31 - __builtin_clz(e);                                               // 0..14
    const int sixteenth = lg >= 4 ? (int)((e >> (lg - 4)) & 15u) : (int)((e << (4 - lg)) & 15u);   // the four bits below the leading one
    return 16 * (14 - lg) + (15 - sixteenth);
}
static_assert(kEstimatePoints == 1024 && kClasses == 16 * 15 + 1, "cost_class: estimates up to 2^14 in 240 classes + one for zero");

// Histogram of the cost classes: a block counts in LDS (one LDS atomic per thread), then one global atomic per block and
// non-empty class, every class counter on its own cache line (atomics on one line serialise at ~10 ns each).
__global__ __launch_bounds__(256) void score_class_count_kernel(const int32_t* __restrict__ estimate, int h_count,
                                                                int32_t* __restrict__ buckets) {
    static_assert(kClasses <= 256, "one thread per class");
    __shared__ int block_count[256];
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    block_count[threadIdx.x] = 0;
    __syncthreads();
    if (h < h_count) atomicAdd(&block_count[cost_class(estimate[b * (int64_t)h_count + h])], 1);
    __syncthreads();
    if (threadIdx.x < kClasses && block_count[threadIdx.x] != 0)
        atomicAdd(buckets + b * kBuckets + threadIdx.x * kClassStride, block_count[threadIdx.x]);
}

// counts -> start offsets (heaviest class first); one block per batch entry, inclusive scan of the 256 class slots in LDS
__global__ __launch_bounds__(256) void score_class_scan_kernel(int32_t* __restrict__ buckets, int64_t batch) {
    __shared__ int scan[256];
    const int64_t b = blockIdx.x;
    const int c = threadIdx.x;
    const int own = c < kClasses ? buckets[b * kBuckets + c * kClassStride] : 0;
    scan[c] = own;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < 256; off <<= 1) {
        const int below = c >= off ? scan[c - off] : 0;
        __syncthreads();
        scan[c] += below;
        __syncthreads();
    }
    if (c < kClasses) buckets[b * kBuckets + c * kClassStride] = scan[c] - own;   // exclusive
}

// order[...] = hypothesis indices grouped by class, heaviest first (arbitrary order inside a class: the rank inside the
// block is the return value of the LDS atomic)
__global__ __launch_bounds__(256) void score_class_scatter_kernel(const int32_t* __restrict__ estimate, int h_count,
                                                                  int32_t* __restrict__ buckets,
                                                                  int32_t* __restrict__ order) {
    __shared__ int block_count[256];
    __shared__ int block_base[256];
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    block_count[threadIdx.x] = 0;
    __syncthreads();
    const int cls = h < h_count ? cost_class(estimate[b * (int64_t)h_count + h]) : -1;
    const int rank = cls >= 0 ? atomicAdd(&block_count[cls], 1) : 0;
    __syncthreads();
    if (threadIdx.x < kClasses && block_count[threadIdx.x] != 0)
        block_base[threadIdx.x] = atomicAdd(buckets + b * kBuckets + threadIdx.x * kClassStride, block_count[threadIdx.x]);
    __syncthreads();
    if (cls >= 0) order[b * (int64_t)h_count + block_base[cls] + rank] = h;
}

// The same with the scan inside (one launch fewer): every block scans the 241 class counts itself — they are final, the count
// kernel ran before — and claims its slots on a second word of each class' cache line (zeroed with the counters).
// wide_from_min > 0 (one pair): block 0 also leaves, in buckets[kWideFromWord], the entry of the order where the scoring launch's
// waves of 64 hypotheses begin — the first CLASS boundary at or behind entry wide_from_min, so that a hypothesis' kind of wave
// depends on its class alone (the order inside a class is the arrival order of atomics; the two kinds of wave add a hypothesis'
// points in different orders) — or 0 (waves of 64 everywhere) when no class starts in [wide_from_min, wide_from_max].
__global__ __launch_bounds__(256) void score_class_scan_scatter_kernel(const int32_t* __restrict__ estimate, int h_count,
                                                                       int32_t* __restrict__ buckets,
                                                                       int32_t* __restrict__ order, int wide_from_min, int wide_from_max) {
    __shared__ int scan[256];
    __shared__ int block_count[256];
    __shared__ int block_base[256];
    const int64_t b = blockIdx.y;
    const int c = threadIdx.x;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    const int own = c < kClasses ? buckets[b * kBuckets + c * kClassStride] : 0;
    scan[c] = own;
    block_count[c] = 0;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < 256; off <<= 1) {
        const int below = c >= off ? scan[c - off] : 0;
        __syncthreads();
        scan[c] += below;
        __syncthreads();
    }
    if (wide_from_min > 0 && blockIdx.x == 0 && c <= kClasses) {
        // start of class c (c == kClasses: the end of the order) and of the class before it
        const int start = c > 0 ? scan[c - 1] : 0;
        const int before = c > 1 ? scan[c - 2] : (c == 1 ? 0 : -1);
        if (start >= wide_from_min && before < wide_from_min) buckets[b * kBuckets + kWideFromWord] = start <= wide_from_max ? start : 0;
    }
    const int cls = h < h_count ? cost_class(estimate[b * (int64_t)h_count + h]) : -1;
    const int rank = cls >= 0 ? atomicAdd(&block_count[cls], 1) : 0;
    __syncthreads();
    if (c < kClasses && block_count[c] != 0)
        block_base[c] = scan[c] - own + atomicAdd(buckets + b * kBuckets + c * kClassStride + 1, block_count[c]);
    __syncthreads();
    if (cls >= 0) order[b * (int64_t)h_count + block_base[cls] + rank] = h;
}

// ------------------------------------------------------------------------------------------------
// Filtered kernel.
// ------------------------------------------------------------------------------------------------
// FUSED (the scoring launch of sfm_ransac_pass_small: small problems, one pair): the workspace was prepared by spare
// blocks of the fit launch, which leave one partial maximum per block of points instead of the data-set maxima
// (sfm_score_ws.h); everything else is the kernel as usual.
#ifndef SFM_SCORE_PREFETCH_DEPTH
#define SFM_SCORE_PREFETCH_DEPTH 2   // steps of point loads in flight in one-hypothesis waves (1 = the plain loop)
#endif
#ifndef SFM_SCORE_PREFETCH_DEPTH_MULTI
#define SFM_SCORE_PREFETCH_DEPTH_MULTI 1   // ... in waves with two or four hypotheses
#endif
template <int HPW, bool ONE_SIDED, bool FUSED = false>
__global__ __launch_bounds__(256, 5) void score_sed_filtered_kernel(
    const Corr* __restrict__ corr, const unsigned char* __restrict__ ws, int n,
    const double* __restrict__ E, const int32_t* __restrict__ S, int h_count, double thr, double a_scale,
    const int32_t* __restrict__ order, int32_t* __restrict__ cnt, double* __restrict__ s1,
    double* __restrict__ s2, int batch, int blocks_per_pair, int prep_blocks, int sync_every = 0, int units = 1,
    int chunks_per_unit = 0, unsigned char* __restrict__ split = nullptr) {
    // survivors' indices, one stack per (wave, hypothesis): pushes append at the top, the exact tier pops the top 64 —
    // which 64 of the queued points a batch takes does not matter (only the summation order depends on it, and that is
    // fixed), and a stack needs neither a wrap-around nor a second cursor
    __shared__ int32_t stack[256 / kWave][HPW][kStack];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    // XCD-aware block -> (pair, block of the pair) map.  Workgroups are dealt round-robin over the 8 XCDs by linear
    // id, each XCD with its own 4 MiB L2.  Every block of a pair re-reads that pair's points (fp32 copy for tier 1,
    // fp64 records gathered by tier 2: ~0.5 MB at N = 10k), so all blocks of a pair get ids of ONE residue class
    // mod 8: eight pairs are in flight at a time, one per L2, instead of every L2 holding a slice of every pair.
    int block_of_pair, pair;
    if (batch > 1 && blocks_per_pair > 0) {
        const int label = blockIdx.x & 7, j = blockIdx.x >> 3;
        pair = (j / blocks_per_pair) * 8 + label;
        block_of_pair = j % blocks_per_pair;
        if (pair >= batch) return;  // padding of the last group of eight
    } else {  // one pair, or a grid too large to flatten: plain (block, pair) grid
        pair = blockIdx.y;
        block_of_pair = blockIdx.x;
    }
    // Range split (single pair, units > 1): the points are cut into `units` ranges of chunks_per_unit 64-point chunks and
    // `units` consecutive blocks take the same hypotheses over one range each.  A launch of few generations ends with a
    // generation that drains for a whole wave duration (435 us of a 2.6 ms launch at 50 000 x 100 000: wave_timeline.py);
    // half as long waves halve that tail, for one more epilogue per hypothesis (< 2 % of a range's work).
    int unit = 0;
    if (units > 1) {
        unit = block_of_pair % units;
        block_of_pair /= units;
    }
    const int wave = block_of_pair * (256 / kWave) + wave_in_block;
    const int h0 = wave * HPW;  // first of this wave's HPW slots in the processing order
    if (h0 >= h_count) return;
#if SFM_WAVE_STAMPS
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memrealtime();
    int stamp_batches = 0;  // exact-tier batches of the wave's first hypothesis
#endif
    const int64_t b = pair;
    // slot -> hypothesis index (longest-first order from the pre-pass, or the identity)
    // With an order, the list (heaviest first) is dealt column-major over the waves: wave w takes entries
    // w, W + w, 2W + w, 3W + w (W = number of waves), i.e. one heavy and progressively lighter hypotheses, so
    // all waves carry about the same load.  Without an order the wave takes HPW consecutive hypotheses.
    const int waves_total = (h_count + HPW - 1) / HPW;
    int hyp[HPW];
    bool slot_valid[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int slot = order != nullptr ? k * waves_total + wave : h0 + k;
        slot_valid[k] = slot < h_count;
        const int s = min(slot, h_count - 1);
        hyp[k] = order != nullptr ? __builtin_amdgcn_readfirstlane(order[b * (int64_t)h_count + s]) : s;
    }
    const Corr* __restrict__ pts = corr + b * (int64_t)n;
    const double* __restrict__ Eb = E + b * (int64_t)h_count * 9;
    const int32_t* __restrict__ Sb = S + b * (int64_t)h_count * 8;
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws + 16 * b);
    const float4* __restrict__ pts32 =
        reinterpret_cast<const float4*>(ws + ws_points_offset(batch)) + b * (int64_t)n;

    // data-set coordinate maxima, inflated so they also bound the unrounded fp64 coordinates
    float Xa, Ya, Xb, Yb;
    if (FUSED) {  // the fit launch left one partial maximum per prepared block of points (sfm_score_ws.h)
        const uint32_t* partial = reinterpret_cast<const uint32_t*>(ws + ws_buckets_offset(n, 1));
        uint4 m = make_uint4(0u, 0u, 0u, 0u);
        if (lane < prep_blocks) m = reinterpret_cast<const uint4*>(partial)[lane];
        // prep_blocks <= 16: lanes 0..15 are one DPP row; four row rotations leave its maximum in every lane of the row
        // (all VALU: a __shfl_xor butterfly is four dependent trips through the LDS crossbar in every wave's prologue)
#define SFM_ROW_MAX(N)                                                        \
        m.x = max(m.x, (uint32_t)sfm::dpp_row_ror<N>((int)m.x));              \
        m.y = max(m.y, (uint32_t)sfm::dpp_row_ror<N>((int)m.y));              \
        m.z = max(m.z, (uint32_t)sfm::dpp_row_ror<N>((int)m.z));              \
        m.w = max(m.w, (uint32_t)sfm::dpp_row_ror<N>((int)m.w))
        SFM_ROW_MAX(8);
        SFM_ROW_MAX(4);
        SFM_ROW_MAX(2);
        SFM_ROW_MAX(1);
#undef SFM_ROW_MAX
        Xa = __uint_as_float(__builtin_amdgcn_readfirstlane(m.x)) * (1.0f + 1e-6f);
        Ya = __uint_as_float(__builtin_amdgcn_readfirstlane(m.y)) * (1.0f + 1e-6f);
        Xb = __uint_as_float(__builtin_amdgcn_readfirstlane(m.z)) * (1.0f + 1e-6f);
        Yb = __uint_as_float(__builtin_amdgcn_readfirstlane(m.w)) * (1.0f + 1e-6f);
    } else {
        Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f);
        Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
        Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f);
        Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    }
    // T = thr (1+k)(1 + 1e-5), rounded up; a negative or NaN threshold switches the filter off
    const float T = (thr >= 0.0) ? (float)(thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5)) * (1.0f + 2e-7f) : INFINITY;

    FilterConsts f[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = hyp[k];
        double e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = Eb[(int64_t)h * 9 + j];
        f[k] = make_filter_consts(e, Xa, Ya, Xb, Yb, a_scale);
        if (ONE_SIDED) arm_one_sided(f[k]);
        // The four multiplier entries go to scalar registers (a VALU instruction takes ONE scalar operand
        // for free); the five addend entries e2, e5, e6, e7, e8 and delta / eta stay in VGPRs as
        // wave-uniform values: an FMA whose multiplier and addend were both scalar would need an extra
        // v_mov, and the SGPR file (102) would be oversubscribed by 4 x 14 constants.
        f[k].e[0] = uniform(f[k].e[0]);
        f[k].e[1] = uniform(f[k].e[1]);
        f[k].e[3] = uniform(f[k].e[3]);
        f[k].e[4] = uniform(f[k].e[4]);
    }
#if SFM_SCORE_PACKED
    PackedConsts fp[(HPW + 1) / 2];   // the steady-state loop tests hypotheses two at a time (reject_masks_packed)
#pragma unroll
    for (int k = 0; k + 1 < HPW; k += 2) fp[k / 2] = pack_consts(f[k], f[k + 1]);
#endif

    int c[HPW];
    double a1[HPW], a2[HPW];
    int top[HPW];  // wave-uniform stack heights
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        c[k] = 0;
        a1[k] = 0.0;
        a2[k] = 0.0;
        top[k] = 0;
    }

    // tier 2: exact fp64 evaluation of `count` (<= 64) queued points of hypothesis k
    // (Deferring the evaluation by one drain, so that the gather of the fp64 records overlaps the next steps' tier 1,
    // changed nothing at any size: profiles/r02/README.md.)
    const sfm::SedGate gate = sfm::sed_gate(thr);
    auto drain = [&](int k, int count) __attribute__((always_inline)) {
        const int h = hyp[k];
        double e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = Eb[(int64_t)h * 9 + j];
        const bool active = lane < count;
        const int idx = active ? stack[wave_in_block][k][top[k] - count + lane] : 0;
        const Corr p = pts[idx];
        double sed;
        const bool ok = sfm::sed_inlier(e, p.xa, p.ya, p.xb, p.yb, gate, sed) && active;
        c[k] += ok ? 1 : 0;
        const double kept = ok ? sed : 0.0;   // masked once; its square is the masked square (same bits, one select less)
        a1[k] += kept;
        a2[k] += kept * kept;
        top[k] -= count;
#if SFM_WAVE_STAMPS
        if (k == 0) ++stamp_batches;
#endif
    };

    // Push the lanes flagged in `mask` onto the stack of hypothesis k: lane l of the mask writes `i` to slot
    // top + (number of flagged lanes below l).  Hand-written: the mask itself becomes the exec mask for the three address
    // instructions and the LDS write, so no per-lane predicate has to be rebuilt from it (the compiled form spends three
    // more VALU instructions per push on that; a push happens for most (chunk pair, hypothesis) combinations even when
    // the hypothesis fits nothing).  The byte address of the top slot is scalar (stack base + 4 top), so the slot
    // address is ONE shift-add on the prefix count.
    static_assert(kStack >= (kWave - 1) + 2 * kWave, "a step pushes up to 128 survivors on top of at most 63 left over");
    unsigned stack_base[HPW];  // LDS byte address of each hypothesis' stack (wave-uniform -> scalar registers)
#pragma unroll
    for (int k = 0; k < HPW; ++k)
        stack_base[k] = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) int32_t*)&stack[wave_in_block][k][0]);
    auto push = [&](int k, unsigned long long mask, int i) __attribute__((always_inline)) {
        const unsigned lo = (unsigned)mask, hi = (unsigned)(mask >> 32);
        const unsigned top_address = stack_base[k] + ((unsigned)top[k] << 2);  // scalar
        unsigned scratch;
        unsigned long long saved;
        asm volatile(
            "s_mov_b64 %[saved], exec\n\t"
            "s_mov_b64 exec, %[mask]\n\t"
            "v_mbcnt_lo_u32_b32 %[t], %[lo], 0\n\t"
            "v_mbcnt_hi_u32_b32 %[t], %[hi], %[t]\n\t"
            "v_lshl_add_u32 %[t], %[t], 2, %[top_address]\n\t"
            "ds_write_b32 %[t], %[index]\n\t"
            "s_mov_b64 exec, %[saved]"
            : [t] "=&v"(scratch), [saved] "=&s"(saved)
            : [mask] "s"(mask), [lo] "s"(lo), [hi] "s"(hi), [top_address] "s"(top_address), [index] "v"(i)
            : "memory");
        top[k] += (int)__popcll(mask);  // scalar arithmetic on a wave-uniform mask
    };

    // Steady state: two full 64-point chunks per step (no validity masks), so the scalar bookkeeping and
    // the drain test are paid once per 128 evaluations of a hypothesis.
    auto process_pair = [&](const float4 p0, const float4 p1, int i0, int i1) __attribute__((always_inline)) {
#if SFM_SCORE_PACKED
        unsigned long long packed_m0[HPW], packed_m1[HPW];
        if (ONE_SIDED && HPW % 2 == 0) {
            const float2v a0 = {p0.x, p0.y}, b0 = {p0.z, p0.w}, a1 = {p1.x, p1.y}, b1 = {p1.z, p1.w};
#pragma unroll
            for (int k = 0; k + 1 < HPW; k += 2) {
                unsigned long long ra, rb;
                reject_masks_packed(fp[k / 2], a0, b0, ra, rb);
                packed_m0[k] = ~ra;
                packed_m0[k + 1] = ~rb;
                reject_masks_packed(fp[k / 2], a1, b1, ra, rb);
                packed_m1[k] = ~ra;
                packed_m1[k + 1] = ~rb;
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            unsigned long long m0, m1;  // survivors of the two chunks
#if SFM_SCORE_PACKED
            if (ONE_SIDED && HPW % 2 == 0) {
                m0 = packed_m0[k];
                m1 = packed_m1[k];
            } else
#endif
            if (ONE_SIDED) {
                m0 = ~reject_mask_one_sided(f[k], p0.x, p0.y, p0.z, p0.w);
                m1 = ~reject_mask_one_sided(f[k], p1.x, p1.y, p1.z, p1.w);
            } else {
                m0 = ~__builtin_amdgcn_ballot_w64(filter_rejects(f[k], T, p0.x, p0.y, p0.z, p0.w));
                m1 = ~__builtin_amdgcn_ballot_w64(filter_rejects(f[k], T, p1.x, p1.y, p1.z, p1.w));
            }
            if ((m0 | m1) != 0ull) {  // wave-uniform
                push(k, m0, i0);
                push(k, m1, i1);
                __builtin_amdgcn_wave_barrier();
                while (top[k] >= kWave) drain(k, kWave);  // wave-uniform, at most twice
            }
        }
    };

    // Tail: one chunk with a validity mask.
    auto process_tail = [&](const float4 p, int i, bool valid) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const unsigned long long mask =
                ONE_SIDED ? ~reject_mask_one_sided(f[k], p.x, p.y, p.z, p.w) & __builtin_amdgcn_ballot_w64(valid)
                          : __builtin_amdgcn_ballot_w64(valid & !filter_rejects(f[k], T, p.x, p.y, p.z, p.w));
            if (mask != 0ull) {
                push(k, mask, i);
                __builtin_amdgcn_wave_barrier();
                if (top[k] >= kWave) drain(k, kWave);
            }
        }
    };

    const int full_chunks = n / kWave;
    const int chunks_total = (n + kWave - 1) / kWave;
    // this wave's chunks: [chunk_begin, chunk_end) — every chunk unless the launch is range-split
    const int chunk_begin = units > 1 ? unit * chunks_per_unit : 0;
    const int chunk_end = units > 1 ? min(chunk_begin + chunks_per_unit, chunks_total) : chunks_total;
    const int pairs = max(0, (min(chunk_end, full_chunks) - chunk_begin) / 2);
    // The point loads run kDepth steps ahead of the step being processed, in kDepth + 1 register stages that rotate by
    // unrolling, not by register moves (the plain "p = q" loop spent four 64-bit moves per step on that).  One
    // hypothesis per wave is the kernel of small launches (at most two generations of waves): whenever a SIMD holds few of
    // them — the waves that could not start with the first generation — a step costs one L2 round trip for the points
    // (~800 cycles against ~300 of work: profiles/r02/README.md, small_pass_timeline.log), so there two steps are in
    // flight; with more hypotheses per wave one step ahead is enough and the registers are needed elsewhere.
    constexpr int kDepth = HPW == 1 ? SFM_SCORE_PREFETCH_DEPTH : SFM_SCORE_PREFETCH_DEPTH_MULTI;
    {
        constexpr int kStages = kDepth + 1;
        // a prefetch reads up to kDepth steps past a pair's last full step: inside the workspace (the next pair's
        // points, or the kPointsPad bytes behind the last pair's) and never used
        static_assert(2048 * kDepth <= kPointsPad, "the prefetch runs this far past a pair's last full step");
        const float4* __restrict__ next = pts32 + chunk_begin * kWave + lane;  // per-lane cursor: one 64-bit add per step, no index clamping
        float4 stage[kStages][2];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {  // reads inside the workspace even when pairs < kDepth (points, then the pad)
            stage[d][0] = next[0];
            stage[d][1] = next[kWave];
            next += 2 * kWave;
        }
        int i0 = chunk_begin * kWave + lane, i1 = i0 + kWave;  // both chunks' point indices are carried: one add each per step, not one per hypothesis
        int pr = 0;
        int sync_countdown = 1;   // loop iterations (kStages steps each) until the next block barrier
        for (; pr + kStages <= pairs; pr += kStages) {
            // Small launches, one hypothesis per wave: every wave streams the whole point set itself, and the waves of a CU
            // drift apart until nothing hits in its 32 KB vector L1 (TCP_TCC_READ_REQ x 128 B = waves x set size: 797 MB,
            // 17 TB/s of L2 traffic at C2; profiles/r03/README.md).  A block barrier every few steps keeps the four waves
            // of a block within one L1's reach of each other, so some of them hit the lines the first one fetched
            // (L1 -> L2 requests -41 %, kernel -4 %: the traffic was the smaller part of that launch's problem).  Every
            // wave of a block runs the same number of steps (a function of n alone), and waves that ended are not waited for.
            if (FUSED && HPW == 1 && sync_every > 0 && --sync_countdown <= 0) {   // wave-uniform
                __builtin_amdgcn_s_barrier();
                sync_countdown = sync_every;
            }
#pragma unroll
            for (int u = 0; u < kStages; ++u) {
                stage[(u + kDepth) % kStages][0] = next[0];
                stage[(u + kDepth) % kStages][1] = next[kWave];
                next += 2 * kWave;
                process_pair(stage[u][0], stage[u][1], i0, i1);
                i0 += 2 * kWave;
                i1 += 2 * kWave;
            }
        }
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {  // the last pairs - pr <= kDepth steps are loaded already
            if (pr + u < pairs) {
                process_pair(stage[u][0], stage[u][1], i0, i1);
                i0 += 2 * kWave;
                i1 += 2 * kWave;
            }
        }
    }
    for (int chunk = chunk_begin + pairs * 2; chunk < chunk_end; ++chunk) {  // at most two iterations
        const int i = chunk * kWave + lane;
        process_tail(pts32[min(i, n - 1)], i, i < n);
    }
    // Epilogue per hypothesis: the last (partial) batch of the exact tier and the fix-up of the eight sample points
    // (finish_hypothesis: the scan treated them like any other point) are one evaluation when they fit one wave — lanes
    // [0, left) take the queued points, lanes [left, left + 8) the sample — instead of two gathers and two passes
    // through the fp64 routine, each with a fraction of the lanes (same-box A/B: 0.3 % of the large launch, 0.4 % of C2).
    // In a range-split launch only the wave of a hypothesis' first range fixes up the sample.
    const bool with_sample = units <= 1 || unit == 0;   // wave-uniform
    int mine_c = 0;               // range split: lane k keeps this range's totals of hypothesis k of the wave
    double mine_a1 = 0.0, mine_a2 = 0.0;
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = hyp[k];
        const int left = top[k];   // 0..63, wave-uniform
        double e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = Eb[(int64_t)h * 9 + j];
        if (with_sample && left <= kWave - 8) {
            const bool queued = lane < left;
            const bool sample = !queued && lane < left + 8;
            int idx = 0;
            if (queued) idx = stack[wave_in_block][k][lane];
            if (sample) idx = Sb[(int64_t)h * 8 + (lane - left)];
            const Corr p = pts[idx];
            double sed;
            const bool in = sfm::sed_inlier(e, p.xa, p.ya, p.xb, p.yb, gate, sed);
            // queued point: counted if it is an inlier.  Sample point: the scan has counted it already if it is within
            // the threshold (take it out of the count, keep it in the sums); otherwise add it to the sums only —
            // NaN / inf propagate: such a model never wins
            const bool add = queued ? in : (sample && !in);
            c[k] += queued ? (in ? 1 : 0) : ((sample && in) ? -1 : 0);
            const double kept = add ? sed : 0.0;
            a1[k] += kept;
            a2[k] += kept * kept;
            top[k] = 0;
#if SFM_WAVE_STAMPS
            if (k == 0 && left > 0) ++stamp_batches;
#endif
        } else {
            if (left > 0) drain(k, left);
            if (with_sample && lane < 8) {   // the sample fix-up of finish_hypothesis, into the per-lane partials
                const Corr p = pts[Sb[(int64_t)h * 8 + lane]];
                const double sed = sfm::sed_value(e, p.xa, p.ya, p.xb, p.yb);
                const bool counted = sed <= thr;  // already in (c, a1, a2)
                c[k] += counted ? -1 : 0;
                const double extra = counted ? 0.0 : sed;   // NaN / inf propagate: such a model never wins
                a1[k] += extra;
                a2[k] += extra * extra;
            }
        }
        if (slot_valid[k]) {  // wave-uniform: this slot exists
            const int ck = sfm::wave_sum(c[k]);
            const double s1k = sfm::wave_sum(a1[k]);
            const double s2k = sfm::wave_sum(a2[k]);
            if (units > 1) {
                if (lane == k) {
                    mine_c = ck;
                    mine_a1 = s1k;
                    mine_a2 = s2k;
                }
            } else if (lane == 0) {
                const int64_t o = b * (int64_t)h_count + h;
                cnt[o] = ck;
                s1[o] = s1k;
                s2[o] = s2k;
            }
        }
    }
    if (units > 1) {
        // Range split: lane k stores this range's partial of hypothesis k at [range][hypothesis]; matrixscore::matrix_fold_kernel,
        // launched behind this kernel, adds the partials in range order — a fixed order, so the sums are the same from run to
        // run — and writes the hypothesis' totals.  (Until round 4 the ranges met on a per-hypothesis arrival counter: device-scope
        // atomics are served by the memory side of the eight L2s at ~1.5 M per ms, and an ACQ_REL arrival at agent scope writes
        // the XCD's L2 back and invalidates it: the kernel boundary orders the same data for nothing — see sfm_score_matrix.h.)
        bool owner = false;
        int my_h = 0;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            owner |= (lane == k) && slot_valid[k];
            my_h = (lane == k) ? hyp[k] : my_h;
        }
        if (owner) {
            const int64_t hp = split_padded(h_count);
            int32_t* part_c = reinterpret_cast<int32_t*>(split) + hp;
            double* part_a1 = reinterpret_cast<double*>(part_c + (int64_t)units * hp);
            double* part_a2 = part_a1 + (int64_t)units * hp;
            part_c[unit * hp + my_h] = mine_c;
            part_a1[unit * hp + my_h] = mine_a1;
            part_a2[unit * hp + my_h] = mine_a2;
        }
    }
#if SFM_WAVE_STAMPS
    if (lane == 0 && wave < 65536) {
        g_wave_stamps[4 * wave] = stamp_begin;
        g_wave_stamps[4 * wave + 1] = __builtin_amdgcn_s_memrealtime();
        g_wave_stamps[4 * wave + 2] = (unsigned long long)hyp[0];
        g_wave_stamps[4 * wave + 3] = (unsigned long long)stamp_batches;
    }
#endif
}

// (The HIP events a benchmark brackets exactly the scoring kernel with travel in the call's sfm_score_options — timing_before /
// timing_after — since ABI 11; until then they were thread-local state set by sfm_score_set_timing_events.)
struct FilteredLaunch {
    const Corr* corr;
    unsigned char* ws;
    int n;
    const double* E;
    const int32_t* S;
    int h_count;
    double thr;
    bool use_order;
    int32_t* cnt;
    double* s1;
    double* s2;
    int32_t* buckets;
    int32_t* order;
    int64_t batch;
    hipStream_t st;
    bool one_sided;
    double a_scale;
    int units, chunks_per_unit;   // range split (single pair): 1, 0 = off
    bool xcd_map = true;          // batches: all blocks of a pair on one XCD (options.xcd_map)
    bool persistent = true;       // matrix-pipe kernel, single pair: persistent waves (options.persistent)
    unsigned* select_state = nullptr;          // fused pass: state words of its selection launch, zeroed with everything else
    sfmhost::LargeScore* deferred = nullptr;   // fused pass: leave the ranges' partials to the selection launch, report them here
    hipEvent_t event_before = nullptr, event_after = nullptr;   // options.timing_before / _after: recorded around the scoring kernel
    bool tables_ready = false;    // fused pass: maxima, zeroing and both operand tables of the matrix-pipe kernel are there already
    WsPlan plan = WsPlan{false, 1, false};   // what the workspace behind the scoring order was sized for
};

template <int HPW>
int launch_filtered(const FilteredLaunch& a) {
    const int64_t waves = (a.h_count + HPW - 1) / HPW;
    SFM_REQUIRE_GRID("sfm_score_sed", waves, 256 / kWave, 256, a.batch);
    const dim3 grid(grid_for(waves, 256 / kWave), (unsigned)a.batch);
    const int32_t* order_arg = nullptr;
    if (a.use_order) {
        // `cnt` doubles as the estimate buffer: it is rewritten by the scoring kernel afterwards
        if (a.one_sided)
            hipLaunchKernelGGL((score_estimate_kernel<HPW, true>), grid, dim3(256), 0, a.st, a.ws, a.n, a.E, a.h_count,
                               a.thr, a.a_scale, a.cnt);
        else
            hipLaunchKernelGGL((score_estimate_kernel<HPW, false>), grid, dim3(256), 0, a.st, a.ws, a.n, a.E, a.h_count,
                               a.thr, a.a_scale, a.cnt);
        SFM_REQUIRE_GRID("sfm_score_sed (ordering pre-pass)", a.h_count, 256, 256, a.batch);
        const dim3 per_hyp(grid_for(a.h_count, 256), (unsigned)a.batch);
        hipLaunchKernelGGL(score_class_count_kernel, per_hyp, dim3(256), 0, a.st, a.cnt, a.h_count, a.buckets);
        hipLaunchKernelGGL(score_class_scan_kernel, dim3((unsigned)a.batch), dim3(256), 0, a.st, a.buckets, a.batch);
        hipLaunchKernelGGL(score_class_scatter_kernel, per_hyp, dim3(256), 0, a.st, a.cnt, a.h_count, a.buckets, a.order);
        const int rc = check_launch("score order kernels");
        if (rc != SFM_OK) return rc;
        order_arg = a.order;
    }
    const int64_t flat_blocks = (int64_t)grid.x * ((a.batch + 7) / 8 * 8);
    const bool remap = a.xcd_map && a.batch > 1 && flat_blocks <= 0x7FFFFFFF;  // see the kernel's block -> (pair, block) map
    const int blocks_per_pair = remap ? (int)grid.x : 0;
    dim3 flat = remap ? dim3((unsigned)flat_blocks) : grid;
    unsigned char* split = nullptr;
    if (a.units > 1) {   // single pair: `units` consecutive blocks per group of waves; their partials are folded behind the launch
        flat = dim3(grid.x * (unsigned)a.units);   // (sfm_score_sed checked that this grid fits one launch)
        split = a.ws + ws_tail_offset(a.n, a.h_count, 1);
    }
    if (a.event_before) (void)hipEventRecord(a.event_before, a.st);
    if (a.one_sided)
        hipLaunchKernelGGL((score_sed_filtered_kernel<HPW, true>), flat, dim3(256), 0, a.st, a.corr, a.ws, a.n, a.E, a.S,
                           a.h_count, a.thr, a.a_scale, order_arg, a.cnt, a.s1, a.s2, (int)a.batch, blocks_per_pair, 0, 0,
                           a.units, a.chunks_per_unit, split);
    else
        hipLaunchKernelGGL((score_sed_filtered_kernel<HPW, false>), flat, dim3(256), 0, a.st, a.corr, a.ws, a.n, a.E, a.S,
                           a.h_count, a.thr, a.a_scale, order_arg, a.cnt, a.s1, a.s2, (int)a.batch, blocks_per_pair, 0, 0,
                           a.units, a.chunks_per_unit, split);
    if (a.event_after) (void)hipEventRecord(a.event_after, a.st);
    if (a.units > 1)
        hipLaunchKernelGGL(matrixscore::matrix_fold_kernel, dim3(grid_stride(a.h_count, 256, 1024), 1), dim3(256), 0, a.st, split,
                           (const unsigned char*)nullptr, a.units, a.h_count, a.cnt, a.s1, a.s2);
    return check_launch("score_sed_filtered_kernel");
}

// Launch options (sfm_score_options, include/sfm_hip.h).  The library never reads the process environment: a call carries its
// options (sfm_score_sed_ex) or takes the process-wide defaults, which the embedding application sets once
// (sfm_score_set_default_options — the Python package translates the SFM_SCORE_* variables at import).  A default set is an
// immutable heap copy behind an atomic pointer, so a call on another thread sees either the old set or the new one, whole.
constexpr sfm_score_options kBuiltinOptions = SFM_SCORE_OPTIONS_DEFAULT;
std::atomic<const sfm_score_options*> g_default_options{&kBuiltinOptions};

sfm_score_options resolve_options(const sfm_score_options* given) {
    return given ? *given : *g_default_options.load(std::memory_order_acquire);
}
bool valid_options(const sfm_score_options& o) {
    return (o.kernel == SFM_SCORE_KERNEL_AUTO || o.kernel == SFM_SCORE_KERNEL_FILTERED || o.kernel == SFM_SCORE_KERNEL_MATRIX) &&
           (o.hyps_per_wave == 0 || o.hyps_per_wave == 1 || o.hyps_per_wave == 2 || o.hyps_per_wave == 4) && o.split >= -1 &&
           o.order >= -1 && o.order <= 1 && o.one_sided >= -1 && o.one_sided <= 1 && o.xcd_map >= -1 && o.xcd_map <= 1 &&
           o.block_sync >= -1 && o.persistent >= -1 && o.persistent <= 1;
}

// The size rule between the two filtered kernels of sfm_score_sed (options.kernel forces the matrix-pipe kernel on where it
// applies / off).
bool use_matrix_kernel(int64_t n, int64_t h_count, int64_t batch, const sfm_score_options& opt) {
    const int matrix_env = opt.kernel == SFM_SCORE_KERNEL_MATRIX ? 1 : opt.kernel == SFM_SCORE_KERNEL_FILTERED ? 0 : -1;
    // By itself when the launch is large enough to fill the chip with its waves of 32 hypotheses over ranges of the points: at
    // least 4000 points, 2048 hypotheses and 3.5 x 10^8 evaluations (8192, 4096 and 5 x 10^8 until the wide waves of round 5; below
    // them now: 6000 x 100 000 481 vs 251 us, 5000 x 80 000 382 vs 245, 4000 x 200 000 469 vs 351, 100 000 x 3600 526 vs 489,
    // 200 000 x 2048 1000 vs 903; 3000 x 150 000 312 vs 335: the floor on the points — whole pass, default
    // route against the fused large pass with this kernel: 10 000 x 40 000 268 vs 227 us, 12 000 x 30 000 268 vs 222, 20 000 x 20 000
    // 275 vs 226, 30 000 x 15 000 321 vs 235, 40 000 x 10 000 337 vs 248; 25 000 x 12 000 219 vs 225 and 16 000 x 16 000 208 vs 214:
    // even; 8192 x 25 000 150 vs 208.  Round 3, same box, VALU vs matrix kernel: 50 000 x 100 000 2.49 vs 1.49
    // ms, x 125 000 2.97 vs 1.91, x 20 000 0.60 vs 0.43, 20 000 x 40 000 0.46 vs 0.31; but 16 000 x 16 000 0.16 vs 0.25, 8192 x
    // 25 000 0.14 vs 0.23: the heaviest wave of a small launch takes ~0.2 ms whatever the size; profiles/r03/README.md)
    // A batch of pairs (whole batched pipeline, VALU vs matrix kernel): C5 = 256 x 10 000 x 2 000 3.26 vs 3.04 ms, 128 x 16 384 x
    // 2 048 2.76 vs 2.38, 64 x 20 000 x 4 000 3.05 vs 2.63 — per pair an operand table, a cost pre-pass over an eighth of the
    // points, a counting sort and ranges with their own prologues, so the gain is smaller than for one large pair.
    const int64_t waves32_all = (h_count + matrixscore::kHyps - 1) / matrixscore::kHyps * batch;
    const bool matrix_fits = n <= matrixscore::kMaxPoints &&
                             sfmhost::grid_fits((int64_t)grid_for((h_count + 31) / 32, 256 / kWave) * ((batch + 7) / 8 * 8), 1, 256);
    const bool matrix_pays = batch == 1 ? (n >= 4000 && h_count >= 2048 && (double)n * (double)h_count >= 3.5e8)
                                        : (n >= 8192 && h_count >= 1024 && waves32_all >= 6144 &&
                                           (double)n * (double)h_count * (double)batch >= 5e8);
    return matrix_fits && (matrix_env > 0 || (matrix_env < 0 && matrix_pays));
}

// What a scoring call with a workspace will launch for (n, h_count, batch) under `opt` — decided ONCE, before anything is
// launched: sfm_score_sed_ex launches from it and sfm_score_workspace_bytes_ex sizes the workspace by it (ScorePlan::ws).
struct ScorePlan {
    WsPlan ws;          // matrix-pipe kernel or VALU filter; ranges of the points; recorded pre-pass
    int hpw;            // VALU filter: hypotheses per wave
    bool use_order;     // heaviest-first processing order (cost pre-pass + counting sort)
    bool one_sided;     // VALU filter: one-sided test
    double a_scale;     // factor the prepared a-side coordinates carry
    int per_unit;       // VALU filter: 64-point chunks per range; matrix-pipe kernel: steps of 32 points per range
};
int plan_score(int64_t n, int64_t h_count, int64_t batch, double thr, const sfm_score_options& opt, ScorePlan* plan) {
    // hypotheses per wave: 4 amortises the point loads best, but a launch with fewer waves than the chip holds
    // (5120) leaves SIMDs idle — then fewer hypotheses per wave = more waves wins (options.hyps_per_wave overrides)
    int hpw = kHypPerWave;
    if (opt.hyps_per_wave != 0) {
        hpw = opt.hyps_per_wave;
    } else {
        while (hpw > 1 && (h_count + hpw - 1) / hpw * batch < 5120) hpw /= 2;
    }
    SFM_REQUIRE_GRID("sfm_score_sed", (h_count + hpw - 1) / hpw, 256 / kWave, 256, batch);
    SFM_REQUIRE_GRID("sfm_score_sed (ordering pre-pass)", h_count, 256, 256, batch);
    const int64_t waves = (h_count + kHypPerWave - 1) / kHypPerWave;
    // longest-first processing order (cost pre-pass + counting sort); options.order = 0 keeps index order
    // It pays only when the launch has few generations of waves (a long wave starting late then idles the chip at
    // the end): 256 CUs x 20 resident waves = 5120 per generation; beyond ~12 generations the tail is negligible
    // and the pre-pass would cost more than it saves.
    const int order_env = opt.order;
    // ... and its fixed cost (~25 us) needs enough points per hypothesis to be won back (measured break-even ~8k).
    const bool use_order = order_env >= 0 ? order_env != 0 : (waves * batch <= 12 * 5120 && n >= 8192);
    // tier 1 on the matrix pipe (sfm_score_matrix.h): options.kernel forces it on (where it applies) / off
    const bool matrix = use_matrix_kernel(n, h_count, batch, opt);
    // options.one_sided = 0 switches tier 1 of the VALU filter back to the two-sided test (ablation)
    const bool one_sided = opt.one_sided != 0;
    plan->hpw = hpw;
    plan->one_sided = one_sided;
    plan->a_scale = matrix ? matrixscore::scale_for(thr) : (one_sided ? one_sided_scale(thr) : 1.0);
    const int split_env = opt.split;
    if (!matrix) {
        // Range split: a single-pair launch of 3 to 8 generations of waves (5120 each) is cut into 2 ranges of the points, so
        // that its last generation — a whole wave duration of draining chip — is half as long: 2.233-2.240 ms against
        // 2.279-2.297 ms at 50 000 x 100 000 (4.9 generations), 2.803 vs 2.831 at 125 000 hypotheses (6.1).  Four ranges
        // give the gain back (2.286 ms), launches of one or two generations lose (20 000 x 40 000: 0.480 vs 0.458 ms with
        // four ranges), more generations have no tail to speak of (profiles/r03/README.md).  options.split = 0 switches it
        // off, = k forces k ranges.
        int units = 1, chunks_per_unit = 0;
        if (batch == 1 && split_env != 0) {
            const int64_t launch_waves = (h_count + hpw - 1) / hpw;
            int want = split_env > 0 ? split_env : (launch_waves >= 3 * 5120 && launch_waves < 8 * 5120 ? 2 : 1);
            want = std::max(1, std::min(want, kSplitMaxUnits));
            const int chunks = (int)((n + kWave - 1) / kWave);
            chunks_per_unit = (chunks + want - 1) / want;
            chunks_per_unit += chunks_per_unit & 1;   // whole chunk pairs
            if (split_env <= 0) chunks_per_unit = std::max(chunks_per_unit, 64);   // ranges under 4096 points are mostly epilogue
            units = (chunks + chunks_per_unit - 1) / chunks_per_unit;
            if (units <= 1 || !sfmhost::grid_fits((int64_t)grid_for((h_count + hpw - 1) / hpw, 256 / kWave) * (int64_t)units, 1, 256)) {
                units = 1;
                chunks_per_unit = 0;
            }
        }
        plan->ws = WsPlan{false, units, false};
        plan->use_order = use_order;
        plan->per_unit = chunks_per_unit;
        return SFM_OK;
    }
    // Waves of 32 hypotheses are few (3125 at 100 000 hypotheses, against 12 288 resident ones) and long: the points are cut
    // into ranges so that the launch has about eight generations of waves — 50 000 x 100 000: 1.79 ms with 8 ranges, 1.96 with
    // 4, 3.4 with 2; x 20 000: 0.43 ms with 16 ranges, 0.69 with 8, 1.24 with 4 (profiles/r03/README.md).  A range keeps at
    // least 64 steps (2048 points): each range of a hypothesis pays its own epilogue.
    const int64_t waves32 = (h_count + matrixscore::kHyps - 1) / matrixscore::kHyps;
    // a batch: ranges until a pair's waves fill one XCD (512 resident waves) — see the kernel's block map
    const int64_t by_size = batch > 1 ? (512 + waves32 - 1) / waves32 : (8 * 3072 + waves32 - 1) / waves32;
    int want = split_env > 0 ? split_env : (int)std::min<int64_t>(kSplitMaxUnits, std::max<int64_t>(1, by_size));
    want = std::max(1, std::min(want, kSplitMaxUnits));
    // a multiple of eight ranges puts range u of every group on XCD u mod 8 (consecutive blocks are the ranges of one group,
    // blocks are dealt round-robin over the XCDs; the persistent waves hand out items the same way): an XCD then streams its
    // own eighth of the operand table through its L2 — 5 .. 11 ranges wanted -> 8, 12 and more -> 16 (50 000 x 125 000: 7
    // ranges 1.86 ms, 8 ranges 1.73)
    if (split_env <= 0 && batch == 1 && want >= 5) want = want >= 12 ? 16 : 8;
    const int steps = (int)matrixscore::steps_of(n);
    int steps_per_unit = (steps + want - 1) / want;
    if (split_env <= 0) steps_per_unit = std::max(steps_per_unit, batch > 1 ? 32 : 64);
    steps_per_unit = (steps_per_unit + 3) & ~3;   // ranges start on the step loop's group boundaries (groups of kAhead + 1 <= 4 steps)
    int m_units = (steps + steps_per_unit - 1) / steps_per_unit;
    const bool must_split = steps > matrixscore::kMaxRangeSteps;   // (a queue entry keeps its step relative to the range in 16 bits)
    if (must_split && (steps_per_unit > matrixscore::kMaxRangeSteps || m_units <= 1 || split_env == 0)) {
        steps_per_unit = matrixscore::kMaxRangeSteps;   // more than 2 M points per pair: ranges of 2^16 steps, whatever was asked for
        m_units = (steps + steps_per_unit - 1) / steps_per_unit;
        if (!sfmhost::grid_fits((int64_t)grid_for(waves32, 256 / kWave) * (int64_t)m_units * ((batch + 7) / 8 * 8), 1, 256))
            return fail(SFM_EINVAL, "sfm_score_sed: hypotheses x ranges exceed what one launch covers");
    } else if (m_units <= 1 || split_env == 0 ||
               !sfmhost::grid_fits((int64_t)grid_for(waves32, 256 / kWave) * (int64_t)m_units * ((batch + 7) / 8 * 8), 1, 256)) {
        m_units = 1;
        steps_per_unit = steps;
    }
    const bool matrix_order = order_env != 0 && h_count > (batch > 1 ? 64 : 2047);
    // The pre-pass' tier-1 results are kept for the scoring launch (MatrixPair::record): the pre-pass then scans the first
    // kReplaySteps steps of every range of the scoring launch — which must all have that many — instead of the first steps of the
    // points, records the reject words, and the scoring waves replay them.  Taken when that scan is exactly as large as the
    // pre-pass would be otherwise (8 ranges x 16 steps = the 128 steps of a large single pair; with 16 ranges — 50 000 x 20 000,
    // 20 000 x 40 000 — the larger pre-pass costs 3-8 us more than the replay saves), and for a single pair only: for
    // the 256 pairs of C5 — 9 ranges of 36 steps, so 44 % of all reject words would go through memory, 590 MB written and read —
    // it was measured a loss (2.71 ms per batch against 2.57-2.69, whether a pre-pass wave took one range or all nine).
    bool record = false;
#if SFM_MATRIX_REPLAY
    using matrixscore::kReplaySteps;
    record = matrix_order && batch == 1 && m_units >= 2 && m_units % 2 == 0 && m_units * kReplaySteps == matrixscore::estimate_steps(n) &&
             steps_per_unit >= kReplaySteps && steps - (m_units - 1) * steps_per_unit >= kReplaySteps;
#endif
    plan->ws = WsPlan{true, m_units, record};
    plan->use_order = matrix_order;
    plan->per_unit = steps_per_unit;
    return SFM_OK;
}

// compute units of the current device (256 on MI355X): the grid of the persistent-wave launch
int compute_units() {
    static std::atomic<int> cached[64];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return 256;
    int cus = cached[device].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        cached[device].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// Scoring launch with tier 1 on the matrix pipe (sfm_score_matrix.h): one pair, workspace prepared with that kernel's scale.
int launch_matrix(const FilteredLaunch& a) {
    using namespace matrixscore;
    const unsigned pairs = (unsigned)a.batch;
    const uint4* table = reinterpret_cast<const uint4*>(a.ws + ws_matrix_offset(a.n, a.h_count, a.batch, a.plan));
    static_assert(kBlocks * 2 * 16 == 96, "sfm_score_ws.h sizes the tables: 3 blocks");
    const uint4* hyp_table = reinterpret_cast<const uint4*>(a.ws + ws_matrix_hyp_offset(a.n, a.h_count, a.batch, a.plan));
    unsigned char* fix = a.ws + ws_matrix_fix_offset(a.n, a.h_count, a.batch, a.plan);   // the sample corrections, per pair
    // partial maxima (in the workspace's fp32-point region, which this kernel does not use: one float4 per setup block and pair) +
    // all zeroing, then both tables — unless a fused pass has done all of that already (tables_ready: launch_large_setup in front of
    // its fit launch, whose lanes wrote the hypotheses' rows and sample corrections; the point table by blocks of the fit launch
    // for one pair, by a points-only launch of matrix_tables_kernel for a batch)
    if (!a.tables_ready) {
        float4* partial = reinterpret_cast<float4*>(a.ws + ws_points_offset(a.batch));
        const unsigned setup_blocks = grid_stride(a.n, 1024, kSetupBlocks);
        hipLaunchKernelGGL(matrix_setup_kernel, dim3(setup_blocks, pairs), dim3(256), 0, a.st, a.corr, a.n, a.a_scale, partial, a.buckets,
                           a.cnt, a.h_count, a.select_state);
        const int step_blocks = (int)((table_steps(a.n) + 3) / 4);
        hipLaunchKernelGGL(matrix_tables_kernel, dim3((unsigned)step_blocks + grid_for(2 * (int64_t)a.h_count, 256), pairs), dim3(256), 0, a.st,
                           a.corr, a.n, a.a_scale, (const float4*)partial, (int)setup_blocks, const_cast<uint4*>(table), step_blocks, a.E,
                           a.h_count, const_cast<uint4*>(hyp_table), a.S, a.thr, fix);
    }
    // a single pair: persistent waves — as many blocks as the chip holds at once, every wave takes (group of 32 hypotheses,
    // range) items from a per-XCD counter (see the kernel); the counters were zeroed with the class counters
    const bool persistent = a.batch == 1 && a.persistent;
    // one pair, in cost order, many hypotheses: waves of 64 hypotheses behind the heaviest entries of the order (the kernel's
    // WIDE_WAVES); everything else — batches, unordered launches, fewer hypotheses — waves of 32
    const bool wide_waves = kWideWaves && a.use_order &&
                            a.h_count >= (a.batch == 1 ? SFM_MATRIX_WIDE_MIN_HYPOTHESES : SFM_MATRIX_WIDE_MIN_HYPOTHESES_BATCH);
    // entries of a pair's order that stay in waves of 32 at least — one pair: SFM_MATRIX_WIDE_FROM; a pair of a batch: an eighth
    // of its hypotheses, in whole blocks — and at most (four times that)
    const int wide_from_min = a.batch == 1 ? SFM_MATRIX_WIDE_FROM : std::max(128, (int)(a.h_count / 8 / 128 * 128));
    const int wide_from_max = 4 * wide_from_min;
    // (where the wide waves begin is decided on the device — the sort leaves it in the pair's buckets[kWideFromWord] —, anywhere from
    // 0 to wide_from_max: the grid covers wide_from_max entries in waves of 32 and all entries in waves of 64; waves without entries return)
    const int64_t waves = wide_waves ? wide_from_max / kHyps + (a.h_count + 2 * kHyps - 1) / (2 * kHyps) : (a.h_count + kHyps - 1) / kHyps;
    const unsigned blocks = grid_for(waves, 256 / kWave);
    const unsigned blocks_of_32 = grid_for((a.h_count + kHyps - 1) / kHyps, 256 / kWave);   // the cost pre-pass: waves of 32 always
    const unsigned resident_blocks = (unsigned)compute_units() * (unsigned)(wide_waves ? kWideOcc : SFM_MATRIX_OCC);
    // batches: flat grid of 8-pair groups (the kernel's block -> (pair, block) map), `units` range blocks per block of a pair; a
    // single pair: blocks x ranges
    const int blocks_per_pair = a.batch > 1 ? (int)blocks : 0;
    const unsigned flat = a.batch > 1 ? blocks * (unsigned)((a.batch + 7) / 8 * 8) : blocks;
    const int32_t* order_arg = nullptr;
    // (whether the pre-pass records its reject words for the scoring launch: plan_score)
    using matrixscore::kReplaySteps;
    uint16_t* record = a.plan.record ? reinterpret_cast<uint16_t*>(a.ws + ws_matrix_record_offset(a.n, a.h_count, a.batch, a.plan)) : nullptr;
    if (a.use_order) {
        // cost pre-pass with this kernel's own tier 1 over the first steps (survivors per 1024 points, in sixteenths, into
        // `cnt`, which the scoring launch rewrites), then the counting sort by class
        const int e_steps = estimate_steps(a.n);
        // a single pair: four ranges of the pre-pass's steps — or, recording, the first steps of each of the scoring launch's ranges
        // (a single pair: half as many units in the launch, each wave takes two of the ranges; a batch: one unit, all ranges — see the kernel)
#ifndef SFM_MATRIX_RECORD_RANGES_PER_WAVE
#define SFM_MATRIX_RECORD_RANGES_PER_WAVE 2   // recording pre-pass, single pair: ranges of the scoring launch one wave scans (1, 2, 4 or 8 of the 8)
#endif
        const int e_units = record != nullptr ? (a.batch == 1 ? std::max(1, a.units / SFM_MATRIX_RECORD_RANGES_PER_WAVE) : 1)
                                              : (a.batch == 1 && e_steps >= 128 ? 4 : 1);
        // (the ranges of the pre-pass add into `cnt`: zeroed by matrix_setup_kernel)
#ifndef SFM_MATRIX_WIDE_PREPASS
#define SFM_MATRIX_WIDE_PREPASS 0   // measurement build: the cost pre-pass of one pair in waves of 64 hypotheses as well
#endif
        const bool wide_prepass = SFM_MATRIX_WIDE_PREPASS && a.batch == 1;
        const unsigned blocks_prepass = wide_prepass ? grid_for((a.h_count + 2 * kHyps - 1) / (2 * kHyps), 256 / kWave) : blocks_of_32;
        const unsigned flat_of_32 = a.batch > 1 ? blocks_of_32 * (unsigned)((a.batch + 7) / 8 * 8) : blocks_prepass;
        const auto prepass_kernel = wide_prepass ? score_sed_matrix_kernel<true, 0, true> : score_sed_matrix_kernel<true, 0, false>;
        hipLaunchKernelGGL(prepass_kernel, dim3(flat_of_32 * (unsigned)e_units), dim3(256), 0, a.st, a.corr, hyp_table, table,
                           a.n, a.E, a.h_count, a.thr, (const int32_t*)nullptr, a.cnt, a.s1, a.s2, e_units,
                           record != nullptr ? kReplaySteps : e_steps / e_units,
                           (unsigned char*)nullptr, (const unsigned char*)nullptr, (int)a.batch, (a.batch > 1 ? (int)blocks_of_32 : 0) * e_units, (int32_t*)nullptr,
                           record, a.chunks_per_unit, a.units, (const int32_t*)nullptr, 0);
        const dim3 per_hyp(grid_for(a.h_count, 256), pairs);
        hipLaunchKernelGGL(score_class_count_kernel, per_hyp, dim3(256), 0, a.st, a.cnt, a.h_count, a.buckets);
        hipLaunchKernelGGL(score_class_scan_scatter_kernel, per_hyp, dim3(256), 0, a.st, a.cnt, a.h_count, a.buckets, a.order,
                           wide_waves ? wide_from_min : 0, wide_from_max);
        const int rc = check_launch("score order kernels");
        if (rc != SFM_OK) return rc;
        order_arg = a.order;
    }
    unsigned char* split = nullptr;   // partials of the ranges: [range][hypothesis], folded by matrix_fold_kernel behind the launch
    if (a.units > 1) split = a.ws + ws_tail_offset(a.n, a.h_count, a.batch);
    if (a.event_before) (void)hipEventRecord(a.event_before, a.st);
    const int64_t item_blocks = (int64_t)flat * a.units;
    const unsigned grid_blocks = persistent ? (unsigned)std::min<int64_t>(item_blocks, resident_blocks) : (unsigned)item_blocks;
    // (the pops of a round that share one execution-mask region: sfm_score_matrix.h, matrix_item's MASK_GROUP)
    const auto scoring_kernel = wide_waves ? score_sed_matrix_kernel<false, SFM_MATRIX_MASK_GROUP_WIDE, true>
                                : a.batch > 1 ? score_sed_matrix_kernel<false, SFM_MATRIX_MASK_GROUP_BATCH, false>
                                              : score_sed_matrix_kernel<false, SFM_MATRIX_MASK_GROUP_SINGLE, false>;
    hipLaunchKernelGGL(scoring_kernel, dim3(grid_blocks), dim3(256), 0, a.st, a.corr, hyp_table, table, a.n,
                       a.E, a.h_count, a.thr, order_arg, a.cnt, a.s1, a.s2, a.units, a.chunks_per_unit, split, fix,
                       (int)a.batch, blocks_per_pair * a.units, persistent ? a.buckets + kTicketWords : (int32_t*)nullptr, record, 0, 0,
                       wide_waves ? a.buckets + kWideFromWord : (const int32_t*)nullptr, wide_from_max);
    if (a.event_after) (void)hipEventRecord(a.event_after, a.st);
    if (a.deferred != nullptr) {   // a fused pass folds the ranges inside its selection launch
        *a.deferred = sfmhost::LargeScore{a.units, split, fix};
    } else if (a.units > 1) {
        hipLaunchKernelGGL(matrix_fold_kernel, dim3(grid_stride(a.h_count, 256, 1024), pairs), dim3(256), 0, a.st, split, fix, a.units,
                           a.h_count, a.cnt, a.s1, a.s2);
    }
    return check_launch("score_sed_matrix_kernel");
}

}  // namespace

namespace sfmhost {

bool score_options_valid(const sfm_score_options* options) { return options == nullptr || valid_options(*options); }

double small_pass_a_scale(double thr) { return one_sided_scale(thr); }

int32_t* small_pass_order(unsigned char* workspace, int64_t n, int64_t h_count) {
    // with every wave of the scoring launch resident at once (one generation) the order cannot change anything
    if (h_count <= 4096) return nullptr;
    return reinterpret_cast<int32_t*>(workspace + ws_order_offset(n, 1));
}

int launch_small_score(const SmallPass& p) {
    // hypotheses per wave: at most 32768 hypotheses are a few generations of waves at best, where two per wave (7
    // waves per SIMD) beat four (5 per SIMD; measured at 20 000 and 30 000 hypotheses: 80 vs 85 and 186 vs 198 us per
    // pass) and one per wave wins as long as two would leave the chip short of waves (profiles/r02/small_pass_hpw.log)
    const sfm_score_options opt = resolve_options(p.options);   // (NULL: the process-wide defaults)
    int hpw = (p.h_count + 1) / 2 >= 5120 ? 2 : 1;
    if (opt.hyps_per_wave != 0) hpw = opt.hyps_per_wave;
    const int64_t waves = (p.h_count + hpw - 1) / hpw;
    const int64_t blocks = (waves + 256 / kWave - 1) / (256 / kWave);
    SFM_REQUIRE_GRID("sfm_ransac_pass_small", blocks, 1, 256);
    const int prep_blocks = (int)((p.n + kPrepPoints - 1) / kPrepPoints);
    const dim3 grid((unsigned)blocks);
    // block barrier every `sync_every` iterations (three 128-point steps each) of the one-hypothesis-per-wave loop (0 = never): see the kernel
    // measured (profiles/r03/small_pass/block_barrier.log): every 1..4 iterations alike, -2 % at 5000 x 10000, -5 % at 7000..8000
    // points (L1 -> L2 requests -41 %), nothing below ~4000 points; options.block_sync overrides
    const int sync_every = opt.block_sync >= 0 ? opt.block_sync : (hpw == 1 && p.n >= 4096 ? 2 : 0);
    if (opt.timing_before) (void)hipEventRecord((hipEvent_t)opt.timing_before, p.stream);
#define SFM_LAUNCH_FUSED(H)                                                                                          \
    hipLaunchKernelGGL((score_sed_filtered_kernel<H, true, true>), grid, dim3(256), 0, p.stream, (const Corr*)p.corr, \
                       p.workspace, (int)p.n, p.E, p.S, (int)p.h_count, p.thr, one_sided_scale(p.thr),               \
                       (const int32_t*)small_pass_order(p.workspace, p.n, p.h_count), p.cnt, p.s1, p.s2, 1, 0, prep_blocks, \
                       sync_every)
    switch (hpw) {
        case 1: SFM_LAUNCH_FUSED(1); break;
        case 2: SFM_LAUNCH_FUSED(2); break;
        default: SFM_LAUNCH_FUSED(4); break;
    }
#undef SFM_LAUNCH_FUSED
    if (opt.timing_after) (void)hipEventRecord((hipEvent_t)opt.timing_after, p.stream);
    return check_launch("score_sed_filtered_kernel (fused small pass)");
}

}  // namespace sfmhost

#if SFM_MATRIX_STATS
extern "C" int sfm_debug_matrix_stats(unsigned long long* out, int reset) {   // rounds, points popped, push iterations (per wave)
    unsigned long long zero[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(matrixscore::g_matrix_stats), 32) != hipSuccess) return -2;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(matrixscore::g_matrix_stats), zero, 32) != hipSuccess) return -2;
    return 0;
}
#endif

#if SFM_MATRIX_STAMPS
extern "C" int sfm_debug_read_matrix_stamps(unsigned long long* out, int64_t waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(matrixscore::g_matrix_stamps), 80 * (size_t)waves) == hipSuccess ? 0 : -2;
}
#endif

#if SFM_WAVE_STAMPS
extern "C" int sfm_debug_read_wave_stamps(unsigned long long* out, int64_t waves) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_stamps), 32 * (size_t)waves) == hipSuccess ? 0 : -2;
}
#endif

namespace {
int score_sed_impl(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count, int64_t batch, double thr,
                   int32_t* cnt, double* s1, double* s2, void* workspace, int64_t workspace_bytes, void* stream,
                   const sfm_score_options& opt, unsigned* select_state, sfmhost::LargeScore* deferred, bool tables_ready = false);
}

extern "C" {

int64_t sfm_score_workspace_bytes_ex(int64_t n, int64_t h_count, int64_t batch, const sfm_score_options* options) {
    if (n < 0 || h_count < 0 || batch < 0 || n > 0x7FFFFFFF || h_count > 0x3FFFFFFF) return -1;
    const sfm_score_options opt = resolve_options(options);
    if (!valid_options(opt)) return -1;
    ScorePlan plan;
    plan.ws = WsPlan{false, 1, false};
    // (sizes no call accepts — fewer than 8 points, nothing to score, a grid beyond one launch — get the plain layout: the call
    // itself reports the error)
    if (n >= 8 && h_count >= 1 && batch >= 1 && plan_score(n, h_count, batch, 1.0, opt, &plan) != SFM_OK) plan.ws = WsPlan{false, 1, false};
    return workspace_bytes_for(n, h_count, batch, plan.ws);
}

int64_t sfm_score_workspace_bytes(int64_t n, int64_t h_count, int64_t batch) {
    return sfm_score_workspace_bytes_ex(n, h_count, batch, nullptr);
}

int sfm_score_kernel_choice_ex(int64_t n, int64_t h_count, int64_t batch, const sfm_score_options* options) {
    if (n < 0 || h_count < 0 || batch < 0) return -1;
    const sfm_score_options opt = resolve_options(options);
    if (!valid_options(opt)) return -1;
    return use_matrix_kernel(n, h_count, batch, opt) ? SFM_SCORE_KERNEL_MATRIX : SFM_SCORE_KERNEL_FILTERED;
}

int sfm_score_kernel_choice(int64_t n, int64_t h_count, int64_t batch) {
    return sfm_score_kernel_choice_ex(n, h_count, batch, nullptr);
}

int sfm_score_set_default_options(const sfm_score_options* options) {
    if (options && !valid_options(*options)) return fail(SFM_EINVAL, "sfm_score_set_default_options: a field is out of range");
    // (earlier sets stay allocated: a call that loaded the pointer just before may still be reading one; they are 32 bytes each
    // and set once per process in practice)
    sfm_score_options* copy = options ? new sfm_score_options(*options) : nullptr;
    if (copy) copy->timing_before = copy->timing_after = nullptr;   // events belong to one call, never to the defaults
    const sfm_score_options* next = copy ? copy : &kBuiltinOptions;
    g_default_options.store(next, std::memory_order_release);
    return SFM_OK;
}

int sfm_score_get_default_options(sfm_score_options* out) {
    if (!out) return fail(SFM_EINVAL, "sfm_score_get_default_options: null pointer");
    *out = resolve_options(nullptr);
    return SFM_OK;
}

int sfm_debug_matrix_filter(const double* corr, int64_t n, const double* E, int64_t h_count, double thr, void* workspace,
                            int64_t workspace_bytes, float* r_out, float* d_out, float* bound_out, void* stream) {
    if (n < 8 || h_count < 1 || n > matrixscore::kMaxPoints || h_count > 0x3FFFFFFF)
        return fail(SFM_EINVAL, "sfm_debug_matrix_filter: 8 <= n <= 4194304 points and at least one hypothesis");
    if (!corr || !E || !workspace || !r_out || !d_out || !bound_out) return fail(SFM_EINVAL, "sfm_debug_matrix_filter: null pointer");
    const WsPlan layout{true, 1, false};   // the matrix-pipe kernel's tables, no ranges
    if (workspace_bytes < workspace_bytes_for(n, h_count, 1, layout) || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_debug_matrix_filter: workspace of sfm_score_workspace_bytes_ex(n, h_count, 1, {kernel = matrix, split = 0}) "
                                "bytes, 16-byte aligned");
    using namespace matrixscore;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const int steps = (int)steps_of(n);
    const int64_t tiles = (h_count + kHyps - 1) / kHyps;
    if (!sfmhost::grid_fits(steps, 1, 64) || tiles > 65535) return fail(SFM_EINVAL, "sfm_debug_matrix_filter: too many tiles");
    // exactly what sfm_score_sed does in front of the matrix-pipe kernel: maxima + fp32 points, then the two operand tables
    const double a_scale = scale_for(thr);
    hipLaunchKernelGGL(score_reset_kernel, dim3(1), dim3(256), 0, st, ws, reinterpret_cast<int32_t*>(ws + ws_buckets_offset(n, 1)));
    hipLaunchKernelGGL(score_prepare_kernel, dim3(n <= 8192 ? 1u : grid_stride(n, 256, 64), 1), dim3(256), 0, st, (const Corr*)corr,
                       n, a_scale, ws);
    uint4* table = reinterpret_cast<uint4*>(ws + ws_matrix_offset(n, h_count, 1, layout));
    uint4* hyp_table = reinterpret_cast<uint4*>(ws + ws_matrix_hyp_offset(n, h_count, 1, layout));
    hipLaunchKernelGGL(matrix_prepare_kernel, dim3((unsigned)((table_steps(n) + 3) / 4), 1), dim3(256), 0, st, (const Corr*)corr, (int)n,
                       a_scale, ws, table);
    hipLaunchKernelGGL(matrix_hypothesis_kernel, dim3(grid_for(2 * h_count, 256), 1), dim3(256), 0, st, ws, E, (int)h_count, a_scale,
                       hyp_table, bound_out, (const Corr*)corr, (int)n, (const int32_t*)nullptr, thr, (unsigned char*)nullptr);
    hipLaunchKernelGGL(matrix_filter_dump_kernel, dim3((unsigned)steps, (unsigned)tiles), dim3(64), 0, st, hyp_table, table, steps,
                       (int)h_count, r_out, d_out);
    return check_launch("matrix_filter_dump_kernel");
}

int sfm_score_sed(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                  int64_t batch, double thr, int32_t* cnt, double* s1, double* s2, void* workspace,
                  int64_t workspace_bytes, void* stream) {
    return sfm_score_sed_ex(corr, n, E, S, h_count, batch, thr, cnt, s1, s2, workspace, workspace_bytes, stream, nullptr);
}

int sfm_score_sed_ex(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count,
                     int64_t batch, double thr, int32_t* cnt, double* s1, double* s2, void* workspace,
                     int64_t workspace_bytes, void* stream, const sfm_score_options* options) {
    const sfm_score_options opt = resolve_options(options);
    if (!valid_options(opt)) return fail(SFM_EINVAL, "sfm_score_sed_ex: an option is out of range");
    return score_sed_impl(corr, n, E, S, h_count, batch, thr, cnt, s1, s2, workspace, workspace_bytes, stream, opt, nullptr, nullptr);
}

}  // extern "C"

namespace {

int score_sed_impl(const double* corr, int64_t n, const double* E, const int32_t* S, int64_t h_count, int64_t batch, double thr,
                   int32_t* cnt, double* s1, double* s2, void* workspace, int64_t workspace_bytes, void* stream,
                   const sfm_score_options& opt, unsigned* select_state, sfmhost::LargeScore* deferred, bool tables_ready) {
    if (h_count < 0 || batch < 0 || n < 0) return fail(SFM_EINVAL, "sfm_score_sed: negative size");
    if (n > 0x7FFFFFFF || h_count > 0x3FFFFFFF) return fail(SFM_EINVAL, "sfm_score_sed: size too large");
    if (h_count == 0 || batch == 0) return SFM_OK;
    if (!corr || !E || !S || !cnt || !s1 || !s2) return fail(SFM_EINVAL, "sfm_score_sed: null pointer");
    if (n < 8) return fail(SFM_EINVAL, "sfm_score_sed: need at least 8 correspondences");
    hipStream_t st = (hipStream_t)stream;
    const int64_t waves = (h_count + kHypPerWave - 1) / kHypPerWave;
    SFM_REQUIRE_GRID("sfm_score_sed", waves, 256 / kWave, 256, batch);
    const dim3 grid(grid_for(waves, 256 / kWave), (unsigned)batch);
    if (workspace == nullptr) {
        if (opt.timing_before) (void)hipEventRecord((hipEvent_t)opt.timing_before, st);
        hipLaunchKernelGGL(score_sed_exact_kernel<kHypPerWave>, grid, dim3(256), 0, st, (const Corr*)corr, (int)n,
                           E, S, (int)h_count, thr, cnt, s1, s2);
        if (opt.timing_after) (void)hipEventRecord((hipEvent_t)opt.timing_after, st);
        return check_launch("score_sed_exact_kernel");
    }
    ScorePlan plan;
    const int rc_plan = plan_score(n, h_count, batch, thr, opt, &plan);   // every size check comes before the first launch: a refused call must not have touched the workspace
    if (rc_plan != SFM_OK) return rc_plan;
    if (workspace_bytes < workspace_bytes_for(n, h_count, batch, plan.ws))
        return fail(SFM_EINVAL, "sfm_score_sed: workspace smaller than sfm_score_workspace_bytes_ex(n, h_count, batch, options)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_score_sed: workspace must be 16-byte aligned");
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    int32_t* buckets = reinterpret_cast<int32_t*>(ws + ws_buckets_offset(n, batch));
    int32_t* order = reinterpret_cast<int32_t*>(ws + ws_order_offset(n, batch));
    if (plan.ws.matrix) {
        // (the matrix-pipe kernel: launch_matrix prepares everything itself in two launches — partial maxima + zeroing, then both
        // operand tables — instead of reset / prepare / point table / hypothesis table / estimate zeroing; batches too since round 5:
        // until then they went through score_prepare_kernel, which also wrote fp32 points that kernel never reads)
        const FilteredLaunch margs{(const Corr*)corr, ws, (int)n, E, S, (int)h_count, thr, plan.use_order, cnt, s1, s2,
                                   buckets, order, batch, st, true, plan.a_scale, plan.ws.units, plan.per_unit, true, opt.persistent > 0,
                                   select_state, deferred, (hipEvent_t)opt.timing_before, (hipEvent_t)opt.timing_after, tables_ready, plan.ws};
        return launch_matrix(margs);
    }
    // Small point sets are prepared by one block per pair, which stores the maxima itself; the zeroing kernel is then
    // only needed for the class counters of the ordering pre-pass (a small pass is a chain of ~4 us launches).
    const unsigned prepare_blocks = n <= 8192 ? 1u : grid_stride(n, 256, 64);
    if (plan.use_order || prepare_blocks > 1)
        hipLaunchKernelGGL(score_reset_kernel, dim3((unsigned)batch), dim3(256), 0, st, ws, buckets);
    hipLaunchKernelGGL(score_prepare_kernel, dim3(prepare_blocks, (unsigned)batch), dim3(256), 0, st, (const Corr*)corr, n, plan.a_scale, ws);
    const int rc = check_launch("score_prepare_kernel");
    if (rc != SFM_OK) return rc;
    if (select_state != nullptr)
        hipLaunchKernelGGL(score_split_reset_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<int32_t*>(select_state), (int64_t)16);
    const FilteredLaunch args{(const Corr*)corr, ws, (int)n, E, S, (int)h_count, thr, plan.use_order, cnt, s1, s2,
                              buckets, order, batch, st, plan.one_sided, plan.a_scale, plan.ws.units, plan.per_unit, opt.xcd_map != 0, true,
                              nullptr, nullptr, (hipEvent_t)opt.timing_before, (hipEvent_t)opt.timing_after, false, plan.ws};
    switch (plan.hpw) {
        case 1: return launch_filtered<1>(args);
        case 2: return launch_filtered<2>(args);
        default: return launch_filtered<4>(args);
    }
}

}  // namespace

namespace sfmhost {

int launch_large_score(const LargePass& p, LargeScore* folded_later) {
    if (folded_later != nullptr)   // (NULL: the scoring launches fold their ranges themselves)
        *folded_later = LargeScore{1, nullptr, nullptr};   // (stays so unless the matrix-pipe kernel split the points into ranges)
    const sfm_score_options opt = resolve_options(p.options);
    if (!valid_options(opt)) return fail(SFM_EINVAL, "sfm_ransac_pass_large: an option is out of range");
    if (p.h_count < 1 || p.n < 8 || p.n > 0x7FFFFFFF || p.h_count > 0x3FFFFFFF)
        return fail(SFM_EINVAL, "sfm_ransac_pass_large: sizes out of range");
    return score_sed_impl(p.corr, p.n, p.E, p.S, p.h_count, p.batch, p.thr, p.cnt, p.s1, p.s2, p.workspace, p.workspace_bytes, p.stream,
                          opt, p.select_state, folded_later, p.tables_ready);
}

int launch_large_setup(const LargePass& p, MatrixTables* t) {
    *t = MatrixTables{false, nullptr, 0, 0.0, nullptr, nullptr, nullptr, 0};
    const sfm_score_options opt = resolve_options(p.options);
    if (!valid_options(opt) || p.h_count < 1 || p.batch < 1 || p.n < 8 || p.n > 0x7FFFFFFF || p.h_count > 0x3FFFFFFF)
        return SFM_OK;   // (the scoring call reports it)
    ScorePlan plan;
    if (plan_score(p.n, p.h_count, p.batch, p.thr, opt, &plan) != SFM_OK || !plan.ws.matrix) return SFM_OK;
    using namespace matrixscore;
    const unsigned setup_blocks = grid_stride(p.n, 1024, kSetupBlocks);
    float4* partial = reinterpret_cast<float4*>(p.workspace + ws_points_offset(p.batch));
    int32_t* buckets = reinterpret_cast<int32_t*>(p.workspace + ws_buckets_offset(p.n, p.batch));
    t->matrix = true;
    t->partial = partial;
    t->partials = (int)setup_blocks;
    t->a_scale = scale_for(p.thr);
    t->hyp_table = reinterpret_cast<uint4*>(p.workspace + ws_matrix_hyp_offset(p.n, p.h_count, p.batch, plan.ws));
    t->fix = p.workspace + ws_matrix_fix_offset(p.n, p.h_count, p.batch, plan.ws);
    t->table = reinterpret_cast<uint4*>(p.workspace + ws_matrix_offset(p.n, p.h_count, p.batch, plan.ws));
    t->step_blocks = p.batch == 1 ? (int)((table_steps(p.n) + 3) / 4) : 0;
    hipLaunchKernelGGL(matrix_setup_kernel, dim3(setup_blocks, (unsigned)p.batch), dim3(256), 0, p.stream, (const Corr*)p.corr, (int)p.n,
                       t->a_scale, partial, buckets, p.cnt, (int)p.h_count, p.select_state);
    if (p.batch > 1) {   // the point tables of a batch: a launch of their own, four steps per 256-thread block
        const int step_blocks = (int)((table_steps(p.n) + 3) / 4);
        hipLaunchKernelGGL(matrix_tables_kernel, dim3((unsigned)step_blocks, (unsigned)p.batch), dim3(256), 0, p.stream, (const Corr*)p.corr,
                           (int)p.n, t->a_scale, (const float4*)partial, (int)setup_blocks, t->table, step_blocks, (const double*)nullptr, 0,
                           (uint4*)nullptr, (const int32_t*)nullptr, p.thr, (unsigned char*)nullptr);
    }
    return check_launch("matrix_setup_kernel");
}

}  // namespace sfmhost
