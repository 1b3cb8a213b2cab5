// Operand tables of the matrix-pipe scoring kernel (sfm_score_matrix.h) as device routines: the point side (prepare_step) and
// the hypothesis side (hypothesis_row / emit_hypothesis_half) with the hypotheses' sample corrections (sample_correction_half).
// Shared by the table kernels of sfm_score.hip and by the eight-point fit kernel of sfm_kernels.hip, whose lanes write their
// hypothesis' rows and correction themselves in a fused pass (E and the sample are in registers there).  The error bound these
// operands carry is derived in the header of sfm_score_matrix.h.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_math.h"
#include "sfm_score_ws.h"

namespace matrixscore {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;        // points per step
constexpr int kHyps = 32;        // hypotheses per wave
constexpr int kBlocks = 3;       // K16 operand blocks per step: r' slots 0..15 and 16..31 (fp16), denominator slots 0..15 (bf16)
constexpr double kKappa = 1.0 / 32.0;
constexpr int kPointTop = 14, kHypTop = 11;   // scaled magnitudes: point terms < 2^14, hypothesis entries < 2^11

__host__ __device__ inline int64_t steps_of(int64_t n) { return (n + kTile - 1) / kTile; }
// Steps the point operand table holds: those of the points, rounded up to a multiple of four with PAD steps.  A row past the
// last point carries zero operands and a NEGATIVE constant slot, so its "denominator" is negative whatever the hypothesis and the
// sign test rejects it by itself: the step loop needs neither a mask for the ragged last step nor a test for the steps a group
// of kAhead + 1 runs past the end of the points (two to four VALU instructions per step, in a loop that is bound by their issue).
__host__ __device__ inline int64_t table_steps(int64_t n) { return sfmws::matrix_table_steps(n); }
__host__ __device__ inline int64_t table_bytes(int64_t n) { return table_steps(n) * kBlocks * 64 * 16; }
__host__ __device__ inline int64_t hyp_table_bytes(int64_t h_count) { return h_count * 2 * kBlocks * 16; }

// factor carried by the prepared a-side coordinates for this kernel (host side)
inline double scale_for(double thr) {
    const double T = thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5);
    return (T > 1e-30 && T < 1e30) ? (1.0 - 1e-6) / sqrt(T * (1.0 + kKappa)) : 0.0;   // NaN compares false
}

SFM_DEVICE float bf_round(double x) { return (float)(__bf16)(float)x; }   // nearest (double rounding: < 2^-8 (1 + 2^-15))
SFM_DEVICE float bf_up(float x) {   // smallest bf16 >= x for x >= 0 (NaN stays NaN, inf stays inf)
    const float r = (float)(__bf16)x;
    if (!(r < x)) return r;
    return __uint_as_float(__float_as_uint(r) + 0x10000u);
}
// x (already scaled into fp16's range) = hi + mid + res
SFM_DEVICE void split2(double x, float& hi, float& mid) {
    hi = (float)(_Float16)(float)x;
    mid = (float)(_Float16)(float)(x - (double)hi);
#if SFM_MATRIX_ABLATE & 16   // measurement build: no fp16 subnormals among the operands (are they slow on the matrix pipe?)
    if (fabsf(mid) < 6.2e-5f) mid = 0.0f;
    if (fabsf(hi) < 6.2e-5f) hi = 0.0f;
#endif
    if (!(fabs(x) < 1e300)) mid = hi;   // inf / NaN: keep the poison in both parts (inf - inf would be NaN anyway)
}
// power of two s with s * x in [2^(top-1), 2^top) (x > 0 finite), else 1
SFM_DEVICE float scale_to(float x, int top) {
    if (!(x > 1e-30f) || !(x < 1e30f)) return 1.0f;
    int ex;
    (void)frexpf(x, &ex);   // x = f 2^ex, f in [0.5, 1)
    return ldexpf(1.0f, top - ex);
}
// Data-set side of the scaling, the same in the table kernel and in the scoring kernel: M_t >= |m_t| of every point (from the
// maxima score_prepare_kernel left; w >= the third coordinate of the scaled a) and the power of two s_p.  ok = false (then
// s_p = 1 and every hypothesis runs with its filter off): magnitudes for which s_p or the s_p^2 of the dB chain would leave
// fp32's comfortable range — no finite input may turn into an inf or NaN inside the filter, because the test reads the SIGN
// of dB - r^2.
struct DataScale {
    float sp;
    bool ok;
    float Q[12];   // data-set maxima of |q_k| (unscaled): the b-side monomials of dB, then the a-side ones of dA
};
SFM_DEVICE DataScale data_scale(const uint32_t* maxima, float w, float (&M)[9]) {
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f), Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f), Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    M[0] = Xb * Xa; M[1] = Xb * Ya; M[2] = Xb * w; M[3] = Yb * Xa; M[4] = Yb * Ya; M[5] = Yb * w; M[6] = Xa; M[7] = Ya; M[8] = w;
    float mmax = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) mmax = fmaxf(mmax, M[j] * (1.0f + 1e-6f));
    DataScale d;
    d.sp = scale_to(mmax, kPointTop);
    const float xa = w > 0.0f ? Xa / w : 0.0f, ya = w > 0.0f ? Ya / w : 0.0f;   // the a side of the denominator form is unscaled
    const float qmax = fmaxf(fmaxf(fmaxf(Xb * Xb, Yb * Yb), fmaxf(xa * xa, ya * ya)), 1.0f) * (d.sp * d.sp);   // largest |q_k| s_p^2
    d.ok = (mmax > 1e-12f) && (mmax < 1e12f) && (qmax < 1e30f) && (d.sp * d.sp > 1e-30f) && (w > 0.0f);
    if (!d.ok) d.sp = 1.0f;
    constexpr float up = 1.0f + 1e-5f;   // xa, ya are quotients of rounded values
    d.Q[0] = Xb * Xb; d.Q[1] = Xb * Yb; d.Q[2] = Yb * Yb; d.Q[3] = Xb; d.Q[4] = Yb; d.Q[5] = 1.0f;
    d.Q[6] = xa * xa * up; d.Q[7] = xa * ya * up; d.Q[8] = ya * ya * up; d.Q[9] = xa * up; d.Q[10] = ya * up; d.Q[11] = 1.0f;
    return d;
}

// slot tables: r' slot s = 3 t + v  (t = term 0..8, v = 0: m_hi E_hi, 1: m_hi E_mid, 2: m_mid E_hi), slots 27..31 zero;
// denominator slot s (one bf16 block): 0..5 qB_k gB_k / 4, 6..11 qA_k gA_k / 4, 12: s_p^2 * (slack + eps sum_k |g_k| Q_k) / s_p^2, 13..15 zero
SFM_DEVICE float point_slot_r(const float (&mh)[9], const float (&mm)[9], int s) {
    if (s >= 27) return 0.0f;
    return (s % 3 == 2) ? mm[s / 3] : mh[s / 3];
}
SFM_DEVICE float hyp_slot_r(const float (&eh)[9], const float (&em)[9], int s) {
    if (s >= 27) return 0.0f;
    return (s % 3 == 1) ? em[s / 3] : eh[s / 3];
}

// Operand table of the points: for step t (32 points), block b, lane l = 32 half + point: the 8 sixteen-bit values of slots
// 8 half .. 8 half + 7 of block b — one coalesced 1 KiB load per block and step.  `ws` holds the data-set maxima (of the
// coordinates scaled by c) that score_prepare_kernel left.  Rows past n (the ragged last step, the pad steps) have zero operands
// and a negative constant slot: the sign test rejects them under every hypothesis.
// Row of the A operand (0..31) -> point of the step it carries: the 32x32 result layout gives lane half H, register j the row
// (j & 3) + 8 (j >> 2) + 4 H; with this map that row holds point 16 H + j.
__host__ __device__ inline int point_of_row(int row) { return 16 * ((row >> 2) & 1) + 4 * (row >> 3) + (row & 3); }

SFM_DEVICE void prepare_step(const Corr* __restrict__ corr, int n, double c, const uint32_t* maxima, uint4* __restrict__ table,
                             int t, int l) {
    const int i = t * kTile + point_of_row(l & 31);
    const int half = l >> 5;
    float M[9];
    const DataScale data = data_scale(maxima, (float)c * (1.0f + 1e-6f), M);
    const double sp = (double)data.sp;
    float mh[9], mm[9], q[12];
#pragma unroll
    for (int j = 0; j < 9; ++j) mh[j] = mm[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 12; ++j) q[j] = 0.0f;
    if (i < n && data.ok) {   // (filter off for the data set: all-zero operands, so that no product can be a NaN)
        const Corr p = corr[i];
        const double xa = p.xa * c, ya = p.ya * c;
        const double m[9] = {p.xb * xa, p.xb * ya, p.xb * c, p.yb * xa, p.yb * ya, p.yb * c, xa, ya, c};
#pragma unroll
        for (int j = 0; j < 9; ++j) split2(m[j] * sp, mh[j], mm[j]);
        const double qq[12] = {p.xb * p.xb, p.xb * p.yb, p.yb * p.yb, p.xb, p.yb, 1.0,      // dB = lb0^2 + lb1^2, lb = E^T b
                               p.xa * p.xa, p.xa * p.ya, p.ya * p.ya, p.xa, p.ya, 1.0};     // dA = la0^2 + la1^2, la = E a
#pragma unroll
        for (int j = 0; j < 12; ++j) q[j] = (float)(qq[j] * (sp * sp));
    }
    // (the slot a lane stores depends on its half: BOTH candidates are named with compile-time indices and one is selected —
    // indexing the arrays with `8 * half + j` put them into scratch memory, 208 bytes a lane, and the kernel that writes the
    // point tables of a batch took 190 us for 246 MB)
    const bool upper = half != 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (_Float16)(upper ? point_slot_r(mh, mm, 16 * b + 8 + j) : point_slot_r(mh, mm, 16 * b + j));
        table[((size_t)t * kBlocks + b) * 64 + l] = __builtin_bit_cast(uint4, v);
    }
    const float inside_sign = (i < n) ? (float)(sp * sp) : -(float)(sp * sp);   // rows past the points: always rejected (a negative "denominator")
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lower_x = bf_round(q[j]);                                                    // slots 0..7
        const float upper_x = (8 + j < 12) ? bf_round(q[(8 + j < 12) ? 8 + j : 0]) : ((8 + j == 12) ? inside_sign : 0.0f);   // slots 8..15
        v[j] = (__bf16)(upper ? upper_x : lower_x);
    }
    table[((size_t)t * kBlocks + 2) * 64 + l] = __builtin_bit_cast(uint4, v);
}
// Operand table of the hypotheses: for hypothesis h and half (0: slots 0..7, 1: slots 8..15 of each block) the three B
// fragments of tier 1 — the scaled fp16 hi / mid split of E for the two r' blocks and the bf16 dB form with its absolute terms
// and the slack.  hypothesis_row computes what both halves share (the splits, the quadratic forms, the bound's constants),
// emit_hypothesis_half picks one half's 3 x 16 bytes out of it.
struct HypothesisRow {
    float eh[9], em[9];   // s_h E = hi + mid (fp16 values)
    float gs[12];         // bf16-rounded quadratic forms of (dA + dB) / 4, scaled by s_h^2
    float slot12;         // the constant slot: slack / s_p^2 + the bf16 rounding term, rounded up (inf: filter off)
    bool armed;
    float delta, slack, sh, sp, rounding;   // what the bound is made of (diagnostics)
};
SFM_DEVICE HypothesisRow hypothesis_row(const uint32_t* maxima, const double (&e)[9], double a_scale) {
    constexpr float up = 1.0f + 1e-5f;
    HypothesisRow row;
    float M[9];
    const DataScale data = data_scale(maxima, (float)a_scale * (1.0f + 1e-6f), M);
    const float sp = data.sp;
    float emax = 0.f, poison = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        emax = fmaxf(emax, fabsf((float)e[j]));
        poison += (float)e[j] * 0.0f;   // NaN anywhere in E must poison the bounds (fmaxf would drop it)
    }
    const float sh = scale_to(emax, kHypTop);
    float weighted = 0.f;   // sum_t |E_t| s_h M_t s_p
    float absolute = 0.f;   // sum_t (M_t s_p + |E_t| s_h): the subnormal part of the fp16 split (header of sfm_score_matrix.h)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        split2(e[j] * (double)sh, row.eh[j], row.em[j]);
        const float ej = fabsf((float)e[j]) * (1.0f + 1e-6f) * sh;
        weighted += ej * (M[j] * sp);
        absolute += ej + M[j] * sp;
    }
    const float delta = (5.2e-6f * weighted + 3.0101e-8f * absolute) * up + poison;   // 3.0101e-8 >= 1.01 * 2^-25
    const float slack = (delta * delta) * (float)((1.0 + kKappa) / kKappa) * up + poison;   // in scaled units
    // (dA + dB) / 4 as quadratic forms of b = (xb, yb, 1) and a = (xa, ya, 1): lb = E^T b (columns of E), la = E a (rows)
    const double g[12] = {0.25 * (e[0] * e[0] + e[1] * e[1]), 0.5 * (e[0] * e[3] + e[1] * e[4]), 0.25 * (e[3] * e[3] + e[4] * e[4]),
                          0.5 * (e[0] * e[6] + e[1] * e[7]), 0.5 * (e[3] * e[6] + e[4] * e[7]), 0.25 * (e[6] * e[6] + e[7] * e[7]),
                          0.25 * (e[0] * e[0] + e[3] * e[3]), 0.5 * (e[0] * e[1] + e[3] * e[4]), 0.25 * (e[1] * e[1] + e[4] * e[4]),
                          0.5 * (e[0] * e[2] + e[3] * e[5]), 0.5 * (e[1] * e[2] + e[4] * e[5]), 0.25 * (e[2] * e[2] + e[5] * e[5])};
    constexpr float eps = 0.008f;   // >= (1 + 2^-8)^2 - 1 + 40 * 2^-23 = 0.00783 with 2 % to spare
    float gmax = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) gmax = fmaxf(gmax, fabsf((float)g[j]));
    // filter off for this hypothesis (every operand zero, infinite slack: r'' = 0, dB'' = +inf, nothing rejected and
    // nothing that could turn into a NaN): magnitudes outside the scaled ranges, a NaN or inf entry, thr off
    row.armed = data.ok && (emax > 1e-12f) && (emax < 1e12f) && (gmax > 1e-20f) && (gmax < 1e24f) && (slack == slack) &&
                (slack < 1e30f) && (a_scale != 0.0);
    const double sh2 = (double)sh * (double)sh;
    // the bf16 roundings of the twelve products and the accumulation: eps sum_k |g_k| Q_k with the data-set maxima Q_k — a
    // constant per hypothesis that rides with the slack (per-point absolute terms would cost a second operand block:
    // 128 instead of 96 bytes per point, measured as the bigger loss)
    float rounding = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) rounding += fabsf((float)(g[j] * sh2)) * (1.0f + 1e-6f) * (data.Q[j] * (1.0f + 1e-6f));
    rounding *= eps * up;   // in units of s_h^2 (the point side multiplies by s_p^2)
#pragma unroll
    for (int j = 0; j < 12; ++j) row.gs[j] = bf_round(g[j] * sh2);
    row.slot12 = row.armed ? bf_up((slack / (sp * sp) + rounding) * up) : INFINITY;   // the point side carries s_p^2 in this slot
    row.delta = delta;
    row.slack = slack;
    row.sh = sh;
    row.sp = sp;
    row.rounding = rounding;
    return row;
}
SFM_DEVICE void emit_hypothesis_half(const HypothesisRow& row, int half, uint4 (&out)[kBlocks]) {
    f16x8 B0, B1;
    bf16x8 B2;
    const bool upper = half != 0;   // (compile-time array indices, one select per slot: see prepare_step)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b0 = upper ? hyp_slot_r(row.eh, row.em, 8 + j) : hyp_slot_r(row.eh, row.em, j);
        const float b1 = upper ? hyp_slot_r(row.eh, row.em, 24 + j) : hyp_slot_r(row.eh, row.em, 16 + j);
        B0[j] = (_Float16)(row.armed ? b0 : 0.0f);
        B1[j] = (_Float16)(row.armed ? b1 : 0.0f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lower_x = row.armed ? row.gs[j] : 0.0f;                                                                   // slots 0..7
        const float upper_x = (8 + j < 12) ? (row.armed ? row.gs[(8 + j < 12) ? 8 + j : 0] : 0.0f) : ((8 + j == 12) ? row.slot12 : 0.0f);   // slots 8..15
        B2[j] = (__bf16)(upper ? upper_x : lower_x);
    }
    out[0] = __builtin_bit_cast(uint4, B0);
    out[1] = __builtin_bit_cast(uint4, B1);
    out[2] = __builtin_bit_cast(uint4, B2);
}

// The SAMPLE CORRECTION of a hypothesis: the eight sample points are never counted and always summed (ransac.py:70-79), while
// the scoring scan treats them like any other point — so for each sample point with sed <= thr the count drops by one, and the
// others add their sed / sed^2 to the sums.  One half = four sample points, summed in sample order; the hypothesis' correction
// is first half + second half: fix = [h_pad] int32 | [h_pad] f64 | [h_pad] f64.
SFM_DEVICE void sample_correction_half(const double (&e)[9], const Corr (&p)[4], double thr, int& dc, double& d1, double& d2) {
    dc = 0;
    d1 = 0.0;
    d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sed = sfm::sed_value(e, p[k].xa, p[k].ya, p[k].xb, p[k].yb);
        const bool counted = sed <= thr;            // the scan counted it and summed it
        dc += counted ? -1 : 0;
        const double extra = counted ? 0.0 : sed;   // NaN / inf propagate: such a model never wins
        d1 += extra;
        d2 += extra * extra;
    }
}

// One thread per (hypothesis, half) — the table kernels of sfm_score.hip.
SFM_DEVICE void prepare_hypothesis(const uint32_t* maxima, const double* __restrict__ E, int h_count, double a_scale,
                                   uint4* __restrict__ hyp_table, float* __restrict__ bound_out, const Corr* __restrict__ pts,
                                   const int32_t* __restrict__ S, double thr, unsigned char* __restrict__ fix, int64_t item_raw) {
    const bool live = item_raw < 2 * (int64_t)h_count;
    const int64_t item = live ? item_raw : 2 * (int64_t)h_count - 1;   // (the tail threads shadow the last one: the pair sums below are wave operations)
    const int64_t h = item >> 1;
    const int half = (int)(item & 1);
    double e[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) e[j] = E[(int64_t)h * 9 + j];
    if (fix != nullptr) {
        const int4 sample = *reinterpret_cast<const int4*>(S + h * 8 + 4 * half);
        const Corr p[4] = {pts[sample.x], pts[sample.y], pts[sample.z], pts[sample.w]};
        int dc;
        double d1, d2;
        sample_correction_half(e, p, thr, dc, d1, d2);
        const int dc_other = __shfl_xor(dc, 1, 64);
        const double d1_other = __shfl_xor(d1, 1, 64), d2_other = __shfl_xor(d2, 1, 64);
        if (live && half == 0) {
            const int64_t hp = sfmws::split_padded(h_count);
            reinterpret_cast<int32_t*>(fix)[h] = dc + dc_other;
            reinterpret_cast<double*>(fix + 4 * hp)[h] = d1 + d1_other;
            reinterpret_cast<double*>(fix + 4 * hp)[hp + h] = d2 + d2_other;
        }
    }
    if (!live) return;
    const HypothesisRow row = hypothesis_row(maxima, e, a_scale);
    uint4 out[kBlocks];
    emit_hypothesis_half(row, half, out);
    if (bound_out != nullptr && half == 0) {   // diagnostic (sfm_debug_matrix_filter): what the bound of this hypothesis is made of
        float* b = bound_out + h * 8;
        b[0] = row.delta;
        b[1] = row.slack;
        b[2] = row.sh;
        b[3] = row.armed ? 1.0f : 0.0f;
        b[4] = row.sp;
        b[5] = row.rounding;
        b[6] = row.slot12;
        b[7] = 0.0f;
    }
    uint4* dst = hyp_table + item * kBlocks;
    dst[0] = out[0];
    dst[1] = out[1];
    dst[2] = out[2];
}

// The same for a lane that holds its hypothesis' E and sample already (the eight-point fit kernel of a fused pass): both halves'
// rows and the whole correction, in the order of the per-half form above — bit for bit what prepare_hypothesis writes.
SFM_DEVICE void prepare_hypothesis_lane(const uint32_t* maxima, const double (&e)[9], const int32_t (&sample)[8], int64_t h, int h_count,
                                        double a_scale, double thr, const Corr* __restrict__ pts, uint4* __restrict__ hyp_table,
                                        unsigned char* __restrict__ fix, bool active) {
    Corr p0[4], p1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p0[k] = pts[sample[k]];
        p1[k] = pts[sample[4 + k]];
    }
    int dc0, dc1;
    double a0, a1, b0, b1;
    sample_correction_half(e, p0, thr, dc0, a0, b0);
    sample_correction_half(e, p1, thr, dc1, a1, b1);
    const HypothesisRow row = hypothesis_row(maxima, e, a_scale);
    uint4 lo[kBlocks], hi[kBlocks];
    emit_hypothesis_half(row, 0, lo);
    emit_hypothesis_half(row, 1, hi);
    if (!active) return;
    const int64_t hp = sfmws::split_padded(h_count);
    reinterpret_cast<int32_t*>(fix)[h] = dc0 + dc1;
    reinterpret_cast<double*>(fix + 4 * hp)[h] = a0 + a1;
    reinterpret_cast<double*>(fix + 4 * hp)[hp + h] = b0 + b1;
    uint4* dst = hyp_table + h * 2 * kBlocks;
#pragma unroll
    for (int b = 0; b < kBlocks; ++b) {
        dst[b] = lo[b];
        dst[kBlocks + b] = hi[b];
    }
}

// Fold of the partial coordinate maxima matrix_setup_kernel left (one float4 per block of points) by ONE wave: every lane ends
// up with the data-set maxima as the bit patterns data_scale reads.  (max is order-independent: the same values whoever folds.)
SFM_DEVICE void fold_partial_maxima(const float4* __restrict__ partial, int partials, int lane, uint32_t (&maxima)[4]) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    for (int i = lane; i < partials; i += kWave) {
        const float4 m = partial[i];
        m0 = fmaxf(m0, m.x);
        m1 = fmaxf(m1, m.y);
        m2 = fmaxf(m2, m.z);
        m3 = fmaxf(m3, m.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 = fmaxf(m0, __shfl_xor(m0, off, 64));
        m1 = fmaxf(m1, __shfl_xor(m1, off, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
        m3 = fmaxf(m3, __shfl_xor(m3, off, 64));
    }
    maxima[0] = __float_as_uint(m0);
    maxima[1] = __float_as_uint(m1);
    maxima[2] = __float_as_uint(m2);
    maxima[3] = __float_as_uint(m3);
}

}  // namespace matrixscore
