// Experiment (round 3): tier 1 of the scoring kernel on the 16-bit matrix pipe.
//
// Round 2 put the filter's two contractions on v_mfma_f32_32x32x2_f32 (tools/micro/mfma_filter.hip) and lost: fp32 MFMA
// runs at the VALU rate.  The 16-bit MFMAs run at 16 x that rate, and a value is the sum of two fp16 values to 2^-22:
//   r'[i,h] = sum_t m_t(i) E_t(h),  m = (xb xa', xb ya', xb c, yb xa', yb ya', yb c, xa', ya', c)   (9 terms)
//           ~ sum_t  m_hi E_hi + m_hi E_mid + m_mid E_hi                                           (27 products, K = 32, fp16)
//   dB[i,h] = sum_k q_k(i) g_k(h),  q = (xb^2, xb yb, yb^2, xb, yb, 1),  g = the quadratic form of lb0^2 + lb1^2 in E
//           <= sum_k bf(q_k) bf(g_k) + sum_k up(|q_k|) up(eps |g_k|) + slack(h)                     (13 products, K = 16, bf16)
// i.e. two v_mfma_f32_32x32x16_f16 for r' and one v_mfma_f32_32x32x16_bf16 for an UPPER bound of dB per 32 points x 32
// hypotheses (96 matrix cycles per 1024 evaluations; the VALU tier 1 spends 12 instructions per 64).  Points are the A
// operand (rows), hypotheses the B operand (columns): a lane then holds results of ONE hypothesis (column lane & 31) for
// 16 of the 32 points, which is the layout a lane-per-hypothesis exact tier would consume.
//
// Ranges.  fp16 holds 2^-14 .. 2^16 at full precision, so both sides of the r' chain are scaled by exact powers of two: the
// point terms by s_p (data set: the largest term maximum lands in [2^13, 2^14)), the hypothesis entries by s_h (per
// hypothesis: the largest entry in [2^10, 2^11)); the dB chain (bf16: fp32's range) carries s_p^2 and s_h^2, so the compare
// r''^2 > dB'' is the unscaled one.
//
// (This harness keeps the first form — one-sided dB with per-point absolute terms; the production kernel, csrc/sfm_score_matrix.h,
// moved on to the harmonic-mean denominator (dA + dB) / 4.  bf16 has 8 significant bits: unit roundoff 2^-8.)
// Bound (in scaled units; u16 = 2^-11).  x = hi + mid + res, |res| <= 2^-22 |x|, or 2^-15 absolute where mid would be
// subnormal (taken as flushed).  Dropped products per term: mid*mid, res*x, x*res <= 3.1 * 2^-22 |m_t E_t| + the flush terms
// (<= 2^-15 * 2^14 + 2^-15 * 2^11 per term: 0.6 against sum_t M_t |E_t| >= 2^23, i.e. < 2^-23 relative).  Accumulation: every
// product of two fp16 is exact in fp32; the probe below shows the matrix unit truncating each aligned addend at the unit
// of the largest one, i.e. an error of at most one ulp (2^-23 relative) of the largest term per addend: <= 17 * 2^-23 sum |products|
// per instruction, 34 * 2^-23 for the chain of two.  Together  |r''_mfma - r''| <= delta'' = 5.2e-6 sum_t |E_t| s_h M_t s_p
// (M_t: data-set maximum of |m_t|).  With (x - d)^2 >= x^2 / (1 + k) - d^2 / k:   r''^2 > dB''_up + delta''^2 (1 + k) / k  =>
// c^2 r^2 >= dB / (1 + k)  =>  r^2 / dB >= T  when  c^2 (1 + k) T <= 1.  The slack rides in the constant slot of the dB chain.
#include "../../structure_from_motion_amd/csrc/sfm_score.hip"

namespace sfmhost {
char* error_buffer() {  // defined in sfm_kernels.hip for the product library
    static thread_local char buffer[kErrorBytes];
    return buffer;
}
}  // namespace sfmhost

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;      // points per step
constexpr int kHyps = 32;      // hypotheses per wave
constexpr int kBlocks = 3;     // K16 blocks per step: r' slots 0..15, r' slots 16..31, dB slots 0..15
constexpr double kKappa = 1.0 / 32.0;

SFM_DEVICE float bf_round(double x) { return (float)(__bf16)(float)x; }            // nearest (double rounding: < 2^-8 (1 + 2^-15))
SFM_DEVICE float bf_up(float x) {   // smallest bf16 >= x, x >= 0 (NaN stays NaN, inf stays inf)
    const float r = (float)(__bf16)x;
    if (!(r < x)) return r;
    return __uint_as_float(__float_as_uint(r) + 0x10000u);
}
// x (already scaled into fp16's range) = hi + mid + res, hi and mid fp16 values, |res| <= 2^-22 |x| (+ 2^-25 if mid is subnormal)
SFM_DEVICE void split2(double x, float& hi, float& mid) {
    hi = (float)(_Float16)(float)x;
    mid = (float)(_Float16)(float)(x - (double)hi);
    if (!(fabs(x) < 1e300)) mid = hi;   // inf / NaN: keep the poison in both parts (inf - inf would be NaN anyway)
}
// power of two s with s * x in [2^(top-1), 2^top) (x > 0 finite), else 1
SFM_DEVICE float scale_to(float x, int top) {
    if (!(x > 1e-30f) || !(x < 1e30f)) return 1.0f;
    int ex;
    (void)frexpf(x, &ex);   // x = f 2^ex, f in [0.5, 1)
    return ldexpf(1.0f, top - ex);
}
SFM_DEVICE void term_maxima(float Xa, float Ya, float Xb, float Yb, float w, float (&M)[9]) {
    M[0] = Xb * Xa; M[1] = Xb * Ya; M[2] = Xb * w; M[3] = Yb * Xa; M[4] = Yb * Ya; M[5] = Yb * w; M[6] = Xa; M[7] = Ya; M[8] = w;
}
constexpr int kPointTop = 14, kHypTop = 11;   // scaled magnitudes: point terms < 2^14, hypothesis entries < 2^11

// slot tables: r' slot s = 3 t + v  (t = term 0..8, v = 0: m_hi E_hi, 1: m_hi E_mid, 2: m_mid E_hi), slots 27..31 zero;
// dB slot s: 0..5 q_k g_k, 6..11 |q_k| (eps |g_k|), 12: 1 * slack, 13..15 zero
SFM_DEVICE float point_slot_r(const float (&mh)[9], const float (&mm)[9], int s) {
    if (s >= 27) return 0.0f;
    return (s % 3 == 2) ? mm[s / 3] : mh[s / 3];
}
SFM_DEVICE float hyp_slot_r(const float (&eh)[9], const float (&em)[9], int s) {
    if (s >= 27) return 0.0f;
    return (s % 3 == 1) ? em[s / 3] : eh[s / 3];
}

// Operand table of the points: for step t (32 points), block b, lane l = 32 half + point: the 8 sixteen-bit values of
// slots 8 half .. 8 half + 7 of block b — one coalesced 1 KiB load per block and step.  ws holds the data-set maxima.
__global__ void bf16_prepare_kernel(const Corr* __restrict__ corr, int n, double c, const unsigned char* __restrict__ ws,
                                    uint4* __restrict__ table) {
    const int t = blockIdx.x;
    const int l = threadIdx.x;  // 64 threads
    const int i = t * kTile + (l & 31);
    const int half = l >> 5;
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws);
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f), Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f), Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    float M[9], mmax = 0.f;
    term_maxima(Xa, Ya, Xb, Yb, (float)c * (1.0f + 1e-6f), M);
#pragma unroll
    for (int j = 0; j < 9; ++j) mmax = fmaxf(mmax, M[j]);
    const double sp = (double)scale_to(mmax, kPointTop);
    float mh[9], mm[9], q[6];
#pragma unroll
    for (int j = 0; j < 9; ++j) mh[j] = mm[j] = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) q[j] = 0.0f;
    if (i < n) {
        const Corr p = corr[i];
        const double xa = p.xa * c, ya = p.ya * c;
        const double m[9] = {p.xb * xa, p.xb * ya, p.xb * c, p.yb * xa, p.yb * ya, p.yb * c, xa, ya, c};
#pragma unroll
        for (int j = 0; j < 9; ++j) split2(m[j] * sp, mh[j], mm[j]);
        const double qq[6] = {p.xb * p.xb, p.xb * p.yb, p.yb * p.yb, p.xb, p.yb, 1.0};
#pragma unroll
        for (int j = 0; j < 6; ++j) q[j] = (float)(qq[j] * (sp * sp));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)point_slot_r(mh, mm, 16 * b + 8 * half + j);
        table[((size_t)t * kBlocks + b) * 64 + l] = __builtin_bit_cast(uint4, v);
    }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = 8 * half + j;
        float x = 0.0f;
        if (s < 6) x = bf_round(q[s]);
        else if (s < 12) x = bf_up(fabsf(q[s - 6]) * (1.0f + 1e-6f));
        else if (s == 12) x = (i < n) ? (float)(sp * sp) : 0.0f;
        v[j] = (__bf16)x;
    }
    table[((size_t)t * kBlocks + 2) * 64 + l] = __builtin_bit_cast(uint4, v);
}

template <bool MASKS, int MODE = 0>   // MODE 1: matrix work alone
__global__ __launch_bounds__(256) void bf16_filter_count_kernel(const uint4* __restrict__ table, const unsigned char* __restrict__ ws,
                                                                int n, const double* __restrict__ E, int h_count, double a_scale,
                                                                int32_t* __restrict__ survivors, unsigned* __restrict__ masks,
                                                                int mask_hyps) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int h0 = wave * kHyps;
    if (h0 >= h_count) return;
    const int col = lane & 31, half = lane >> 5;
    const int h = min(h0 + col, h_count - 1);
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws);
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f), Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f), Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    constexpr float up = 1.0f + 1e-5f;
    float M[9], mmax = 0.f;
    term_maxima(Xa, Ya, Xb, Yb, (float)a_scale * (1.0f + 1e-6f), M);
#pragma unroll
    for (int j = 0; j < 9; ++j) mmax = fmaxf(mmax, M[j]);
    const float sp = scale_to(mmax, kPointTop);
    double e[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) e[j] = E[(int64_t)h * 9 + j];
    float emax = 0.f, poison = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        emax = fmaxf(emax, fabsf((float)e[j]));
        poison += (float)e[j] * 0.0f;
    }
    const float sh = scale_to(emax, kHypTop);
    float eh[9], em[9];
    float weighted = 0.f;   // sum_t |E_t| s_h M_t s_p
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        split2(e[j] * (double)sh, eh[j], em[j]);
        weighted += fabsf((float)e[j]) * sh * (M[j] * sp);
    }
    const float delta = 5.2e-6f * weighted * up + poison;
    const float ss = (sp * sh) * (sp * sh);
    float slack = (delta * delta) * (float)((1.0 + kKappa) / kKappa) * up + poison;   // in scaled units already
    // quadratic form of dB in fp64 from the fp64 E
    const double g[6] = {e[0] * e[0] + e[1] * e[1], 2.0 * (e[0] * e[3] + e[1] * e[4]), e[3] * e[3] + e[4] * e[4],
                         2.0 * (e[0] * e[6] + e[1] * e[7]), 2.0 * (e[3] * e[6] + e[4] * e[7]), e[6] * e[6] + e[7] * e[7]};
    constexpr float eps = 0.008f;   // >= (1 + 2^-8)^2 - 1 + 20 * 2^-23 (bf16: 8 significant bits, unit roundoff 2^-8)
    float gmax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) gmax = fmaxf(gmax, fabsf((float)g[j]));
    // filter off for this hypothesis: magnitudes outside the range the scaling handles, NaN, thr off
    const bool armed = (emax > 1e-15f) && (emax < 1e15f) && (mmax < 1e15f) && (mmax > 1e-15f) && (gmax > 1e-20f) && (gmax < 1e30f) &&
                       (slack == slack) && (a_scale != 0.0);
    // B operand (hypotheses): lane holds slots 8 half .. 8 half + 7 of each block for column `col`
    f16x8 B0, B1;
    bf16x8 B2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        B0[j] = (_Float16)hyp_slot_r(eh, em, 8 * half + j);
        B1[j] = (_Float16)hyp_slot_r(eh, em, 16 + 8 * half + j);
    }
    const double sh2 = (double)sh * (double)sh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = 8 * half + j;
        float x = 0.0f;
        if (s < 6) x = bf_round(g[s] * sh2);
        else if (s < 12) x = bf_up(fabsf((float)(g[s - 6] * sh2)) * eps * up);
        else if (s == 12) x = armed ? bf_up(slack / (sp * sp) * up) : INFINITY;   // the point side carries s_p^2 in this slot
        B2[j] = (__bf16)x;
    }
    (void)ss;

    unsigned count = 0;
    const int tiles = (n + kTile - 1) / kTile;
    const uint4* __restrict__ src = table + lane;
    uint4 A[kBlocks];
#pragma unroll
    for (int b = 0; b < kBlocks; ++b) A[b] = src[b * 64];
    for (int t = 0; t < tiles; ++t) {
        uint4 An[kBlocks];
        const uint4* nxt = src + (size_t)min(t + 1, tiles - 1) * kBlocks * 64;
#pragma unroll
        for (int b = 0; b < kBlocks; ++b) An[b] = nxt[b * 64];
        float16v r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d = r;
        r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), B0, r, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[1]), B1, r, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[2]), B2, d, 0, 0, 0);
        unsigned keep = 0;   // bit j: point row(j, half) of this step survives under this lane's hypothesis
        if (MODE == 1) {
            keep = (r[0] + r[5] + r[10] + r[15] > d[0] + d[5] + d[10] + d[15]) ? 0u : 1u;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) keep |= (r[j] * r[j] > d[j]) ? 0u : (1u << j);
        }
        count += __popc(keep);
        if (MASKS && h0 + col < mask_hyps) {
            // spread the 16 bits to their rows: row(j, half) = (j & 3) + 8 (j >> 2) + 4 half
            unsigned spread = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) spread |= ((keep >> j) & 1u) << ((j & 3) + 8 * (j >> 2) + 4 * half);
            atomicOr(masks + (size_t)(h0 + col) * tiles + t, spread);
        }
#pragma unroll
        for (int b = 0; b < kBlocks; ++b) A[b] = An[b];
    }
    const int padded = tiles * kTile - n;  // zero rows of the last tile are never rejected (r' = 0, dB = 0)
    const unsigned total = count + __shfl_xor(count, 32, 64);
    if (half == 0 && h0 + col < h_count) survivors[h0 + col] = (int)total - padded;
}

// What one v_mfma_f32_32x32x16_bf16 does with its 16 products and the carried sum: row 0 x column 0 of the result.
template <bool HALF>
__global__ void probe_kernel(const float* __restrict__ a, const float* __restrict__ b, float c_in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    bf16x8 A, B;
    f16x8 Ah, Bh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 5) + j;
        A[j] = (__bf16)((lane & 31) == 0 ? a[k] : 0.0f);
        B[j] = (__bf16)((lane & 31) == 0 ? b[k] : 0.0f);
        Ah[j] = (_Float16)((lane & 31) == 0 ? a[k] : 0.0f);
        Bh[j] = (_Float16)((lane & 31) == 0 ? b[k] : 0.0f);
    }
    float16v c;
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = c_in;
    if (HALF) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

}  // namespace

#define EXPORT extern "C" __attribute__((visibility("default")))

EXPORT int64_t bf16_table_bytes(int64_t n) { return ((n + kTile - 1) / kTile) * kBlocks * 64 * 16; }
EXPORT double bf16_filter_scale(double thr) {
    const double T = thr * (1.0 + 1.0 / 1024.0) * (1.0 + 1e-5);
    return (T > 1e-30 && T < 1e30) ? (1.0 - 1e-6) / sqrt(T * (1.0 + kKappa)) : 0.0;
}

// ws: a score workspace; fills its fp32 points + maxima (scaled by this filter's c) and the bf16 operand table
EXPORT int bf16_prepare(const double* corr, int64_t n, double thr, void* ws, void* table, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const double c = bf16_filter_scale(thr);
    hipLaunchKernelGGL(score_prepare_kernel, dim3(1, 1), dim3(256), 0, st, (const Corr*)corr, n, c, (unsigned char*)ws);
    hipLaunchKernelGGL(bf16_prepare_kernel, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(64), 0, st, (const Corr*)corr, (int)n, c,
                       (const unsigned char*)ws, (uint4*)table);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

EXPORT int bf16_filter_count(const void* table, const void* ws, int64_t n, const double* E, int64_t h, double thr,
                             int32_t* survivors, unsigned* masks, int64_t mask_hyps, void* stream) {
    const int64_t waves = (h + kHyps - 1) / kHyps;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const double c = bf16_filter_scale(thr);
    if (masks != nullptr)
        hipLaunchKernelGGL(bf16_filter_count_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)table,
                           (const unsigned char*)ws, (int)n, E, (int)h, c, survivors, masks, (int)mask_hyps);
    else if (mask_hyps == -1)
        hipLaunchKernelGGL((bf16_filter_count_kernel<false, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)table,
                           (const unsigned char*)ws, (int)n, E, (int)h, c, survivors, masks, 0);
    else
        hipLaunchKernelGGL(bf16_filter_count_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)table,
                           (const unsigned char*)ws, (int)n, E, (int)h, c, survivors, masks, 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

EXPORT int bf16_probe(const float* a, const float* b, float c_in, float* out, int half, void* stream) {
    if (half) hipLaunchKernelGGL(probe_kernel<true>, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c_in, out);
    else hipLaunchKernelGGL(probe_kernel<false>, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c_in, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
