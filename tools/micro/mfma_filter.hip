// Experiment (round 2): can the idle matrix pipe take the tier-1 filter of the scoring kernel?
//
// Tier 1 is a dense contraction: r'[h,i] = sum_k E^[h,k] phi_k(i) over phi = (xb xa', xb ya', xb c, yb xa', yb ya', yb c,
// xa', ya', c) — an [H x 9].[9 x N] fp32 GEMM — and the one-sided denominator dB[h,i] = lb0^2 + lb1^2 expands into a
// quadratic form q[h,.] . psi(i), psi = (xb^2, xb yb, yb^2, xb, yb, 1): an [H x 6].[6 x N] GEMM.  v_mfma_f32_32x32x2_f32 is
// exact fp32 (fmaf-chain semantics, MI355X_MICROARCH.md) and issues on a pipe that runs beside the VALU.
//
// This harness isolates tier 1 (no compaction ring, no fp64 tier): both kernels count, per hypothesis, the points the
// filter does NOT reject.
//   valu_filter_count   the production test (reject_mask_one_sided of sfm_score.hip: 12 VALU per evaluation), 4
//                       hypotheses per wave, 128 points per step
//   mfma_filter_count   32 hypotheses per wave, 32 points per tile: 5 MFMAs (K = 10, one zero pad) for r', 3 for dB,
//                       then one multiply + one compare per accumulator register
// Conservativeness of the MFMA form (it must never reject a pair whose fp64 SED is <= thr) is checked by the driver
// (tools/time_mfma_filter.py) against the exact fp64 kernel: survivors >= inliers for every hypothesis, and the
// per-point masks of a sample of hypotheses contain the exact inlier masks.
//
// Bound for the MFMA form (u = 2^-24, k = 2^-10): each term E^_k phi^_k carries one rounding of E_k, one of phi_k (computed
// in fp64, rounded once) and at most 9 accumulation roundings, so |r32 - c r_fl| <= delta' := 16 u Emax M (M as in
// sfm_score.hip, from the data-set maxima of the scaled coordinates).  dB32 = chain of 6 fmaf over q^_k psi^_k + cb with
// |dB32 - cb - dB| <= 9 u (b0^2 + b1^2) =: eta (b_j = |e_j0| Xb + |e_j1| Yb + |e_j2| bound |lb_j|, and sum |q_k| |psi_k| <=
// b0^2 + b1^2); with cb := 12 u (b0^2 + b1^2) + delta'^2 (1 + k) / k the accumulated value is an upper bound of the fp64
// denominator plus the delta slack, and the production derivation (r'^2 > dB + delta'^2 (1+k)/k  =>  sed_fl > thr) applies.
#include "../../structure_from_motion_amd/csrc/sfm_score.hip"

namespace sfmhost {
char* error_buffer() {  // defined in sfm_kernels.hip for the product library
    static thread_local char buffer[kErrorBytes];
    return buffer;
}
}  // namespace sfmhost

namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;          // points per MFMA tile (N of 32x32x2)
constexpr int kRowsPerWave = 32;   // hypotheses per wave (M of 32x32x2)
constexpr int kSteps = 8;          // MFMAs per tile: 5 for r' (K = 10), 3 for dB (K = 6)

// B-operand table: for tile t, step m, lane l: value[(t * kSteps + m) * 64 + l] = operand k = 2 m' + l / 32 of point
// t * 32 + l % 32 (m' = m for the phi steps, m - 5 for the psi steps): one coalesced 256-byte load per MFMA.
__global__ void mfma_prepare_kernel(const Corr* __restrict__ corr, int n, double c, float* __restrict__ table) {
    const int t = blockIdx.x;
    const int l = threadIdx.x;  // 64 threads
    const int i = t * kTile + (l & 31);
    const int half = l >> 5;
    double phi[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, psi[6] = {0, 0, 0, 0, 0, 0};
    if (i < n) {
        const Corr p = corr[i];
        const double xa = p.xa * c, ya = p.ya * c;
        phi[0] = p.xb * xa; phi[1] = p.xb * ya; phi[2] = p.xb * c;
        phi[3] = p.yb * xa; phi[4] = p.yb * ya; phi[5] = p.yb * c;
        phi[6] = xa; phi[7] = ya; phi[8] = c; phi[9] = 0.0;
        psi[0] = p.xb * p.xb; psi[1] = p.xb * p.yb; psi[2] = p.yb * p.yb; psi[3] = p.xb; psi[4] = p.yb; psi[5] = 1.0;
    }
    float* out = table + (size_t)t * kSteps * 64 + l;
#pragma unroll
    for (int m = 0; m < 5; ++m) out[m * 64] = (float)phi[2 * m + half];
#pragma unroll
    for (int m = 0; m < 3; ++m) out[(5 + m) * 64] = (float)psi[2 * m + half];
}

template <bool MASKS, bool MFMA_ONLY = false>
__global__ __launch_bounds__(256) void mfma_filter_count_kernel(const float* __restrict__ table, const unsigned char* __restrict__ ws,
                                                                int n, const double* __restrict__ E, int h_count, double a_scale,
                                                                int32_t* __restrict__ survivors, unsigned long long* __restrict__ masks,
                                                                int mask_hyps) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int h0 = wave * kRowsPerWave;
    if (h0 >= h_count) return;
    const int row = lane & 31, half = lane >> 5;
    const int h = min(h0 + row, h_count - 1);
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws);
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f), Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f), Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    constexpr float u = 5.9604644775390625e-08f, up = 1.0f + 1e-5f, kappa = 1.0f / 1024.0f;
    double e[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) e[j] = E[(int64_t)h * 9 + j];
    float ef[10];
    float emax = 0.f, poison = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        ef[j] = (float)e[j];
        emax = fmaxf(emax, fabsf(ef[j]));
        poison += ef[j] * 0.0f;
    }
    ef[9] = 0.0f;
    const float w = (float)a_scale * (1.0f + 1e-6f);
    const float M = (Xa + Ya + w) * (Xb + Yb + 1.0f);
    const float delta = (16.0f * u) * emax * M * up + poison;
    const float b0 = fabsf(ef[0]) * Xb + fabsf(ef[3]) * Yb + fabsf(ef[6]);
    const float b1 = fabsf(ef[1]) * Xb + fabsf(ef[4]) * Yb + fabsf(ef[7]);
    const float bb = (b0 * b0 + b1 * b1) * up;
    float cb = (12.0f * u) * bb * up + (delta * delta) * ((1.0f + kappa) / kappa * up) + poison;
    if (!(bb > 1e-30f) || !(cb == cb) || a_scale == 0.0) cb = INFINITY;  // filter off for this hypothesis
    // quadratic form of dB in fp64 from the fp64 E, rounded once
    float q[6];
    q[0] = (float)(e[0] * e[0] + e[1] * e[1]);
    q[1] = (float)(2.0 * (e[0] * e[3] + e[1] * e[4]));
    q[2] = (float)(e[3] * e[3] + e[4] * e[4]);
    q[3] = (float)(2.0 * (e[0] * e[6] + e[1] * e[7]));
    q[4] = (float)(2.0 * (e[3] * e[6] + e[4] * e[7]));
    q[5] = (float)(e[6] * e[6] + e[7] * e[7]) + cb;  // psi_5 = 1: the slack rides in the constant term
    float A[kSteps];
#pragma unroll
    for (int m = 0; m < 5; ++m) A[m] = half ? ef[2 * m + 1] : ef[2 * m];
#pragma unroll
    for (int m = 0; m < 3; ++m) A[5 + m] = half ? q[2 * m + 1] : q[2 * m];

    // survivors are counted per lane (lane = one point column, 16 hypothesis rows); the production kernel would take
    // the compare's lane mask for its ring push instead, so this costs one VALU more per register than it would there
    int count[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) count[j] = 0;
    const int tiles = (n + kTile - 1) / kTile;
    const float* __restrict__ src = table + lane;
    float B[kSteps];
#pragma unroll
    for (int m = 0; m < kSteps; ++m) B[m] = src[m * 64];
    for (int t = 0; t < tiles; ++t) {
        float Bn[kSteps];
        const float* nxt = src + (size_t)min(t + 1, tiles - 1) * kSteps * 64;
#pragma unroll
        for (int m = 0; m < kSteps; ++m) Bn[m] = nxt[m * 64];
        float16v r = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d = r;
#pragma unroll
        for (int m = 0; m < 5; ++m) r = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m], B[m], r, 0, 0, 0);
#pragma unroll
        for (int m = 5; m < 8; ++m) d = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m], B[m], d, 0, 0, 0);
        if (MFMA_ONLY) {  // ablation: the matrix work alone (one compare per tile keeps the accumulators alive)
            count[0] += (r[0] + r[5] + r[10] + r[15] > d[0] + d[5] + d[10] + d[15]) ? 0 : 1;
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) count[j] += (r[j] * r[j] > d[j]) ? 0 : 1;
        }
        if (MASKS) {  // validation mode: per-(hypothesis, tile) survivor masks of the first mask_hyps hypotheses
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned long long keep = ~__builtin_amdgcn_ballot_w64(r[j] * r[j] > d[j]);
                const int r0 = 8 * (j / 4) + (j % 4);
                if (lane == 0) {
                    if (h0 + r0 < mask_hyps) masks[(size_t)(h0 + r0) * tiles + t] = keep & 0xffffffffull;
                    if (h0 + r0 + 4 < mask_hyps) masks[(size_t)(h0 + r0 + 4) * tiles + t] = keep >> 32;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < kSteps; ++m) B[m] = Bn[m];
    }
    const int padded = tiles * kTile - n;  // zero columns of the last tile are never rejected
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int total = count[j];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);  // within each half of the wave
        const int hyp = h0 + 8 * (j / 4) + 4 * half + (j % 4);
        if (row == 0 && hyp < h_count) survivors[hyp] = total - padded;
    }
}

// The production tier-1 test with nothing behind it: 4 hypotheses per wave, 128 points per step, survivors counted.
__global__ __launch_bounds__(256, 5) void valu_filter_count_kernel(const unsigned char* __restrict__ ws, int n,
                                                                   const double* __restrict__ E, int h_count, double thr,
                                                                   double a_scale, int32_t* __restrict__ survivors) {
    constexpr int HPW = 4;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h0 = wave * HPW;
    if (h0 >= h_count) return;
    const uint32_t* maxima = reinterpret_cast<const uint32_t*>(ws);
    const float4* __restrict__ pts32 = reinterpret_cast<const float4*>(ws + ws_points_offset(1));
    const float Xa = __uint_as_float(maxima[0]) * (1.0f + 1e-6f), Ya = __uint_as_float(maxima[1]) * (1.0f + 1e-6f);
    const float Xb = __uint_as_float(maxima[2]) * (1.0f + 1e-6f), Yb = __uint_as_float(maxima[3]) * (1.0f + 1e-6f);
    FilterConsts f[HPW];
    int c[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int h = min(h0 + k, h_count - 1);
        double e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) e[j] = E[(int64_t)h * 9 + j];
        f[k] = make_filter_consts(e, Xa, Ya, Xb, Yb, a_scale);
        arm_one_sided(f[k]);
        f[k].e[0] = uniform(f[k].e[0]); f[k].e[1] = uniform(f[k].e[1]);
        f[k].e[3] = uniform(f[k].e[3]); f[k].e[4] = uniform(f[k].e[4]);
        c[k] = 0;
    }
    const int pairs = n / 128;
    const float4* __restrict__ next = pts32 + lane;
    float4 p0 = next[0], p1 = next[64];
    for (int pr = 0; pr < pairs; ++pr) {
        next += 128;
        const float4 q0 = next[0], q1 = next[64];  // the last step reads into the workspace pad
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const unsigned long long m0 = ~reject_mask_one_sided(f[k], p0.x, p0.y, p0.z, p0.w);
            const unsigned long long m1 = ~reject_mask_one_sided(f[k], p1.x, p1.y, p1.z, p1.w);
            c[k] += (int)__popcll(m0) + (int)__popcll(m1);
        }
        p0 = q0; p1 = q1;
    }
    for (int i = pairs * 128 + lane; i < n + 63; i += 64) {
        const bool ok = i < n;
        const float4 p = pts32[min(i, n - 1)];
#pragma unroll
        for (int k = 0; k < HPW; ++k)
            c[k] += (int)__popcll(~reject_mask_one_sided(f[k], p.x, p.y, p.z, p.w) & __builtin_amdgcn_ballot_w64(ok));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < HPW; ++k)
            if (h0 + k < h_count) survivors[h0 + k] = c[k];
    }
}

}  // namespace

#define EXPORT extern "C" __attribute__((visibility("default")))

EXPORT int64_t mfma_table_floats(int64_t n) { return ((n + kTile - 1) / kTile) * kSteps * 64; }

// ws: a score workspace (sfm_score_workspace_bytes(n, h, 1)); fills its fp32 points + maxima and the MFMA operand table
EXPORT int filter_prepare(const double* corr, int64_t n, double thr, void* ws, float* table, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const double c = one_sided_scale(thr);
    hipLaunchKernelGGL(score_prepare_kernel, dim3(1, 1), dim3(256), 0, st, (const Corr*)corr, n, c, (unsigned char*)ws);
    hipLaunchKernelGGL(mfma_prepare_kernel, dim3((unsigned)((n + kTile - 1) / kTile)), dim3(64), 0, st, (const Corr*)corr, (int)n, c,
                       table);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

EXPORT int valu_filter_count(const void* ws, int64_t n, const double* E, int64_t h, double thr, int32_t* survivors, void* stream) {
    const int64_t waves = (h + 3) / 4;
    hipLaunchKernelGGL(valu_filter_count_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)ws, (int)n, E, (int)h, thr, one_sided_scale(thr), survivors);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

EXPORT int mfma_filter_count(const float* table, const void* ws, int64_t n, const double* E, int64_t h, double thr,
                             int32_t* survivors, unsigned long long* masks, int64_t mask_hyps, void* stream) {
    const int64_t waves = (h + kRowsPerWave - 1) / kRowsPerWave;
    if (masks != nullptr)
        hipLaunchKernelGGL(mfma_filter_count_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           table, (const unsigned char*)ws, (int)n, E, (int)h, one_sided_scale(thr), survivors, masks,
                           (int)mask_hyps);
    else if (mask_hyps == -1)  // ablation switch: MFMAs only
        hipLaunchKernelGGL((mfma_filter_count_kernel<false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, table, (const unsigned char*)ws, (int)n, E, (int)h, one_sided_scale(thr),
                           survivors, masks, 0);
    else
        hipLaunchKernelGGL(mfma_filter_count_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           table, (const unsigned char*)ws, (int)n, E, (int)h, one_sided_scale(thr), survivors, masks,
                           (int)mask_hyps);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
