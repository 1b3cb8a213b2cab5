// Model selection (reference lib/ransac/ransac.py:75-86) and the winner's inlier mask (:70-76) as block-level device
// routines, shared by the stand-alone kernels (sfm_kernels.hip) and by the last block of a fused small pass
// (sfm_score.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_fit.h"
#include "sfm_math.h"

namespace sfmsel {

// "no model" key: above every finite non-negative double's bit pattern, and still positive when the
// record is viewed as int64 (so a cross-GPU MIN on int64 works).
constexpr uint64_t kNoModelKey = 0x7FFFFFFFFFFFFFFFull;

SFM_DEVICE uint64_t hypothesis_key(const int32_t* cnt, const double* s1, const double* s2, const int32_t* flags,
                                   int64_t h, double min_extra, int aggregation, bool& flagged) {
    const int ch = cnt[h];
    const double err = sfmfit::aggregate_error(aggregation, ch, s1[h], s2[h]);
    // Hypotheses whose sample was flagged degenerate never compete (the reference aborts on them).
    flagged = flags != nullptr && flags[h] != 0;
    // ransac.py:75 gate and :83 strict compare against an initial +inf: NaN and inf never win.
    const bool ok = ((double)ch >= min_extra) && (err < INFINITY) && !flagged;
    uint64_t bits = (uint64_t)__double_as_longlong(err);
    if (bits == 0x8000000000000000ull) bits = 0;  // -0.0 orders as +0.0
    return ok ? bits : kNoModelKey;
}

// Wave-level fold of (key, best index, first flagged index, flag count) with the rule of ransac.py:83-86 (lowest key,
// then lowest index): four DPP row rotations leave every lane of a 16-lane row with the row's result, the four rows
// are then read out with v_readlane and folded in row order.  All VALU: the __shfl_xor butterfly this replaces went
// through the LDS crossbar (ds_bpermute) six times in a dependent chain with seven words each — most of what a
// latency-bound selection block waited on.
template <int N>
SFM_DEVICE uint64_t dpp_ror_u64(uint64_t x) {
    const uint32_t lo = (uint32_t)sfm::dpp_row_ror<N>((int)(uint32_t)x), hi = (uint32_t)sfm::dpp_row_ror<N>((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
SFM_DEVICE uint64_t read_lane_u64(uint64_t x, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
SFM_DEVICE void wave_fold(uint64_t& key, int64_t& best, int64_t& first_flag, int& n_flag) {
    auto combine = [&](uint64_t ok, int64_t ob, int64_t of, int on) {
        if (ok < key || (ok == key && ob < best)) {
            key = ok;
            best = ob;
        }
        first_flag = of < first_flag ? of : first_flag;
        n_flag += on;
    };
#define SFM_FOLD_STEP(N)                                                                                       \
    combine(dpp_ror_u64<N>(key), (int64_t)dpp_ror_u64<N>((uint64_t)best), (int64_t)dpp_ror_u64<N>((uint64_t)first_flag), \
            sfm::dpp_row_ror<N>(n_flag))
    SFM_FOLD_STEP(8);
    SFM_FOLD_STEP(4);
    SFM_FOLD_STEP(2);
    SFM_FOLD_STEP(1);
#undef SFM_FOLD_STEP
    // rows 1..3 into row 0's result (every lane ends up with the wave's result)
    uint64_t k0 = read_lane_u64(key, 0);
    int64_t b0 = (int64_t)read_lane_u64((uint64_t)best, 0), f0 = (int64_t)read_lane_u64((uint64_t)first_flag, 0);
    int n0 = __builtin_amdgcn_readlane(n_flag, 0);
#pragma unroll
    for (int row = 1; row < 4; ++row) {
        const uint64_t kr = read_lane_u64(key, 16 * row);
        const int64_t br = (int64_t)read_lane_u64((uint64_t)best, 16 * row);
        const int64_t fr = (int64_t)read_lane_u64((uint64_t)first_flag, 16 * row);
        const int nr = __builtin_amdgcn_readlane(n_flag, 16 * row);
        if (kr < k0 || (kr == k0 && br < b0)) {
            k0 = kr;
            b0 = br;
        }
        f0 = fr < f0 ? fr : f0;
        n0 += nr;
    }
    key = k0;
    best = b0;
    first_flag = f0;
    n_flag = n0;
}

// Shared-memory scratch of block_select for a block of THREADS threads.
template <int THREADS>
struct SelectScratch {
    uint64_t key[THREADS / kWave];
    int64_t best[THREADS / kWave], first[THREADS / kWave];
    int flags[THREADS / kWave];
};

// Block-wide fold of per-thread candidates (key, best index, first flagged index, flag count) with the rule of
// ransac.py:83-86 (lowest key, then lowest index); on return thread 0 holds the block's result.
template <int THREADS>
__device__ __forceinline__ void block_combine(uint64_t& key, int64_t& best, int64_t& first_flag, int& n_flag,
                                              SelectScratch<THREADS>& sh) {
    auto combine = [&](uint64_t ok, int64_t ob, int64_t of, int on) {
        if (ok < key || (ok == key && ob < best)) {
            key = ok;
            best = ob;
        }
        first_flag = of < first_flag ? of : first_flag;
        n_flag += on;
    };
    wave_fold(key, best, first_flag, n_flag);
    const int wave = threadIdx.x / kWave;
    __syncthreads();  // the scratch may still be read from a previous fold
    if ((threadIdx.x & (kWave - 1)) == 0) {
        sh.key[wave] = key; sh.best[wave] = best; sh.first[wave] = first_flag; sh.flags[wave] = n_flag;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < THREADS / kWave; ++w) combine(sh.key[w], sh.best[w], sh.first[w], sh.flags[w]);
}

// The whole selection over h_count hypotheses by ONE block of THREADS threads: lexicographic minimum of (error bits,
// index) — lowest aggregated error among gated hypotheses, earliest index on ties — plus the flag statistics.
// Thread 0 writes the record.  Returns (to every thread) the winner's LOCAL index, or -1.
template <int THREADS>
__device__ __forceinline__ int64_t block_select(const int32_t* __restrict__ cnt, const double* __restrict__ s1,
                                                const double* __restrict__ s2, const int32_t* __restrict__ flags,
                                                int64_t h_count, int64_t h_offset, double min_extra, int aggregation,
                                                sfm_select_result* __restrict__ record, SelectScratch<THREADS>& sh,
                                                int64_t* sh_winner) {
    uint64_t key = kNoModelKey;
    int64_t best = INT64_MAX, first_flag = INT64_MAX;
    int n_flag = 0;
    // four hypotheses per trip with their loads issued together (the loop is pure load latency otherwise)
    for (int64_t h0 = threadIdx.x; h0 < h_count; h0 += 4 * THREADS) {  // increasing h: strict < keeps the earliest
        uint64_t k[4];
        bool flagged[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t h = h0 + u * THREADS;
            flagged[u] = false;
            k[u] = h < h_count ? hypothesis_key(cnt, s1, s2, flags, h, min_extra, aggregation, flagged[u]) : kNoModelKey;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t h = h0 + u * THREADS;
            if (k[u] < key) {
                key = k[u];
                best = h;
            }
            if (flagged[u]) {
                first_flag = h < first_flag ? h : first_flag;
                ++n_flag;
            }
        }
    }
    auto combine = [&](uint64_t ok, int64_t ob, int64_t of, int on) {
        if (ok < key || (ok == key && ob < best)) {
            key = ok;
            best = ob;
        }
        first_flag = of < first_flag ? of : first_flag;
        n_flag += on;
    };
    wave_fold(key, best, first_flag, n_flag);
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        sh.key[wave] = key; sh.best[wave] = best; sh.first[wave] = first_flag; sh.flags[wave] = n_flag;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < THREADS / kWave; ++w) combine(sh.key[w], sh.best[w], sh.first[w], sh.flags[w]);
        const bool found = key != kNoModelKey && best != INT64_MAX;
        sfm_select_result r;
        r.key = found ? key : kNoModelKey;
        r.best_h = found ? best + h_offset : -1;
        r.best_err = found ? __longlong_as_double((long long)key) : INFINITY;
        r.first_flagged = first_flag != INT64_MAX ? first_flag + h_offset : INT64_MAX;
        r.n_flagged = n_flag;
        r.best_cnt = found ? cnt[best] : 0;
        *record = r;
        *sh_winner = found ? best : -1;
    }
    __syncthreads();
    return *sh_winner;
}

// mask[i] = 2 for the 8 sample points of hypothesis h, 1 for the other points with sed <= thr, 0 otherwise; all zero
// for h outside [0, h_count).  Points i = first, first + stride, ... (a block- or grid-stride walk).
template <int UNROLL = 1>
__device__ __forceinline__ void write_inlier_mask(const Corr* __restrict__ pts, int64_t n, const double* __restrict__ E,
                                                  const int32_t* __restrict__ S, int64_t h_count, int64_t h, double thr,
                                                  uint8_t* __restrict__ out, int64_t first, int64_t stride) {
    if (h < 0 || h >= h_count) {
        for (int64_t i = first; i < n; i += stride) out[i] = 0;
        return;
    }
    double e[9];
    int32_t smp[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[h * 9 + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) smp[k] = S[h * 8 + k];
    // UNROLL points per trip with their loads issued together (a single block walking many points is load-latency bound)
    for (int64_t i0 = first; i0 < n; i0 += stride * UNROLL) {
        Corr p[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + u * stride;
            p[u] = pts[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n) {
                const double sed = sfm::sed_value(e, p[u].xa, p[u].ya, p[u].xb, p[u].yb);
                bool in_sample = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) in_sample |= (smp[k] == (int32_t)i);
                out[i] = in_sample ? 2 : ((sed <= thr) ? 1 : 0);
            }
        }
    }
}

}  // namespace sfmsel
