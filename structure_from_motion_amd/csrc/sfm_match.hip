// Brute-force window matching on the GPU (reference lib/feature_matching/matching.py:36-81 with
// ncc.py:7-54 / ssd.py:7-36 as the score function and util.py:8-27 for the windows).
//
//   patch_extract_kernel   one lane per feature: bounds test, window gather, (NCC) mean removal and sum of
//                          squares.  Patches are stored k-major  P[k][feature]  so that the score kernel's
//                          loads are coalesced across features.
//   pair_scores_kernel     |A| x |B| scores, 128x128 tile per 256-thread block, 8x8 outputs per lane, window
//                          chunks staged through LDS.  The window sum runs over k = 0..K-1 in order with
//                          separate multiply / add roundings (no FMA), so every score is bit-identical to the
//                          oracle's sequential evaluation whatever the tiling.  fp64 VALU bound (2K flop per
//                          pair); MFMA is deliberately not used: its fused accumulation would change the bits.
//   row_summary_kernel     one wave per A-feature scans its row in B order and reproduces what the reference's
//                          per-feature heapq holds at positions 0 and 1 after pushing the scores in order:
//                          heap[0] = first minimum; heap[1] (the LEFT child, which the ratio test divides by —
//                          not necessarily the second best) = min over pushes i landing in the left subtree of
//                          max(score_i, running minimum before i).
//   pair_summary_kernel +  the product path: the same tile core, but each block reduces its tile to four numbers
//   summary_combine_kernel per row and a second small kernel walks them in B order — the |A| x |B| matrix is never
//                          written (3.2 GB at 20k x 20k) nor re-read, with bit-identical summaries.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_math.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;

constexpr int kRowsPerLane = 8;                 // A-features per lane
constexpr int kColGroups = 4;                   // B-features per lane: kColGroups groups of 2 adjacent columns
constexpr int kColsPerLane = 2 * kColGroups;
constexpr int kTileA = 16 * kRowsPerLane;       // 128 A-features per 256-thread block (16 x 16 lanes)
constexpr int kTileB = 32 * kColGroups;         // 128 B-features per block
constexpr int kChunk = 32;                      // window elements staged per LDS pass

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
    if (subtract_mean == SFM_PATCH_RAW64) {
        // integer images (ssd.py:31-35 in the image dtype): the pixels are int64 bit patterns, copied as they are — no
        // floating-point instruction may touch them
        const unsigned long long* raw = reinterpret_cast<const unsigned long long*>(image);
        unsigned long long* out = reinterpret_cast<unsigned long long*>(patches);
        int k = 0;
        for (int r = 0; r < side; ++r)
            for (int c = 0; c < side; ++c) out[(int64_t)(k++) * stride + i] = raw[(y0 + r) * width + (x0 + c)];
        ssq[i] = 0.0;
        ok[i] = 1;
        return;
    }
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

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2) by linear id.
// With a plain (B tile, A tile) grid every XCD touches every B tile for every A row: 2.1 GB of fetches at
// 20k x 20k for 26 MB of patches.  Here the 1-D block id is decoded so that each residue class mod 8 — one XCD —
// owns a contiguous band of A tiles and walks the B tiles with its band as the inner loop: a B tile (83 KB) serves
// the whole band while it is L2-resident, and the band's A tiles (~1.7 MB) stay resident throughout.
struct TileCoord {
    int64_t a_tile, b_tile;
    bool valid;
};
SFM_DEVICE TileCoord decode_tile(int64_t tiles_a, int64_t tiles_b) {
    const int64_t band = (tiles_a + 7) / 8;  // A tiles per XCD
    const int64_t label = blockIdx.x & 7, j = blockIdx.x >> 3;
    TileCoord t;
    t.a_tile = label * band + j % band;
    t.b_tile = j / band;
    t.valid = t.a_tile < tiles_a && t.b_tile < tiles_b;
    return t;
}
inline unsigned tile_grid(int64_t tiles_a, int64_t tiles_b) { return (unsigned)(8 * ((tiles_a + 7) / 8) * tiles_b); }

// Tile core shared by the score-matrix kernel and the fused summary kernel.  A 256-thread block owns a
// kTileA x kTileB tile; lane (ty, tx) accumulates rows a0 + ty*8 + i (i < 8) against columns
// b0 + g*32 + tx*2 + j (g < 4, j < 2): its B operands are four 16-byte LDS reads at consecutive-lane addresses
// (conflict-free), its A operands two 32-byte broadcasts.  16 LDS doubles feed 64 multiply + 64 add per window
// element (the 4x4 tile this replaces needed 8 for 16 + 16).  The window sum runs over k = 0..K-1 in order with
// separate multiply / add roundings, so the result does not depend on the tiling.
// MODE 0: NCC numerator sum a*b (patches mean-removed); MODE 1: SSD sum (a-b)^2.
static_assert(kTileA == 2 * kWave && kTileB == 2 * kWave, "one LDS-DMA instruction (64 x 16 B) = one tile row");
// 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS starting at `row` (wave-uniform)
SFM_DEVICE void direct_row(const double* src, double* row) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)row, 16, 0, 0);
}

// MODEs of the tile core.  0: NCC numerator sum a*b (patches mean-removed); 1: SSD sum (a-b)^2 in float64;
// 2 / 3 / 4: SSD of INTEGER images in the image dtype's modular arithmetic (ssd.py:31-35: `diff` and `np.square(diff)`
// wrap modulo 2^bits — into the signed range for signed types —, np.sum accumulates in int64 / uint64) on patches that
// hold the pixels as int64 bit patterns (SFM_PATCH_RAW64):
//   2  signed 8 / 16-bit types, 3  unsigned 8 / 16-bit types: 32-bit arithmetic (subtract, bit-field extract, 24-bit
//      multiply, bit-field extract, add — five full-rate integer instructions per pair and window element), int32
//      accumulators: windows of at most kNarrowMaxWindow elements
//   4  any width (32- and 64-bit types; narrow types with huge windows): 64-bit arithmetic, int64 accumulators
constexpr int kModeNcc = 0, kModeSsd = 1, kModeSsdNarrowSigned = 2, kModeSsdNarrowUnsigned = 3, kModeSsdWide = 4;
constexpr int kNarrowMaxWindow = 32768;   // 32768 terms of magnitude < 2^16 fit an int32 / uint32 accumulator
struct IntKind {
    int bits;        // 8, 16, 32, 64
    int is_signed;
};

template <bool DIRECT>
SFM_DEVICE void stage_chunk(const double* __restrict__ Pa, int64_t stride_a, const double* __restrict__ Pb, int64_t stride_b,
                            int64_t nA, int64_t nB, int k0, int kc, int64_t a0, int64_t b0, double (*sA)[kTileA],
                            double (*sB)[kTileB]) {
    const int tid = threadIdx.x;
    if constexpr (DIRECT) {
        // LDS-DMA staging: one global_load_lds_dwordx4 per window row and matrix moves 64 lanes x 16 B = the
        // 128 features of the row straight into sA[kk] / sB[kk] — no staging registers (the kernel sits at the
        // two-waves-per-SIMD register limit with its 8x8 accumulators), all rows of the chunk in flight at once.
        // Needs 16-byte aligned rows padded to a multiple of 128 features (checked by the launcher).
        const int wave = tid / kWave, lane = tid % kWave;
        for (int kk = wave; kk < kc; kk += 256 / kWave) {
            direct_row(Pa + (int64_t)(k0 + kk) * stride_a + a0 + 2 * lane, &sA[kk][0]);
            direct_row(Pb + (int64_t)(k0 + kk) * stride_b + b0 + 2 * lane, &sB[kk][0]);
        }
    } else {
        for (int idx = tid; idx < kc * kTileA; idx += 256) {
            const int kk = idx / kTileA, f = idx % kTileA;
            const int64_t ia = a0 + f;
            sA[kk][f] = ia < nA ? Pa[(int64_t)(k0 + kk) * stride_a + ia] : 0.0;   // (0.0 is also the integer 0)
        }
        for (int idx = tid; idx < kc * kTileB; idx += 256) {
            const int kk = idx / kTileB, f = idx % kTileB;
            const int64_t ib = b0 + f;
            sB[kk][f] = ib < nB ? Pb[(int64_t)(k0 + kk) * stride_b + ib] : 0.0;
        }
    }
}

template <int MODE, bool DIRECT>
SFM_DEVICE void tile_accumulate(const double* __restrict__ Pa, int64_t stride_a, const double* __restrict__ Pb,
                                int64_t stride_b, int64_t nA, int64_t nB, int K, int64_t a0, int64_t b0,
                                double (*sA)[kTileA], double (*sB)[kTileB],
                                double (&acc)[kRowsPerLane][kColsPerLane], IntKind kind) {
    const int tid = threadIdx.x;
    const int ty = tid / 16, tx = tid % 16;
    if constexpr (MODE == kModeNcc || MODE == kModeSsd) {
#pragma unroll
        for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j) acc[i][j] = 0.0;
        for (int k0 = 0; k0 < K; k0 += kChunk) {
            const int kc = min(kChunk, K - k0);
            __syncthreads();
            stage_chunk<DIRECT>(Pa, stride_a, Pb, stride_b, nA, nB, k0, kc, a0, b0, sA, sB);
            __syncthreads();
            for (int kk = 0; kk < kc; ++kk) {
                double av[kRowsPerLane], bv[kColsPerLane];
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i) av[i] = sA[kk][ty * kRowsPerLane + i];
#pragma unroll
                for (int g = 0; g < kColGroups; ++g) {
                    bv[2 * g] = sB[kk][g * 32 + tx * 2];
                    bv[2 * g + 1] = sB[kk][g * 32 + tx * 2 + 1];
                }
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
                    for (int j = 0; j < kColsPerLane; ++j) {
                        if (MODE == kModeNcc) {
                            acc[i][j] += av[i] * bv[j];
                        } else {
                            const double d = av[i] - bv[j];
                            acc[i][j] += d * d;
                        }
                    }
            }
        }
    } else if constexpr (MODE == kModeSsdNarrowSigned || MODE == kModeSsdNarrowUnsigned) {
        // 8 / 16-bit pixels: everything fits 32-bit registers — the low dword of the staged int64 IS the pixel (two's
        // complement), the difference is reduced to `bits` by one bit-field extract (sign- or zero-extending: the wrap of the
        // image dtype), its square (|d| < 2^16: a 24-bit multiply, exact in 32 bits) is reduced the same way, and the sum of at
        // most kNarrowMaxWindow such terms cannot leave 32 bits.
        constexpr bool kSigned = MODE == kModeSsdNarrowSigned;
        const unsigned bits = (unsigned)kind.bits;
        int total[kRowsPerLane][kColsPerLane];
#pragma unroll
        for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j) total[i][j] = 0;
        auto narrow = [&](int x) __attribute__((always_inline)) {
            return kSigned ? __builtin_amdgcn_sbfe(x, 0u, bits) : (int)__builtin_amdgcn_ubfe((unsigned)x, 0u, bits);
        };
        for (int k0 = 0; k0 < K; k0 += kChunk) {
            const int kc = min(kChunk, K - k0);
            __syncthreads();
            stage_chunk<DIRECT>(Pa, stride_a, Pb, stride_b, nA, nB, k0, kc, a0, b0, sA, sB);
            __syncthreads();
            for (int kk = 0; kk < kc; ++kk) {
                int av[kRowsPerLane], bv[kColsPerLane];
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i) av[i] = __double2loint(sA[kk][ty * kRowsPerLane + i]);
#pragma unroll
                for (int g = 0; g < kColGroups; ++g) {
                    bv[2 * g] = __double2loint(sB[kk][g * 32 + tx * 2]);
                    bv[2 * g + 1] = __double2loint(sB[kk][g * 32 + tx * 2 + 1]);
                }
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
                    for (int j = 0; j < kColsPerLane; ++j) {
                        const int d = narrow(av[i] - bv[j]);
                        const int sq = kSigned ? __mul24(d, d) : (int)__umul24((unsigned)d, (unsigned)d);
                        total[i][j] += narrow(sq);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j) acc[i][j] = kSigned ? (double)total[i][j] : (double)(unsigned)total[i][j];
    } else {
        // any width: 64-bit two's-complement arithmetic wraps modulo 2^64 by itself; a reduction to `bits` is a shift up and an
        // arithmetic (signed) or logical (unsigned) shift back down.  The int64 / uint64 accumulator of np.sum wraps the same way.
        const unsigned sh = 64u - (unsigned)kind.bits;
        const bool is_signed = kind.is_signed != 0;
        unsigned long long total[kRowsPerLane][kColsPerLane];
#pragma unroll
        for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j) total[i][j] = 0ull;
        auto narrow = [&](unsigned long long x) __attribute__((always_inline)) {
            const unsigned long long up = x << sh;
            return is_signed ? (unsigned long long)((long long)up >> sh) : up >> sh;
        };
        for (int k0 = 0; k0 < K; k0 += kChunk) {
            const int kc = min(kChunk, K - k0);
            __syncthreads();
            stage_chunk<DIRECT>(Pa, stride_a, Pb, stride_b, nA, nB, k0, kc, a0, b0, sA, sB);
            __syncthreads();
            for (int kk = 0; kk < kc; ++kk) {
                unsigned long long av[kRowsPerLane], bv[kColsPerLane];
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i) av[i] = (unsigned long long)__double_as_longlong(sA[kk][ty * kRowsPerLane + i]);
#pragma unroll
                for (int g = 0; g < kColGroups; ++g) {
                    bv[2 * g] = (unsigned long long)__double_as_longlong(sB[kk][g * 32 + tx * 2]);
                    bv[2 * g + 1] = (unsigned long long)__double_as_longlong(sB[kk][g * 32 + tx * 2 + 1]);
                }
#pragma unroll
                for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
                    for (int j = 0; j < kColsPerLane; ++j) {
                        const unsigned long long d = narrow(av[i] - bv[j]);
                        total[i][j] += narrow(d * d);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < kRowsPerLane; ++i)
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j)
                acc[i][j] = is_signed ? (double)(long long)total[i][j] : (double)total[i][j];
    }
}

// Window sum -> score.  NCC: (num / sqrt(qa qb)) * -1 + 1, 2.0 if a window is out of the image or the denominator is zero
// (ncc.py:24-54).  Every SSD mode: sum / K in float64 (for integer images: the int64 / uint64 sum converted to float64,
// ssd.py:36), +inf if a window is out of the image (ssd.py:24-29).
template <int MODE>
SFM_DEVICE double finish_score(double acc, bool inside, double qa, double qb, int K) {
    if (MODE == kModeNcc) {
        const double den = sqrt(qa * qb);
        return (!inside || den == 0.0) ? 2.0 : (acc / den) * -1.0 + 1.0;
    }
    return inside ? acc / (double)K : INFINITY;
}

template <int MODE, bool DIRECT>
__global__ __launch_bounds__(256, 2) void pair_scores_kernel(
    const double* __restrict__ Pa, int64_t stride_a, const double* __restrict__ Pb, int64_t stride_b,
    const double* __restrict__ qa, const double* __restrict__ qb, const uint8_t* __restrict__ oka,
    const uint8_t* __restrict__ okb, int64_t nA, int64_t nB, int K, double* __restrict__ scores, IntKind kind) {
    __shared__ double sA[kChunk][kTileA];
    __shared__ double sB[kChunk][kTileB];
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    const TileCoord tile = decode_tile((nA + kTileA - 1) / kTileA, (nB + kTileB - 1) / kTileB);
    if (!tile.valid) return;  // whole block
    const int64_t a0 = tile.a_tile * kTileA, b0 = tile.b_tile * kTileB;
    double acc[kRowsPerLane][kColsPerLane];
    tile_accumulate<MODE, DIRECT>(Pa, stride_a, Pb, stride_b, nA, nB, K, a0, b0, sA, sB, acc, kind);
#pragma unroll
    for (int i = 0; i < kRowsPerLane; ++i) {
        const int64_t ia = a0 + ty * kRowsPerLane + i;
        if (ia >= nA) continue;
        const double qai = qa[ia];
        const bool oki = oka[ia] != 0;
#pragma unroll
        for (int j = 0; j < kColsPerLane; ++j) {
            const int64_t ib = b0 + (j / 2) * 32 + tx * 2 + (j % 2);
            if (ib >= nB) continue;
            scores[ia * nB + ib] = finish_score<MODE>(acc[i][j], oki && okb[ib] != 0, qai, qb[ib], K);
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

// ---- fused path: scores never leave the chip ---------------------------------------------------------------
// The heap summary of a row is a scan in B order, but it splits over column tiles: with c = minimum of everything
// before the tile and p_i = minimum of the tile's own columns before i,
//     max(v_i, min(c, p_i)) = min(max(v_i, p_i), max(v_i, c)),   and   min_i max(v_i, c) = max(c, min_i v_i),
// so per (row, tile) four numbers suffice: M = tile minimum, its first column, A = min over left-subtree columns of
// max(v_i, p_i), B = min over left-subtree columns of v_i.  All of it is selection (min / max / compare), no
// arithmetic: the result is bit-identical to row_summary_kernel on the full matrix.  32 B per 128 scores leave
// the kernel instead of 1 KiB, and the matrix is never re-read.
// 16-lane row rotations by DPP (a VALU move modifier: no LDS crossbar, unlike __shfl): lane l of a row reads lane
// (l + N) mod 16 of the same row.
template <int N>
SFM_DEVICE int row_ror(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0x120 | N, 0xf, 0xf, false);
}
template <int N>
SFM_DEVICE double row_ror(double x) {
    return __hiloint2double(row_ror<N>(__double2hiint(x)), row_ror<N>(__double2loint(x)));
}

// Smallest value with its first column, and the second smallest (counting duplicates) of a set of scores.
struct TopTwo {
    double m1;
    int32_t at;  // column of m1 (first occurrence), INT32_MAX if the set has no comparable element
    double m2;
};
SFM_DEVICE void top_two_add(TopTwo& t, double v, int32_t column) {  // columns arrive in increasing order; NaN is ignored
    const bool first = v < t.m1;
    t.m2 = first ? t.m1 : ((v < t.m2) ? v : t.m2);
    t.at = first ? column : t.at;
    t.m1 = first ? v : t.m1;
}
SFM_DEVICE TopTwo top_two_merge(const TopTwo& a, const TopTwo& b) {
    const bool take_b = (b.m1 < a.m1) || (b.m1 == a.m1 && b.at < a.at);
    TopTwo r;
    r.m1 = take_b ? b.m1 : a.m1;
    r.at = take_b ? b.at : a.at;
    r.m2 = fmin(fmin(a.m2, b.m2), take_b ? a.m1 : b.m1);
    return r;
}
template <int N>
SFM_DEVICE TopTwo top_two_ror(const TopTwo& t) {
    TopTwo r;
    r.m1 = row_ror<N>(t.m1);
    r.at = row_ror<N>(t.at);
    r.m2 = row_ror<N>(t.m2);
    return r;
}

struct TileSummary {
    double tile_min, left_prefixed, left_min;
    int64_t tile_arg;
};
static_assert(sizeof(TileSummary) == 32, "workspace sizing in sfm_match_summary_workspace_bytes");

template <int MODE, bool DIRECT>
__global__ __launch_bounds__(256, 2) void pair_summary_kernel(
    const double* __restrict__ Pa, int64_t stride_a, const double* __restrict__ Pb, int64_t stride_b,
    const double* __restrict__ qa, const double* __restrict__ qb, const uint8_t* __restrict__ oka,
    const uint8_t* __restrict__ okb, int64_t nA, int64_t nB, int K, TileSummary* __restrict__ tiles, IntKind kind) {
    __shared__ double sA[kChunk][kTileA];
    __shared__ double sB[kChunk][kTileB];
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    const TileCoord tile = decode_tile((nA + kTileA - 1) / kTileA, (nB + kTileB - 1) / kTileB);
    if (!tile.valid) return;  // whole block
    const int64_t a0 = tile.a_tile * kTileA, b0 = tile.b_tile * kTileB;
    double acc[kRowsPerLane][kColsPerLane];
    tile_accumulate<MODE, DIRECT>(Pa, stride_a, Pb, stride_b, nA, nB, K, a0, b0, sA, sB, acc, kind);
    // Per-column and per-row operands of this lane's 8 x 8 block, fetched in one batch: unconditional loads at
    // clamped indices (a predicated load per element compiles to load -> wait -> next load, 32 memory latencies in a
    // row at two waves per SIMD — it was most of the kernel's fixed 1.5 ms).
    double qbj[kColsPerLane], qai_all[kRowsPerLane];
    uint8_t okb_raw[kColsPerLane], oka_raw[kRowsPerLane];
#pragma unroll
    for (int j = 0; j < kColsPerLane; ++j) {
        const int64_t ib = min(b0 + (j / 2) * 32 + tx * 2 + (j % 2), nB - 1);
        qbj[j] = qb[ib];
        okb_raw[j] = okb[ib];
    }
#pragma unroll
    for (int i = 0; i < kRowsPerLane; ++i) {
        const int64_t ia = min(a0 + ty * kRowsPerLane + i, nA - 1);
        qai_all[i] = qa[ia];
        oka_raw[i] = oka[ia];
    }
    bool okj[kColsPerLane], validj[kColsPerLane], leftj[kColsPerLane];
#pragma unroll
    for (int j = 0; j < kColsPerLane; ++j) {
        const int64_t ib = b0 + (j / 2) * 32 + tx * 2 + (j % 2);
        validj[j] = ib < nB;
        okj[j] = validj[j] && okb_raw[j] != 0;
        leftj[j] = validj[j] && ib >= 1 && in_left_subtree(ib + 1);
    }
    const bool uniform_tile = b0 >= 256;                      // see the fast path below
    const bool first_left = in_left_subtree(b0 + 1);          // heap position of the tile's first column (and 126 more)
    const bool last_left = in_left_subtree(b0 + kTileB);      // ... of its last column
    const bool last_valid = b0 + kTileB - 1 < nB;
#pragma unroll
    for (int i = 0; i < kRowsPerLane; ++i) {
        const int64_t ia = a0 + ty * kRowsPerLane + i;
        const bool row_valid = ia < nA;  // uniform over the 16 lanes of the row
        const double qai = qai_all[i];
        const bool oki = row_valid && oka_raw[i] != 0;
        if (uniform_tile) {  // block-uniform
            // From column 256 on, a 128-column tile lies inside one run of the heap's left/right pattern except,
            // possibly, for its LAST column (runs end at positions that are multiples of 128).  With F = the first
            // 127 columns and p_i the minimum of the tile's columns before i:
            //   min over i in F of max(v_i, p_i) = second smallest element of F (duplicates counted) — before the
            //   first minimum every term is >= the prefix minimum in front of it, at it the term IS that prefix
            //   minimum, after it the term is v_i itself —
            // so no scan is needed, only a top-two reduction (selection only: bit-identical to the general path).
            TopTwo t = {INFINITY, INT32_MAX, INFINITY};
            double v_last = INFINITY;  // the tile's last column, meaningful in lane tx == 15 only
#pragma unroll
            for (int j = 0; j < kColsPerLane; ++j) {
                const double v = finish_score<MODE>(acc[i][j], oki && okj[j], qai, qbj[j], K);
                const int32_t column = (int32_t)(b0 + (j / 2) * 32 + tx * 2 + (j % 2));
                const bool is_last = (j == kColsPerLane - 1) && (tx == 15);
                if (is_last) v_last = validj[j] ? v : INFINITY;
                if (validj[j] && !is_last) top_two_add(t, v, column);
            }
            t = top_two_merge(t, top_two_ror<8>(t));
            t = top_two_merge(t, top_two_ror<4>(t));
            t = top_two_merge(t, top_two_ror<2>(t));
            t = top_two_merge(t, top_two_ror<1>(t));
            if (tx == 15 && row_valid) {
                TileSummary r;
                const bool last_wins = v_last < t.m1;  // strict: an equal earlier column keeps the minimum
                r.tile_min = last_wins ? v_last : t.m1;
                r.tile_arg = last_wins ? (b0 + kTileB - 1) : (t.at == INT32_MAX ? INT64_MAX : (int64_t)t.at);
                double left_prefixed = INFINITY, left_min = INFINITY;
                if (first_left) {
                    left_prefixed = t.m2;
                    left_min = t.m1;
                }
                if (last_left && last_valid) {
                    left_prefixed = fmin(left_prefixed, (v_last < t.m1) ? t.m1 : v_last);  // p_last = min of F
                    left_min = fmin(left_min, v_last);
                }
                r.left_prefixed = left_prefixed;
                r.left_min = left_min;
                tiles[tile.b_tile * nA + ia] = r;
            }
            continue;
        }
        double carry = INFINITY;  // minimum of the tile's columns before the current group
        double left_prefixed = INFINITY, left_min = INFINITY;
        MinAt top = {INFINITY, INT64_MAX};
#pragma unroll
        for (int g = 0; g < kColGroups; ++g) {
            const int64_t c0 = b0 + g * 32 + tx * 2;
            const double v0 = finish_score<MODE>(acc[i][2 * g], oki && okj[2 * g], qai, qbj[2 * g], K);
            const double v1 = finish_score<MODE>(acc[i][2 * g + 1], oki && okj[2 * g + 1], qai, qbj[2 * g + 1], K);
            const double m0 = validj[2 * g] ? v0 : INFINITY, m1 = validj[2 * g + 1] ? v1 : INFINITY;
            // fmin drops NaN operands; the trailing fmin with +inf also clears the both-NaN case
            double incl = fmin(fmin(m0, m1), INFINITY);
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const double other = __shfl_up(incl, off, 16);
                if (tx >= off) incl = fmin(incl, other);
            }
            double excl = __shfl_up(incl, 1, 16);
            excl = tx == 0 ? INFINITY : excl;
            const double p0 = fmin(carry, excl);
            const double p1 = fmin(p0, m0);
            if (leftj[2 * g]) {
                left_prefixed = fmin(left_prefixed, (v0 < p0) ? p0 : v0);  // displaced root, or the new item itself
                left_min = fmin(left_min, v0);
            }
            if (leftj[2 * g + 1]) {
                left_prefixed = fmin(left_prefixed, (v1 < p1) ? p1 : v1);
                left_min = fmin(left_min, v1);
            }
            if (validj[2 * g]) top = min_at(top, MinAt{v0, c0});
            if (validj[2 * g + 1]) top = min_at(top, MinAt{v1, c0 + 1});
            carry = fmin(carry, __shfl(incl, 15, 16));
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            MinAt o;
            o.v = __shfl_xor(top.v, off, 16);
            o.i = __shfl_xor(top.i, off, 16);
            top = min_at(top, o);
            left_prefixed = fmin(left_prefixed, __shfl_xor(left_prefixed, off, 16));
            left_min = fmin(left_min, __shfl_xor(left_min, off, 16));
        }
        if (tx == 0 && row_valid) {
            TileSummary r;
            r.tile_min = top.v;
            r.left_prefixed = left_prefixed;
            r.left_min = left_min;
            r.tile_arg = top.i;
            tiles[tile.b_tile * nA + ia] = r;
        }
    }
}

// one thread per A-feature walks its tile summaries in B order
__global__ void summary_combine_kernel(const TileSummary* __restrict__ tiles, int64_t nA, int64_t nB,
                                       int64_t n_tiles, double* __restrict__ best, int32_t* __restrict__ arg,
                                       double* __restrict__ second) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nA) return;
    double before = INFINITY, sec = INFINITY;  // `before`: minimum of all columns ahead of the tile
    MinAt top = {INFINITY, INT64_MAX};
    for (int64_t t = 0; t < n_tiles; ++t) {
        const TileSummary r = tiles[t * nA + row];
        const double via_before = (r.left_min < before) ? before : r.left_min;
        sec = fmin(sec, fmin(r.left_prefixed, via_before));
        top = min_at(top, MinAt{r.tile_min, r.tile_arg});
        before = fmin(before, r.tile_min);
    }
    best[row] = top.v;
    arg[row] = (int32_t)top.i;
    second[row] = nB > 1 ? sec : NAN;
}

// LDS-DMA staging reads whole 128-feature rows: every row must be 16-byte aligned and long enough that the last
// tile stays inside it (what it reads beyond n only feeds outputs that are never written).
bool direct_staging_ok(const double* patches, int64_t stride, int64_t n) {
    return (reinterpret_cast<uintptr_t>(patches) & 15u) == 0 && (stride & 1) == 0 &&
           stride >= (n + kTileA - 1) / kTileA * kTileA;
}

// metric code of the C ABI -> tile-core mode (+ the integer kind); -1: unknown
int decode_metric(int metric, int window_elements, IntKind* kind) {
    *kind = IntKind{0, 0};
    if (metric == SFM_MATCH_NCC) return kModeNcc;
    if (metric == SFM_MATCH_SSD) return kModeSsd;
    if ((metric & ~0xFF) != 0x100) return -1;
    const int bits = metric & 0x7F, is_signed = (metric >> 7) & 1;
    if (bits != 8 && bits != 16 && bits != 32 && bits != 64) return -1;
    *kind = IntKind{bits, is_signed};
    if (bits <= 16 && window_elements <= kNarrowMaxWindow) return is_signed ? kModeSsdNarrowSigned : kModeSsdNarrowUnsigned;
    return kModeSsdWide;
}

}  // namespace

extern "C" {

int sfm_patch_extract(const double* image, int64_t height, int64_t width, const double* feats, int64_t n,
                      int window_size, int subtract_mean, int64_t stride, double* patches, double* ssq,
                      uint8_t* ok, void* stream) {
    if (n < 0 || height <= 0 || width <= 0 || window_size < 1)
        return fail(SFM_EINVAL, "sfm_patch_extract: bad size");
    if (subtract_mean != SFM_PATCH_PLAIN && subtract_mean != SFM_PATCH_MEAN_REMOVED && subtract_mean != SFM_PATCH_RAW64)
        return fail(SFM_EINVAL, "sfm_patch_extract: unknown patch mode");
    if (n == 0) return SFM_OK;
    if (stride < n) return fail(SFM_EINVAL, "sfm_patch_extract: stride < n");
    if (!image || !feats || !patches || !ssq || !ok) return fail(SFM_EINVAL, "sfm_patch_extract: null pointer");
    SFM_REQUIRE_GRID("sfm_patch_extract", n, 64, 64);
    hipLaunchKernelGGL(patch_extract_kernel, dim3(grid_for(n, 64)), dim3(64), 0, (hipStream_t)stream, image,
                       height, width, feats, n, stride, window_size / 2, subtract_mean, patches, ssq, ok);
    return check_launch("patch_extract_kernel");
}

int sfm_pair_scores(int metric, const double* patches_a, int64_t stride_a, const double* patches_b,
                    int64_t stride_b, const double* ssq_a, const double* ssq_b, const uint8_t* ok_a,
                    const uint8_t* ok_b, int64_t n_a, int64_t n_b, int window_elements, double* scores,
                    void* stream) {
    if (n_a < 0 || n_b < 0 || window_elements < 1) return fail(SFM_EINVAL, "sfm_pair_scores: bad size");
    IntKind kind;
    const int mode = decode_metric(metric, window_elements, &kind);
    if (mode < 0) return fail(SFM_EINVAL, "sfm_pair_scores: unknown metric");
    if (n_a == 0 || n_b == 0) return SFM_OK;
    if (!patches_a || !patches_b || !ssq_a || !ssq_b || !ok_a || !ok_b || !scores)
        return fail(SFM_EINVAL, "sfm_pair_scores: null pointer");
    const int64_t tiles_a = (n_a + kTileA - 1) / kTileA, tiles_b = (n_b + kTileB - 1) / kTileB;
    if (8 * ((tiles_a + 7) / 8) * tiles_b > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_pair_scores: too many tiles");
    const dim3 grid(tile_grid(tiles_a, tiles_b));
    const bool direct = direct_staging_ok(patches_a, stride_a, n_a) && direct_staging_ok(patches_b, stride_b, n_b);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, (hipStream_t)stream, patches_a, stride_a, patches_b, stride_b,
                           ssq_a, ssq_b, ok_a, ok_b, n_a, n_b, window_elements, scores, kind);
    };
#define SFM_LAUNCH_MODE(M) do { if (direct) launch(pair_scores_kernel<M, true>); else launch(pair_scores_kernel<M, false>); } while (0)
    switch (mode) {
        case kModeNcc: SFM_LAUNCH_MODE(kModeNcc); break;
        case kModeSsd: SFM_LAUNCH_MODE(kModeSsd); break;
        case kModeSsdNarrowSigned: SFM_LAUNCH_MODE(kModeSsdNarrowSigned); break;
        case kModeSsdNarrowUnsigned: SFM_LAUNCH_MODE(kModeSsdNarrowUnsigned); break;
        default: SFM_LAUNCH_MODE(kModeSsdWide); break;
    }
#undef SFM_LAUNCH_MODE
    return check_launch("pair_scores_kernel");
}

int64_t sfm_match_summary_workspace_bytes(int64_t n_a, int64_t n_b) {
    if (n_a < 0 || n_b < 0) return -1;
    return (int64_t)sizeof(TileSummary) * n_a * ((n_b + kTileB - 1) / kTileB);
}

int sfm_match_summary(int metric, const double* patches_a, int64_t stride_a, const double* patches_b,
                      int64_t stride_b, const double* ssq_a, const double* ssq_b, const uint8_t* ok_a,
                      const uint8_t* ok_b, int64_t n_a, int64_t n_b, int window_elements, void* workspace,
                      int64_t workspace_bytes, double* best, int32_t* arg, double* second, void* stream) {
    if (n_a < 0 || n_b < 0 || window_elements < 1) return fail(SFM_EINVAL, "sfm_match_summary: bad size");
    IntKind kind;
    const int mode = decode_metric(metric, window_elements, &kind);
    if (mode < 0) return fail(SFM_EINVAL, "sfm_match_summary: unknown metric");
    if (n_a == 0) return SFM_OK;
    if (n_b == 0) return fail(SFM_EINVAL, "sfm_match_summary: empty rows");
    if (n_b > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_match_summary: rows too long");
    if (!patches_a || !patches_b || !ssq_a || !ssq_b || !ok_a || !ok_b || !workspace || !best || !arg || !second)
        return fail(SFM_EINVAL, "sfm_match_summary: null pointer");
    if (workspace_bytes < sfm_match_summary_workspace_bytes(n_a, n_b))
        return fail(SFM_EINVAL, "sfm_match_summary: workspace smaller than sfm_match_summary_workspace_bytes(n_a, n_b)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0)
        return fail(SFM_EINVAL, "sfm_match_summary: workspace must be 16-byte aligned");
    const int64_t n_tiles = (n_b + kTileB - 1) / kTileB;
    const int64_t tiles_a = (n_a + kTileA - 1) / kTileA;
    if (8 * ((tiles_a + 7) / 8) * n_tiles > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_match_summary: too many tiles");
    SFM_REQUIRE_GRID("sfm_match_summary", n_a, 256, 256);
    const dim3 grid(tile_grid(tiles_a, n_tiles));
    TileSummary* tiles = static_cast<TileSummary*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const bool direct = direct_staging_ok(patches_a, stride_a, n_a) && direct_staging_ok(patches_b, stride_b, n_b);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, st, patches_a, stride_a, patches_b, stride_b, ssq_a, ssq_b,
                           ok_a, ok_b, n_a, n_b, window_elements, tiles, kind);
    };
#define SFM_LAUNCH_MODE(M) do { if (direct) launch(pair_summary_kernel<M, true>); else launch(pair_summary_kernel<M, false>); } while (0)
    switch (mode) {
        case kModeNcc: SFM_LAUNCH_MODE(kModeNcc); break;
        case kModeSsd: SFM_LAUNCH_MODE(kModeSsd); break;
        case kModeSsdNarrowSigned: SFM_LAUNCH_MODE(kModeSsdNarrowSigned); break;
        case kModeSsdNarrowUnsigned: SFM_LAUNCH_MODE(kModeSsdNarrowUnsigned); break;
        default: SFM_LAUNCH_MODE(kModeSsdWide); break;
    }
#undef SFM_LAUNCH_MODE
    hipLaunchKernelGGL(summary_combine_kernel, dim3(grid_for(n_a, 256)), dim3(256), 0, st, tiles, n_a, n_b, n_tiles,
                       best, arg, second);
    return check_launch("pair_summary_kernel");
}

int sfm_match_row_summary(const double* scores, int64_t n_a, int64_t n_b, double* best, int32_t* arg,
                          double* second, void* stream) {
    if (n_a < 0 || n_b < 0) return fail(SFM_EINVAL, "sfm_match_row_summary: negative size");
    if (n_a == 0) return SFM_OK;
    if (n_b == 0) return fail(SFM_EINVAL, "sfm_match_row_summary: empty rows");
    if (n_b > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_match_row_summary: rows too long");
    if (!scores || !best || !arg || !second) return fail(SFM_EINVAL, "sfm_match_row_summary: null pointer");
    SFM_REQUIRE_GRID("sfm_match_row_summary", n_a, 256 / kWave, 256);
    hipLaunchKernelGGL(row_summary_kernel, dim3(grid_for(n_a, 256 / kWave)), dim3(256), 0, (hipStream_t)stream,
                       scores, n_a, n_b, best, arg, second);
    return check_launch("row_summary_kernel");
}

}  // extern "C"
