// Harris corner detector stencils (reference lib/harris/harris_detector.py:57-113 and
// lib/common/correlate.py:4-39).
//
//   correlate_kernel    zero-'same' cross-correlation with an odd square kernel; sums run row-major, left to
//                       right, multiply and add rounded separately (bit-identical to the oracle)
//   cornerness_kernel   block sums of Ix^2, IxIy, Iy^2, then det(M) - k trace(M)^2 (optionally clamped at 0)
//   nms_inplace_kernel  the reference suppresses non-maxima IN PLACE in raster order, so neighbours visited
//                       earlier may already be zero when a pixel is tested.  Pixel (r, c) depends on (r, c-1) and
//                       on row r-1 up to column c+1, hence all pixels with equal t = c + 2r are independent:
//                       one 1024-thread block sweeps t = 0 .. (w-1) + 2(h-1) with a barrier per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfm_common.h"
#include "sfm_math.h"

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;
using sfmhost::grid_stride;

__global__ void correlate_kernel(const double* __restrict__ image, int64_t h, int64_t w,
                                 const double* __restrict__ kernel, int ks, double* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c >= w) return;
    const int half = ks / 2;
    double acc = 0.0;
    const bool interior = r >= half && r < h - half && c >= half && c < w - half;
    if (interior) {
        for (int dr = 0; dr < ks; ++dr)
            for (int dc = 0; dc < ks; ++dc)
                acc = acc + image[(r - half + dr) * w + (c - half + dc)] * kernel[dr * ks + dc];
    }
    out[r * w + c] = acc;
}

__global__ void cornerness_kernel(const double* __restrict__ sx, const double* __restrict__ sy, int64_t h,
                                  int64_t w, int block, double k, int clamp_negative, int64_t out_h,
                                  int64_t out_w, double* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = blockIdx.y;
    if (c >= out_w || r >= out_h) return;
    double value = 0.0;
    if (r < h - block && c < w - block) {  // harris_detector.py:76-79: range(height - block), range(width - block)
        double a = 0.0, b = 0.0, d = 0.0;
        for (int dr = 0; dr < block; ++dr)
            for (int dc = 0; dc < block; ++dc) {
                const double gx = sx[(r + dr) * w + (c + dc)], gy = sy[(r + dr) * w + (c + dc)];
                a = a + gx * gx;
                b = b + gx * gy;
                d = d + gy * gy;
            }
        const double trace = a + d;
        value = (a * d - b * b) - k * (trace * trace);
        if (clamp_negative && value < 0.0) value = 0.0;
    }
    out[r * out_w + c] = value;
}

__global__ __launch_bounds__(1024) void nms_inplace_kernel(double* image, int64_t h, int64_t w) {
    const int64_t steps = (w - 1) + 2 * (h - 1);
    for (int64_t t = 0; t <= steps; ++t) {
        // rows r with 0 <= t - 2r <= w-1
        const int64_t r_lo = t > (w - 1) ? (t - (w - 1) + 1) / 2 : 0;
        const int64_t r_hi = min(h - 1, t / 2);
        for (int64_t r = r_lo + threadIdx.x; r <= r_hi; r += blockDim.x) {
            const int64_t c = t - 2 * r;
            const double v = image[r * w + c];
            double top = v;
            bool any_nan = v != v;   // np.amax propagates a NaN: `pixel < nan` is false, a window with a NaN keeps its centre
            const int64_t r0 = max((int64_t)0, r - 1), r1 = min(h - 1, r + 1);
            const int64_t c0 = max((int64_t)0, c - 1), c1 = min(w - 1, c + 1);
            for (int64_t rr = r0; rr <= r1; ++rr)
                for (int64_t cc = c0; cc <= c1; ++cc) {
                    const double x = image[rr * w + cc];
                    any_nan |= x != x;
                    top = fmax(top, x);
                }
            if (!any_nan && v < top) image[r * w + c] = 0.0;
        }
        __syncthreads();  // workgroup-scope visibility of the stores: the whole sweep runs on one CU
    }
}

// The same in-place raster-order suppression as a parallel fixpoint.  A pixel p is suppressed iff some neighbour
// holds a larger value at the moment p is visited: a raster-LATER neighbour still has its original value; a
// raster-EARLIER neighbour q counts only if it survived itself (otherwise it already reads 0).  So
//   dead(p)  <=> exists later q: orig(q) > orig(p),  or  exists earlier q: orig(q) > orig(p) and alive(q)
//   alive(p) <=> no later q is larger, and every larger earlier q is dead.
// Dependencies only point to strictly larger earlier neighbours, so they form a DAG; each round resolves every
// pixel whose larger earlier neighbours are resolved (states only move unknown -> alive/dead, so reading a stale
// "unknown" merely postpones a decision).  Rounds needed = longest such chain: a handful on natural images.
// state byte: bits 0-1  0 unknown, 1 alive, 2 dead;  bits 4-7 (unknown pixels) which raster-earlier neighbours are larger — left,
// upper left, upper, upper right.  A ZERO byte is a pixel no round has looked at yet: the first round on a zero-initialised
// state CLASSIFIES every pixel from the image (dead by a later neighbour / alive with no larger earlier neighbour / unknown with
// its mask); every further round reads state bytes only — one per larger earlier neighbour of a pixel still unknown — and
// never the image again.  (Until round 5 every round re-read the 3 x 3 doubles of every unknown pixel, and since the unknown
// pixels are scattered nearly every wave still had one: rounds 1-4 of a 1080p image took 116, 105, 99 and 70 us with 12 %, 5 %,
// 2 % and 0.6 % of the pixels unknown.)  A thread owns four consecutive pixels of a row: the classification loads a 3 x 6
// window for them instead of 4 x 9 values.  unresolved[0] receives the number of pixels still unknown after the round.
constexpr int kNmsPixels = 4;
__global__ __launch_bounds__(256) void nms_round_kernel(const double* __restrict__ image, uint8_t* state, int64_t h, int64_t w,
                                                        int32_t* __restrict__ unresolved) {
    __shared__ int block_pending;
    if (threadIdx.x == 0) block_pending = 0;
    __syncthreads();
    const int64_t c0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kNmsPixels;
    const int64_t r = blockIdx.y;
    int pending = 0;
    if (c0 < w) {
        uint8_t st[kNmsPixels];
        bool fresh = false, open = false;
#pragma unroll
        for (int k = 0; k < kNmsPixels; ++k) {
            st[k] = c0 + k < w ? state[r * w + c0 + k] : (uint8_t)2;
            fresh |= st[k] == 0;
            open |= (st[k] & 3) == 0;
        }
        if (fresh) {
            // rows r - 1 .. r + 1, columns c0 - 1 .. c0 + 4; outside the image: -inf (never larger than anything)
            double win[3][kNmsPixels + 2];
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < kNmsPixels + 2; ++dc) {
                    const int64_t rr = r + dr - 1, cc = c0 + dc - 1;
                    win[dr][dc] = (rr >= 0 && rr < h && cc >= 0 && cc < w) ? image[rr * w + cc] : -INFINITY;
                }
#pragma unroll
            for (int k = 0; k < kNmsPixels; ++k) {
                if (st[k] != 0) continue;   // (a pixel beyond the row, or — never by the protocol — one already classified)
                const double v = win[1][k + 1];
                bool any_nan = false;   // a NaN anywhere in the window (a NaN pixel is never zeroed, so it stays one): np.amax is NaN,
#pragma unroll                          // `pixel < nan` is false and the centre survives whatever else the window holds
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) any_nan |= win[dr][k + dc] != win[dr][k + dc];
                if (any_nan) {
                    state[r * w + c0 + k] = 1;
                    continue;
                }
                const bool dead = (win[1][k + 2] > v) | (win[2][k] > v) | (win[2][k + 1] > v) | (win[2][k + 2] > v);   // later: original values
                const unsigned mask = (win[1][k] > v ? 1u : 0u) | (win[0][k] > v ? 2u : 0u) | (win[0][k + 1] > v ? 4u : 0u) |
                                      (win[0][k + 2] > v ? 8u : 0u);                                                  // earlier: only survivors will count
                const uint8_t now = dead ? (uint8_t)2 : (mask == 0u ? (uint8_t)1 : (uint8_t)(mask << 4));
                state[r * w + c0 + k] = now;
                pending += (now & 3) == 0 ? 1 : 0;
            }
        } else if (open) {
#pragma unroll
            for (int k = 0; k < kNmsPixels; ++k) {
                if ((st[k] & 3) != 0) continue;
                const int64_t c = c0 + k;
                const unsigned mask = st[k] >> 4;
                bool dead = false, wait = false;
                auto earlier = [&](unsigned bit, int64_t rr, int64_t cc) {
                    if (mask & bit) {   // (the mask is only ever set for neighbours inside the image)
                        const uint8_t q = state[rr * w + cc] & 3;
                        dead |= q == 1;
                        wait |= q == 0;
                    }
                };
                earlier(1u, r, c - 1);
                earlier(2u, r - 1, c - 1);
                earlier(4u, r - 1, c);
                earlier(8u, r - 1, c + 1);
                if (dead) state[r * w + c] = 2;
                else if (!wait) state[r * w + c] = 1;
                else pending += 1;
            }
        }
    }
    // what is left: one add per wave into LDS, one per BLOCK with pending pixels on the counter (one per wave — 32 000 atomics on
    // one word in the early rounds of a 1080p image — took 94 us a round: profiles/r05/README.md)
    const int wave_pending = sfm::wave_sum(pending);
    if ((threadIdx.x & (kWave - 1)) == 0 && wave_pending != 0) atomicAdd(&block_pending, wave_pending);
    __syncthreads();
    if (threadIdx.x == 0 && block_pending != 0) atomicAdd(unresolved, block_pending);
}

__global__ void nms_finalize_kernel(double* __restrict__ image, const uint8_t* __restrict__ state, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && (state[i] & 3) != 1) image[i] = 0.0;
}

__global__ void zero_counter_kernel(int32_t* counter) { *counter = 0; }

// Compaction of the suppressed cornerness image: (flat index, value) of every non-zero pixel, in no particular
// order.  After suppression a few thousand of ~300 k pixels survive (tens of thousands on integer-valued images, whose
// equal neighbours all survive); only they need to reach the host for the top-k selection of harris_detector.py:32-42.
// A block first COUNTS the survivors of all the pixels it walks, reserves their slots with one atomic, and walks the pixels
// again to write them (the image is L2-resident): at most 1024 atomics on the counter per launch — one per wave and
// iteration were 32 000 at 1080p, 370 us of serialised read-modify-writes for a 20 us scan.
__global__ __launch_bounds__(256) void compact_nonzero_kernel(const double* __restrict__ image, int64_t count,
                                                              int32_t capacity, int32_t* __restrict__ counter,
                                                              int32_t* __restrict__ index,
                                                              double* __restrict__ value) {
    __shared__ int wave_total[256 / kWave];
    __shared__ int block_start;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane);   // whole waves iterate together (the ballots need every lane)
    int mine = 0;   // survivors this wave will write (wave-uniform)
    for (int64_t base = first; base < count; base += stride) {
        const int64_t i = base + lane;
        const double v = i < count ? image[i] : 0.0;
        mine += (int)__popcll(__ballot(v != 0.0));   // NaN != 0 is true: kept, as `cornerness != 0` keeps it in the reference
    }
    if (lane == 0) wave_total[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 256 / kWave; ++w) total += wave_total[w];
        block_start = total != 0 ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int slot0 = block_start;
    for (int w = 0; w < wave; ++w) slot0 += wave_total[w];
    for (int64_t base = first; base < count; base += stride) {
        const int64_t i = base + lane;
        const double v = i < count ? image[i] : 0.0;
        const bool keep = v != 0.0;
        const unsigned long long votes = __ballot(keep);
        if (votes == 0ull) continue;
        const int slot = slot0 + __builtin_amdgcn_mbcnt_hi((unsigned)(votes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)votes, 0));
        if (keep && slot < capacity) {
            index[slot] = (int32_t)i;
            value[slot] = v;
        }
        slot0 += (int)__popcll(votes);
    }
}


// Pruning of the compacted maxima to the ones that can be among the m largest (the top-k selection of harris_detector.py:32-42
// needs num_corners + 1 of them; a 1080p texture leaves 220 000, and copying them back — 2.6 MB — with the host's partition of
// them was half of the whole detector call).  A value's bit pattern, mapped so that unsigned order is numeric order (a NaN of
// either sign on top, so the host still sees it and takes the reference's literal expression), is cut into two 12-bit digits:
// a histogram of the first digit (sign + exponent) finds the digit d1 the m-th largest value has, a histogram of the second
// digit (the top twelve mantissa bits) among the candidates with that d1 finds its d2, and a second compaction keeps the
// candidates at or above (d1, d2): everything at or above the m-th largest value, ties included, plus at most the ones within
// 2^-12 of it.  Both histograms are counted in LDS per block and merged with one atomic per block and non-empty bin (a global
// histogram of 2^16 keys, the first version, took 280 us for the 220 000 maxima of a 1080p image: thousands of them share a key,
// and atomics on one word serialise).
constexpr int kPruneBins = 4096;
SFM_DEVICE unsigned long long prune_order(double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (v != v) return ~0ull;
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}
SFM_DEVICE unsigned prune_digit1(double v) { return (unsigned)(prune_order(v) >> 52); }
SFM_DEVICE unsigned prune_digit2(double v) { return (unsigned)(prune_order(v) >> 40) & (kPruneBins - 1); }
// By a whole block of 256 threads: the largest bin b with at least m entries in bins >= b (0 when there are fewer than m
// altogether: keep everything), and how many entries the bins above b hold.
SFM_DEVICE void prune_threshold(const int32_t* __restrict__ hist, int m, int& bin, int& above) {
    __shared__ int chunk[256];
    __shared__ int result[2];
    const int t = threadIdx.x;
    int own = 0;
#pragma unroll
    for (int k = 0; k < kPruneBins / 256; ++k) own += hist[(kPruneBins / 256) * t + k];
    chunk[t] = own;
    if (t == 0) result[0] = 0, result[1] = 0;
    __syncthreads();
    int higher = 0;   // entries in the chunks above this thread's
    for (int c = t + 1; c < 256; ++c) higher += chunk[c];
    if (higher < m && higher + own >= m) {   // at most one thread: the m-th largest lies in this chunk
        int running = higher;
        for (int k = kPruneBins / 256 - 1; k >= 0; --k) {
            const int here = hist[(kPruneBins / 256) * t + k];
            if (running + here >= m) {
                result[0] = (kPruneBins / 256) * t + k;
                result[1] = running;
                break;
            }
            running += here;
        }
    }
    __syncthreads();
    bin = result[0];
    above = result[1];
    __syncthreads();   // (the shared words are reused by a second call)
}
// LEVEL 1: digit 1 of every candidate; LEVEL 2: digit 2 of the candidates whose digit 1 is the threshold's
template <int LEVEL>
__global__ __launch_bounds__(256) void prune_histogram_kernel(const double* __restrict__ value, const int32_t* __restrict__ found,
                                                              int32_t capacity, int32_t m, const int32_t* __restrict__ hist1,
                                                              int32_t* __restrict__ hist) {
    __shared__ int bins[kPruneBins];
    for (int k = threadIdx.x; k < kPruneBins; k += 256) bins[k] = 0;
    int d1 = 0, above = 0;
    if (LEVEL == 2) prune_threshold(hist1, m, d1, above);
    __syncthreads();
    const int n = min(*found, capacity);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = value[i];
        if (LEVEL == 1) atomicAdd(&bins[prune_digit1(v)], 1);
        else if (prune_digit1(v) == (unsigned)d1) atomicAdd(&bins[prune_digit2(v)], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kPruneBins; k += 256)
        if (bins[k] != 0) atomicAdd(hist + k, bins[k]);
}
__global__ __launch_bounds__(256) void prune_filter_kernel(const double* __restrict__ value, const int32_t* __restrict__ index,
                                                           const int32_t* __restrict__ found, int32_t capacity, int32_t m,
                                                           const int32_t* __restrict__ hist1, const int32_t* __restrict__ hist2,
                                                           int32_t capacity_out, int32_t* __restrict__ counter_out,
                                                           int32_t* __restrict__ index_out, double* __restrict__ value_out) {
    __shared__ int wave_total[256 / kWave];
    __shared__ int block_start;
    int d1, above1, d2, above2;
    prune_threshold(hist1, m, d1, above1);
    prune_threshold(hist2, m - above1, d2, above2);
    const unsigned long long T = ((unsigned long long)d1 << 52) | ((unsigned long long)d2 << 40);
    const int n = min(*found, capacity);
    if (n <= 0) return;   // (the same for every thread of the launch)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int stride = gridDim.x * blockDim.x;
    const int first = blockIdx.x * blockDim.x + (threadIdx.x - lane);
    int mine = 0;
    for (int base = first; base < n; base += stride) {
        const int i = base + lane;
        mine += (int)__popcll(__ballot(i < n && prune_order(value[min(i, n - 1)]) >= T));
    }
    if (lane == 0) wave_total[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 256 / kWave; ++w) total += wave_total[w];
        block_start = total != 0 ? atomicAdd(counter_out, total) : 0;
    }
    __syncthreads();
    int slot0 = block_start;
    for (int w = 0; w < wave; ++w) slot0 += wave_total[w];
    for (int base = first; base < n; base += stride) {
        const int i = base + lane;
        const double v = value[min(i, n - 1)];
        const bool keep = i < n && prune_order(v) >= T;
        const unsigned long long votes = __ballot(keep);
        if (votes == 0ull) continue;
        const int slot = slot0 + __builtin_amdgcn_mbcnt_hi((unsigned)(votes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)votes, 0));
        if (keep && slot < capacity_out) {
            index_out[slot] = index[i];
            value_out[slot] = v;
        }
        slot0 += (int)__popcll(votes);
    }
}

}  // namespace

extern "C" {

int sfm_cross_correlate(const double* image, int64_t height, int64_t width, const double* kernel,
                        int kernel_size, double* out, void* stream) {
    if (height <= 0 || width <= 0 || kernel_size < 1 || (kernel_size % 2) == 0)
        return fail(SFM_EINVAL, "sfm_cross_correlate: need a non-empty image and an odd kernel size");
    if (height < kernel_size || width < kernel_size || height > 65535)
        return fail(SFM_EINVAL, "sfm_cross_correlate: kernel larger than image (or more than 65535 rows)");
    if (!image || !kernel || !out) return fail(SFM_EINVAL, "sfm_cross_correlate: null pointer");
    SFM_REQUIRE_GRID("sfm_cross_correlate", width, 256, 256, height);
    hipLaunchKernelGGL(correlate_kernel, dim3(grid_for(width, 256), (unsigned)height), dim3(256), 0,
                       (hipStream_t)stream, image, height, width, kernel, kernel_size, out);
    return check_launch("correlate_kernel");
}

int sfm_harris_cornerness(const double* sobel_x, const double* sobel_y, int64_t height, int64_t width,
                          int block_size, double k, int clamp_negative, int64_t out_height, int64_t out_width,
                          double* out, void* stream) {
    if (height <= 0 || width <= 0 || block_size < 1 || out_height < 0 || out_width < 0 || out_height > 65535)
        return fail(SFM_EINVAL, "sfm_harris_cornerness: bad size");
    if (out_height == 0 || out_width == 0) return SFM_OK;
    if (out_height > height || out_width > width) return fail(SFM_EINVAL, "sfm_harris_cornerness: output larger than input");
    if (!sobel_x || !sobel_y || !out) return fail(SFM_EINVAL, "sfm_harris_cornerness: null pointer");
    SFM_REQUIRE_GRID("sfm_harris_cornerness", out_width, 256, 256, out_height);
    hipLaunchKernelGGL(cornerness_kernel, dim3(grid_for(out_width, 256), (unsigned)out_height), dim3(256), 0,
                       (hipStream_t)stream, sobel_x, sobel_y, height, width, block_size, k, clamp_negative,
                       out_height, out_width, out);
    return check_launch("cornerness_kernel");
}

int sfm_nms_inplace(double* image, int64_t height, int64_t width, void* stream) {
    if (height < 0 || width < 0) return fail(SFM_EINVAL, "sfm_nms_inplace: negative size");
    if (height == 0 || width == 0) return SFM_OK;
    if (!image) return fail(SFM_EINVAL, "sfm_nms_inplace: null pointer");
    hipLaunchKernelGGL(nms_inplace_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, image, height, width);
    return check_launch("nms_inplace_kernel");
}

int sfm_nms_round(const double* image, uint8_t* state, int64_t height, int64_t width, int32_t* unresolved,
                  void* stream) {
    if (height <= 0 || width <= 0 || height > 65535) return fail(SFM_EINVAL, "sfm_nms_round: bad size");
    if (!image || !state || !unresolved) return fail(SFM_EINVAL, "sfm_nms_round: null pointer");
    SFM_REQUIRE_GRID("sfm_nms_round", (width + kNmsPixels - 1) / kNmsPixels, 256, 256, height);
    hipLaunchKernelGGL(nms_round_kernel, dim3(grid_for((width + kNmsPixels - 1) / kNmsPixels, 256), (unsigned)height), dim3(256), 0,
                       (hipStream_t)stream, image, state, height, width, unresolved);
    return check_launch("nms_round_kernel");
}

int sfm_nms_finalize(double* image, const uint8_t* state, int64_t height, int64_t width, void* stream) {
    if (height < 0 || width < 0) return fail(SFM_EINVAL, "sfm_nms_finalize: negative size");
    if (height == 0 || width == 0) return SFM_OK;
    if (!image || !state) return fail(SFM_EINVAL, "sfm_nms_finalize: null pointer");
    SFM_REQUIRE_GRID("sfm_nms_finalize", height * width, 256, 256);
    hipLaunchKernelGGL(nms_finalize_kernel, dim3(grid_for(height * width, 256)), dim3(256), 0, (hipStream_t)stream,
                       image, state, height * width);
    return check_launch("nms_finalize_kernel");
}

int sfm_compact_nonzero(const double* image, int64_t count, int32_t capacity, int32_t* counter, int32_t* index,
                        double* value, void* stream) {
    if (count < 0 || capacity < 0) return fail(SFM_EINVAL, "sfm_compact_nonzero: negative size");
    if (count > 0x7FFFFFFF) return fail(SFM_EINVAL, "sfm_compact_nonzero: image too large for int32 indices");
    if (!counter || (count > 0 && !image) || (capacity > 0 && (!index || !value)))
        return fail(SFM_EINVAL, "sfm_compact_nonzero: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_counter_kernel, dim3(1), dim3(1), 0, st, counter);
    if (count > 0)
        hipLaunchKernelGGL(compact_nonzero_kernel, dim3(grid_stride(count, 256, 1024)), dim3(256), 0, st, image, count,
                           capacity, counter, index, value);
    return check_launch("compact_nonzero_kernel");
}

int sfm_prune_top(const double* value, const int32_t* index, const int32_t* found, int32_t capacity, int32_t m, void* workspace,
                  int32_t capacity_out, int32_t* counter_out, int32_t* index_out, double* value_out, void* stream) {
    if (capacity < 0 || capacity_out < 0 || m < 1) return fail(SFM_EINVAL, "sfm_prune_top: bad size");
    if (!found || !workspace || !counter_out || (capacity > 0 && (!value || !index)) || (capacity_out > 0 && (!index_out || !value_out)))
        return fail(SFM_EINVAL, "sfm_prune_top: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int32_t* hist1 = reinterpret_cast<int32_t*>(workspace);   // the two histograms of kPruneBins counters
    int32_t* hist2 = hist1 + kPruneBins;
    if (hipMemsetAsync(hist1, 0, (size_t)(2 * kPruneBins) * sizeof(int32_t), st) != hipSuccess)
        return fail(SFM_EHIP, "sfm_prune_top: hipMemsetAsync failed");
    hipLaunchKernelGGL(zero_counter_kernel, dim3(1), dim3(1), 0, st, counter_out);
    if (capacity > 0) {
        const unsigned blocks = (unsigned)grid_stride(capacity, 256, 256);
        hipLaunchKernelGGL(prune_histogram_kernel<1>, dim3(blocks), dim3(256), 0, st, value, found, capacity, m, hist1, hist1);
        hipLaunchKernelGGL(prune_histogram_kernel<2>, dim3(blocks), dim3(256), 0, st, value, found, capacity, m, hist1, hist2);
        hipLaunchKernelGGL(prune_filter_kernel, dim3(blocks), dim3(256), 0, st, value, index, found, capacity, m, hist1, hist2,
                           capacity_out, counter_out, index_out, value_out);
    }
    return check_launch("prune_filter_kernel");
}

}  // extern "C"
