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

namespace {

using sfmhost::check_launch;
using sfmhost::fail;
using sfmhost::grid_for;

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
            const int64_t r0 = max((int64_t)0, r - 1), r1 = min(h - 1, r + 1);
            const int64_t c0 = max((int64_t)0, c - 1), c1 = min(w - 1, c + 1);
            for (int64_t rr = r0; rr <= r1; ++rr)
                for (int64_t cc = c0; cc <= c1; ++cc) top = fmax(top, image[rr * w + cc]);
            if (v < top) image[r * w + c] = 0.0;
        }
        __syncthreads();  // workgroup-scope visibility of the stores: the whole sweep runs on one CU
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

}  // extern "C"
