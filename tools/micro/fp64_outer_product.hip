// Microbenchmark (gfx950): the matcher's inner body — an 8x8 fp64 outer-product accumulate per lane, operands in
// registers (no LDS) — as separate multiply + add (what -ffp-contract=off gives) and as FMA, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/fp64_outer_product.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int C, int OP>  // OP 0: mul then add; 1: fma; 2: R x C adds only
__global__ __launch_bounds__(256) void k(int iters, const double* __restrict__ in, double* __restrict__ out) {
    double acc[R][C], a[R], b[C];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = in[threadIdx.x + i];
#pragma unroll
    for (int j = 0; j < C; ++j) b[j] = in[threadIdx.x + 64 + j];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) acc[i][j] = 0.0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j) {
                if (OP == 0) acc[i][j] += a[i] * b[j];
                if (OP == 1) acc[i][j] = fma(a[i], b[j], acc[i][j]);
                if (OP == 2) acc[i][j] += a[i];
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int C, int OP>
float run(int blocks, int iters, const double* in, double* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<R, C, OP>), dim3(blocks), dim3(256), 0, 0, iters, in, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<R, C, OP>), dim3(blocks), dim3(256), 0, 0, iters, in, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

template <int R, int C>
void report(const double* in, double* out) {
    const int it = 4000;
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w;  // one 4-wave block per CU per `w`: w waves per SIMD
        const double per = (double)w * R * C * it;  // accumulate steps per SIMD
        const float t0 = run<R, C, 0>(blocks, it, in, out), t1 = run<R, C, 1>(blocks, it, in, out),
                    t2 = run<R, C, 2>(blocks, it, in, out);
        printf("%dx%d tile, %d waves/SIMD: nominal 2.4 GHz cycles per accumulate step: mul+add %.2f (2 instr)  fma %.2f  add only %.2f\n",
               R, C, w, t0 * 2.4e6 / per, t1 * 2.4e6 / per, t2 * 2.4e6 / per);
    }
}

int main() {
    double *in, *out; CHECK(hipMalloc(&in, 1 << 16)); CHECK(hipMalloc(&out, 256 * 4 * 256 * 8));
    CHECK(hipMemset(in, 0, 1 << 16));
    report<8, 8>(in, out);
    report<4, 4>(in, out);
    report<8, 4>(in, out);
    return 0;
}
