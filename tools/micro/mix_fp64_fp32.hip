// Microbenchmark: VALU issue cost when fp64 and fp32 waves share a SIMD (gfx950).
// Each wave runs ITER iterations of 8 independent FMA chains, fp64 or fp32 depending on its wave index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>  // 0: all fp64, 1: all fp32, 2: odd waves fp64 / even fp32, 3: one wave in four fp64
__global__ __launch_bounds__(256) void k(int iters64, int iters32, float* out) {
    const int wave = threadIdx.x / 64 + blockIdx.x * 4;
    // type by (blockIdx / #CUs): blocks are dealt round-robin over the 256 CUs, so every CU hosts both kinds
    const int gen = blockIdx.x >> 8;
    bool dbl = MODE == 0 ? true : MODE == 1 ? false : MODE == 2 ? (gen & 1) : ((gen & 3) == 0);
    float acc = 0.f;
    if (dbl) {
        double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters64; ++i) {
            a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
            a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
        }
        acc = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    } else {
        float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
        const float m = 1.0000001f, c = 1e-9f;
        for (int i = 0; i < iters32; ++i) {
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        }
        acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    if (acc == 12345.678f) out[wave] = acc;
}

template <int MODE>
float run(int blocks, int it64, int it32, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, it64, it32, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, it64, it32, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 1 << 24));
    const int blocks = 256 * 8;          // 8 blocks per CU = 8 waves per SIMD
    const int it = 20000;
    // issue cycles per wave: fp64 8*it*4, fp32 8*it*2 (if 4 / 2 cycles per instruction)
    float t64 = run<0>(blocks, it, it, out);
    float t32 = run<1>(blocks, it, it, out);
    float tmix = run<2>(blocks, it, it, out);       // half the blocks fp64, half fp32, same instruction count
    float tmix4 = run<3>(blocks, it, it, out);      // a quarter fp64
    printf("all fp64 %.3f ms, all fp32 %.3f ms, half/half %.3f ms (sum/2 = %.3f), quarter fp64 %.3f ms (expected %.3f)\n",
           t64, t32, tmix, 0.5f * (t64 + t32), tmix4, 0.25f * t64 + 0.75f * t32);
    // per-instruction cycles at 2.4 GHz nominal: waves per SIMD = 8
    double inst_per_simd = 8.0 * 8.0 * it;  // waves * chains * iters
    printf("cycles/instr (at 2.4 GHz): fp64 %.2f, fp32 %.2f\n", t64 * 1e-3 * 2.4e9 / inst_per_simd, t32 * 1e-3 * 2.4e9 / inst_per_simd);
    return 0;
}
