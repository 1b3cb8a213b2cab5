// Microbenchmark (gfx950): issue cost of v_fma_f64 vs v_mul_f64 vs v_add_f64, 8 independent chains per wave,
// 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/fp64_op_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>  // 0 fma, 1 mul, 2 add, 3 mul+add pairs (dependent), 4 fp32 fma
__global__ __launch_bounds__(256) void k(int iters, double m, double c, float* out) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = threadIdx.x + j;
    const float mf = (float)m, cf = (float)c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(m));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 3) {
                double t;
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(a[j]), "v"(m));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(t));
            }
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(mf), "v"(cf));
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + f[j];
    if (s == 12345.678) out[threadIdx.x] = (float)s;
}

template <int OP>
float run(int blocks, int iters, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001, 1e-9, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 1 << 20));
    const int it = 20000;
    for (int waves_per_simd : {8, 2, 1}) {
        const int blocks = 256 * waves_per_simd;
        const double inst = (double)waves_per_simd * 8.0 * it;  // wave-instructions per SIMD (OP 3: twice that)
        const float t0 = run<0>(blocks, it, out), t1 = run<1>(blocks, it, out), t2 = run<2>(blocks, it, out),
                    t3 = run<3>(blocks, it, out), t4 = run<4>(blocks, it, out);
        printf("%d waves/SIMD  cycles per wave-instruction at 2.4 GHz: v_fma_f64 %.2f  v_mul_f64 %.2f  v_add_f64 %.2f  "
               "mul->add pair %.2f per instr  v_fma_f32 %.2f\n", waves_per_simd, t0 * 2.4e6 / inst, t1 * 2.4e6 / inst,
               t2 * 2.4e6 / inst, t3 * 2.4e6 / (2 * inst), t4 * 2.4e6 / inst);
    }
    return 0;
}
