// Microbenchmark (gfx950): v_fma_f32 vs v_pk_fma_f32 issue cost (8 independent chains, 8 / 4 / 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>  // 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_pk_mul_f32 + v_pk_add_f32
__global__ __launch_bounds__(256) void k(int iters, float m, float c, float* out) {
    float f[8];
    float2v p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] = threadIdx.x + j; p[j] = float2v{(float)threadIdx.x, (float)j}; }
    const float2v pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
            if (OP == 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pm));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pc));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j] + p[j].x + p[j].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
float run(int blocks, int iters, float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, 1e-9f, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, 1e-9f, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out;
    if (hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    const int it = 40000;
    for (int w : {8, 4, 2}) {
        const int blocks = 256 * w;
        const double inst = (double)w * 8.0 * it;
        const float t0 = run<0>(blocks, it, out), t1 = run<1>(blocks, it, out), t2 = run<2>(blocks, it, out);
        printf("%d waves/SIMD: nominal 2.4 GHz cycles per wave-instruction: v_fma_f32 %.2f  v_pk_fma_f32 %.2f  "
               "v_pk_mul_f32+v_pk_add_f32 %.2f each\n", w, t0 * 2.4e6 / inst, t1 * 2.4e6 / inst, t2 * 2.4e6 / (2 * inst));
    }
    return 0;
}
