// Microbenchmark (gfx950): issue cost of v_pk_fma_f16 (two fp16 FMAs per lane and instruction) next to v_fma_f32, with
// all-VGPR operands, with an SGPR multiplier, and with an op_sel broadcast of one half — the operand forms a packed
// half-precision pre-filter of the scoring kernel would use.  8 independent chains per lane, 8 / 5 waves per SIMD.
// Prints nominal 2.4 GHz cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

template <int OP>  // 0: v_fma_f32   1: v_pk_fma_f16 vgpr   2: v_pk_fma_f16 sgpr multiplier   3: v_pk_fma_f16 op_sel broadcast
                   // 4: v_cmp_gt_f16 (vcc)   5: v_pk_mul_f16
__global__ __launch_bounds__(256) void k(int iters, float m, float c, float* out) {
    float f[8];
    half2v p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] = threadIdx.x + j; p[j] = half2v{(_Float16)(threadIdx.x * 0.001f), (_Float16)(j * 0.01f)}; }
    const half2v pm = {(_Float16)m, (_Float16)m}, pc = {(_Float16)c, (_Float16)c};
    unsigned sm;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sm) : "v"(pm));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
            if (OP == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(p[j]) : "s"(sm), "v"(pc));
            if (OP == 3) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[j]) : "v"(pm), "v"(pc));
            if (OP == 4) asm volatile("v_cmp_gt_f16 vcc, %0, %1" : : "v"(p[j]), "v"(pm) : "vcc");
            if (OP == 5) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(p[j]) : "v"(pm));
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j] + (float)p[j].x + (float)p[j].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
float run(int blocks, int iters, float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0001f, 1e-4f, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, 1.0001f, 1e-4f, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out;
    if (hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    const int it = 40000;
    for (int w : {8, 5}) {
        const int blocks = 256 * w;
        const double inst = (double)w * 8.0 * it;
        printf("%d waves/SIMD: nominal 2.4 GHz cycles per wave-instruction: v_fma_f32 %.2f  v_pk_fma_f16 %.2f  "
               "v_pk_fma_f16 (SGPR multiplier) %.2f  v_pk_fma_f16 (op_sel broadcast) %.2f  v_cmp_gt_f16 %.2f  v_pk_mul_f16 %.2f\n", w,
               run<0>(blocks, it, out) * 2.4e6 / inst, run<1>(blocks, it, out) * 2.4e6 / inst, run<2>(blocks, it, out) * 2.4e6 / inst,
               run<3>(blocks, it, out) * 2.4e6 / inst, run<4>(blocks, it, out) * 2.4e6 / inst, run<5>(blocks, it, out) * 2.4e6 / inst);
    }
    return 0;
}
