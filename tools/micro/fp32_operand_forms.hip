// Microbenchmark (gfx950): issue cost of v_fma_f32 by operand form — does the 3.05-cycle figure of fp32_packed_rate.hip
// (v_fma_f32 acc, acc, m, c with m, c in fixed VGPRs) come from VGPR bank conflicts?  The VGPR file has 4 banks
// (register index mod 4); an instruction whose source VGPRs collide in a bank needs extra read cycles.
//   form 0  d = fma(d, m, c)    three VGPR sources, registers pinned so that banks are all DIFFERENT
//   form 1  d = fma(d, m, c)    three VGPR sources pinned to the SAME bank
//   form 2  d = fma(d, s, c)    one SGPR multiplier, two VGPR sources (different banks)
//   form 3  d = fma(d, s, d)    one SGPR, one VGPR read twice
//   form 4  d = fma(d, m, c)    compiler-allocated registers (as fp32_packed_rate.hip)
//   form 5  d += m * c          v_fmac_f32_e32 (VOP2, 4-byte encoding), VGPR sources
//   form 6  d += s * c          v_fmac_f32_e32 with an SGPR src0
//   form 7  d = d * m           v_mul_f32_e32 (VOP2)
//   form 8  vcc = d > m         v_cmp_gt_f32_e32 (VOPC)
//   form 9  s[n:n+1] = d > m    v_cmp_gt_f32_e64 (VOP3, SGPR-pair destination)
// 8 independent chains per lane, 8 / 5 / 2 waves per SIMD.  Prints nominal 2.4 GHz cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

// in-kernel clock of the load: (shader cycles, 100 MHz real-time ticks) of block 0's first lane (guide: DVFS give-back)
__device__ unsigned long long g_stamp[2];

template <int FORM>
__global__ __launch_bounds__(256) void k(int iters, float m_in, float c_in, float* out) {
    float r = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if (FORM <= 3 || FORM >= 5) {
        // chains in v8..v15; m / c pinned per form.  Banks: v8,v12 -> 0; v9,v13 -> 1; v10,v14 -> 2; v11,v15 -> 3.
        asm volatile(
            "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n"
            "v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"
            "v_mov_b32 v16, %2\n v_mov_b32 v17, %2\n v_mov_b32 v18, %2\n v_mov_b32 v19, %2\n"   // m copies: banks 0..3
            "v_mov_b32 v20, %3\n v_mov_b32 v21, %3\n v_mov_b32 v22, %3\n v_mov_b32 v23, %3\n"   // c copies: banks 0..3
            "v_readfirstlane_b32 s20, %2\n"
            : "=v"(r) : "v"((float)threadIdx.x), "v"(m_in), "v"(c_in)
            : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s20");
        for (int i = 0; i < iters; ++i) {
            if (FORM == 0)  // d bank b, m bank b+1, c bank b+2
                asm volatile(
                    "v_fma_f32 v8, v8, v17, v22\n v_fma_f32 v9, v9, v18, v23\n v_fma_f32 v10, v10, v19, v20\n v_fma_f32 v11, v11, v16, v21\n"
                    "v_fma_f32 v12, v12, v17, v22\n v_fma_f32 v13, v13, v18, v23\n v_fma_f32 v14, v14, v19, v20\n v_fma_f32 v15, v15, v16, v21\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 1)  // all three sources in the chain register's bank
                asm volatile(
                    "v_fma_f32 v8, v8, v16, v20\n v_fma_f32 v9, v9, v17, v21\n v_fma_f32 v10, v10, v18, v22\n v_fma_f32 v11, v11, v19, v23\n"
                    "v_fma_f32 v12, v12, v16, v20\n v_fma_f32 v13, v13, v17, v21\n v_fma_f32 v14, v14, v18, v22\n v_fma_f32 v15, v15, v19, v23\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 2)
                asm volatile(
                    "v_fma_f32 v8, v8, s20, v21\n v_fma_f32 v9, v9, s20, v22\n v_fma_f32 v10, v10, s20, v23\n v_fma_f32 v11, v11, s20, v20\n"
                    "v_fma_f32 v12, v12, s20, v21\n v_fma_f32 v13, v13, s20, v22\n v_fma_f32 v14, v14, s20, v23\n v_fma_f32 v15, v15, s20, v20\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 5)
                asm volatile(
                    "v_fmac_f32_e32 v8, v17, v22\n v_fmac_f32_e32 v9, v18, v23\n v_fmac_f32_e32 v10, v19, v20\n v_fmac_f32_e32 v11, v16, v21\n"
                    "v_fmac_f32_e32 v12, v17, v22\n v_fmac_f32_e32 v13, v18, v23\n v_fmac_f32_e32 v14, v19, v20\n v_fmac_f32_e32 v15, v16, v21\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 6)
                asm volatile(
                    "v_fmac_f32_e32 v8, s20, v22\n v_fmac_f32_e32 v9, s20, v23\n v_fmac_f32_e32 v10, s20, v20\n v_fmac_f32_e32 v11, s20, v21\n"
                    "v_fmac_f32_e32 v12, s20, v22\n v_fmac_f32_e32 v13, s20, v23\n v_fmac_f32_e32 v14, s20, v20\n v_fmac_f32_e32 v15, s20, v21\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 7)
                asm volatile(
                    "v_mul_f32_e32 v8, v8, v17\n v_mul_f32_e32 v9, v9, v18\n v_mul_f32_e32 v10, v10, v19\n v_mul_f32_e32 v11, v11, v16\n"
                    "v_mul_f32_e32 v12, v12, v17\n v_mul_f32_e32 v13, v13, v18\n v_mul_f32_e32 v14, v14, v19\n v_mul_f32_e32 v15, v15, v16\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
            if (FORM == 8)
                asm volatile(
                    "v_cmp_gt_f32_e32 vcc, v8, v17\n v_cmp_gt_f32_e32 vcc, v9, v18\n v_cmp_gt_f32_e32 vcc, v10, v19\n v_cmp_gt_f32_e32 vcc, v11, v16\n"
                    "v_cmp_gt_f32_e32 vcc, v12, v17\n v_cmp_gt_f32_e32 vcc, v13, v18\n v_cmp_gt_f32_e32 vcc, v14, v19\n v_cmp_gt_f32_e32 vcc, v15, v16\n"
                    ::: "vcc");
            if (FORM == 9)
                asm volatile(
                    "v_cmp_gt_f32_e64 s[22:23], v8, v17\n v_cmp_gt_f32_e64 s[24:25], v9, v18\n v_cmp_gt_f32_e64 s[26:27], v10, v19\n v_cmp_gt_f32_e64 s[28:29], v11, v16\n"
                    "v_cmp_gt_f32_e64 s[22:23], v12, v17\n v_cmp_gt_f32_e64 s[24:25], v13, v18\n v_cmp_gt_f32_e64 s[26:27], v14, v19\n v_cmp_gt_f32_e64 s[28:29], v15, v16\n"
                    ::: "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
            if (FORM == 3)
                asm volatile(
                    "v_fma_f32 v8, v8, s20, v8\n v_fma_f32 v9, v9, s20, v9\n v_fma_f32 v10, v10, s20, v10\n v_fma_f32 v11, v11, s20, v11\n"
                    "v_fma_f32 v12, v12, s20, v12\n v_fma_f32 v13, v13, s20, v13\n v_fma_f32 v14, v14, s20, v14\n v_fma_f32 v15, v15, s20, v15\n"
                    ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
        }
        asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n"
                     "v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v15\n"
                     : "=v"(r) :: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    } else {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = threadIdx.x + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(m_in), "v"(c_in));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) r += f[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_stamp[0] = __builtin_amdgcn_s_memtime() - t0;
        g_stamp[1] = __builtin_amdgcn_s_memrealtime() - w0;
    }
}

double last_clock_ghz() {
    unsigned long long h[2];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
    return h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
}

template <int FORM>
float run(int blocks, int iters, float* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, 1e-9f, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, iters, 1.0000001f, 1e-9f, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out;
    if (hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    const int it = 40000;
    for (int w : {8, 5, 2}) {
        (void)run<0>(256 * w, it, out);
        printf("%d waves/SIMD: in-kernel clock under the all-VGPR FMA load %.2f GHz", w, last_clock_ghz());
        (void)run<2>(256 * w, it, out);
        printf(", under the SGPR-source load %.2f GHz\n", last_clock_ghz());
        const int blocks = 256 * w;
        const double inst = (double)w * 8.0 * it;
        printf("%d waves/SIMD, nominal 2.4 GHz cycles per v_fma_f32: banks all different %.2f | same bank %.2f | sgpr x vgpr + vgpr %.2f | "
               "sgpr, one vgpr twice %.2f | compiler-allocated %.2f\n", w, run<0>(blocks, it, out) * 2.4e6 / inst,
               run<1>(blocks, it, out) * 2.4e6 / inst, run<2>(blocks, it, out) * 2.4e6 / inst,
               run<3>(blocks, it, out) * 2.4e6 / inst, run<4>(blocks, it, out) * 2.4e6 / inst);
        printf("    v_fmac_f32_e32 vgpr %.2f | v_fmac_f32_e32 sgpr src0 %.2f | v_mul_f32_e32 %.2f | v_cmp_gt_f32_e32 (vcc) %.2f | "
               "v_cmp_gt_f32_e64 (sgpr pair) %.2f\n", run<5>(blocks, it, out) * 2.4e6 / inst, run<6>(blocks, it, out) * 2.4e6 / inst,
               run<7>(blocks, it, out) * 2.4e6 / inst, run<8>(blocks, it, out) * 2.4e6 / inst, run<9>(blocks, it, out) * 2.4e6 / inst);
    }
    return 0;
}
