// Probe: issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU forms the attention blend can use.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                    \
    __global__ void NAME(float* out, int iters, long long* cyc) {                          \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float w = 0.5f; unsigned pr = 0x3c003c00u + threadIdx.x;                           \
        long long t0 = clock64();                                                          \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(pr)); \
        }                                                                                  \
        long long t1 = clock64();                                                          \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                         \
    }
BODY(k_fma32, "v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %1, %8, %1, %2\n v_fma_f32 %2, %8, %2, %3\n v_fma_f32 %3, %8, %3, %4\n v_fma_f32 %4, %8, %4, %5\n v_fma_f32 %5, %8, %5, %6\n v_fma_f32 %6, %8, %6, %7\n v_fma_f32 %7, %8, %7, %0\n")
BODY(k_mix, "v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n")
BODY(k_cvt, "v_cvt_f32_f16 %0, %9\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %9\n v_cvt_f32_f16 %3, %9\n v_cvt_f32_f16 %4, %9\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %9\n v_cvt_f32_f16 %7, %9\n")
BODY(k_cvt_sdwa, "v_cvt_f32_f16_sdwa %0, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
BODY(k_pkf16, "v_pk_fma_f16 %0, %9, %9, %0\n v_pk_fma_f16 %1, %9, %9, %1\n v_pk_fma_f16 %2, %9, %9, %2\n v_pk_fma_f16 %3, %9, %9, %3\n v_pk_fma_f16 %4, %9, %9, %4\n v_pk_fma_f16 %5, %9, %9, %5\n v_pk_fma_f16 %6, %9, %9, %6\n v_pk_fma_f16 %7, %9, %9, %7\n")
BODY(k_dot2, "v_dot2_f32_f16 %0, %9, %9, %0\n v_dot2_f32_f16 %1, %9, %9, %1\n v_dot2_f32_f16 %2, %9, %9, %2\n v_dot2_f32_f16 %3, %9, %9, %3\n v_dot2_f32_f16 %4, %9, %9, %4\n v_dot2_f32_f16 %5, %9, %9, %5\n v_dot2_f32_f16 %6, %9, %9, %6\n v_dot2_f32_f16 %7, %9, %9, %7\n")
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %8, %1\n v_cvt_pk_f16_f32 %1, %8, %2\n v_cvt_pk_f16_f32 %2, %8, %3\n v_cvt_pk_f16_f32 %3, %8, %4\n v_cvt_pk_f16_f32 %4, %8, %5\n v_cvt_pk_f16_f32 %5, %8, %6\n v_cvt_pk_f16_f32 %6, %8, %7\n v_cvt_pk_f16_f32 %7, %8, %0\n")
BODY(k_exp, "v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8\n")
BODY(k_mixlo, "v_fma_mixlo_f16 %0, %8, %9, %0 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %1, %8, %9, %1 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %2, %8, %9, %2 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %3, %8, %9, %3 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %4, %8, %9, %4 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %5, %8, %9, %5 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %6, %8, %9, %6 op_sel_hi:[0,1,0]\n v_fma_mixlo_f16 %7, %8, %9, %7 op_sel_hi:[0,1,0]\n")

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_pkf32(float* out, int iters, long long* cyc) {
    f2 a0 = {1.f * threadIdx.x, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, w = {0.5f, 0.25f};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_pk_fma_f32 %0, %4, %0, %1\n v_pk_fma_f32 %1, %4, %1, %2\n v_pk_fma_f32 %2, %4, %2, %3\n v_pk_fma_f32 %3, %4, %3, %0\n v_pk_fma_f32 %0, %4, %0, %1\n v_pk_fma_f32 %1, %4, %1, %2\n v_pk_fma_f32 %2, %4, %2, %3\n v_pk_fma_f32 %3, %4, %3, %0\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0.x + a1.y + a2.x + a3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
void run(const char* name, K k, int waves_per_simd) {
    float* out; hipMalloc(&out, 1 << 22); long long* dc; hipMalloc(&dc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, 100, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-12s waves/simd=%d  %.3f ms  %.2f ns per instr per wave, clock64 ticks per instr %.2f\n", name, waves_per_simd, ms,
           ms * 1e6 / (iters * 64.0) , (double)c / (iters * 64.0));
    hipFree(out); hipFree(dc);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run("fma_f32", k_fma32, w); run("fma_mix_f32", k_mix, w); run("cvt_f32_f16", k_cvt, w); run("cvt_sdwa", k_cvt_sdwa, w);
        run("pk_fma_f16", k_pkf16, w); run("pk_fma_f32", k_pkf32, w); run("dot2_f32_f16", k_dot2, w); run("cvt_pk_f16", k_cvtpk, w);
        run("exp_f32", k_exp, w); run("fma_mixlo", k_mixlo, w);
    }
    return 0;
}
