// Which clock does s_memtime count, and what does the core run at?  A single wave runs a dependent chain of N v_add_f32
// (4 cycles each at issue) bracketed by s_memtime / s_memrealtime (100 MHz), the host times the launch with events;
// then the same chain on every CU with MFMA + LDS traffic alongside, to see the clock under load.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(unsigned long long* out, int n, float seed) {
    float x = seed;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(x));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void k_load(unsigned long long* out, int n, float seed) {
    half8 a = (half8)(_Float16)seed, b = a;
    float16v acc = (float16v)(0.f);
    float x = seed;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(x));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)(x + acc[0]); }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    unsigned long long h[3];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        const int n = 2000000;
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, n, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("idle chip, one wave: %d x 4 dependent v_add: s_memtime %llu ticks, s_memrealtime %llu ticks (100 MHz = %.3f ms), events %.3f ms -> s_memtime %.1f MHz, %.2f ticks per v_add\n",
               n, h[0], h[1], h[1] / 1e5, ms, h[0] / (h[1] / 100.0), (double)h[0] / (4.0 * n));
    }
    for (int rep = 0; rep < 2; ++rep) {
        const int n = 400000;
        hipEventRecord(e0); hipLaunchKernelGGL(k_load, dim3(256 * 8), dim3(256), 0, 0, d, n, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("loaded chip (8 x 256-thread blocks per CU, MFMA + VALU): s_memtime %llu, realtime %llu (%.3f ms), events %.3f ms -> s_memtime %.1f MHz; MFMA rate %.1f TFLOP/s\n",
               h[0], h[1], h[1] / 1e5, ms, h[0] / (h[1] / 100.0), 256.0 * 8 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
