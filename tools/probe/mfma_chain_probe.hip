// Probe (round 4): issue rate of v_mfma_f32_16x16x32_f16 for a lone wavefront per SIMD when consecutive MFMAs accumulate into the
// same registers (the x16 tails' products: 24 in a row per row tile) against 2 / 4 independent accumulators in rotation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_chain(float* out, int iters, unsigned long long* clk) {
    half8 a = (half8)(_Float16)(0.001f * (threadIdx.x & 7)), b = (half8)(_Float16)0.5f;
    float4v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (float4v)(0.f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 48; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0];
    if (r == 1.2345f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
// the x16 product stream: 32 fragments (1 KB each) of a 32 KB LDS chunk, DEPTH ds_read_b128 in flight, 3 MFMAs per fragment pair
template <int DEPTH, int WAVES, bool MFMA>
__global__ __launch_bounds__(WAVES * 64) void k_frag(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) half8 chunk[2048];
    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) chunk[i] = (half8)(_Float16)(0.001f * (i & 7));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    half8 b = (half8)(_Float16)0.5f;
    float4v acc[2] = {(float4v)(0.f), (float4v)(0.f)};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        half8 w[DEPTH];
#pragma unroll
        for (int f = 0; f < DEPTH; ++f) w[f] = chunk[f * 64 + lane];
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            const half8 v = w[f % DEPTH];
            if (MFMA) {
                if ((f & 1) == 0) {
                    acc[f / 16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, b, acc[f / 16], 0, 0, 0);
                    acc[f / 16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, b, acc[f / 16], 0, 0, 0);
                } else acc[f / 16] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, b, acc[f / 16], 0, 0, 0);
            } else acc[f / 16][0] += (float)v[0] + (float)v[7];
            __builtin_amdgcn_sched_barrier(0);
            if (f + DEPTH < 32) w[f % DEPTH] = chunk[(f + DEPTH) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const float r = acc[0][0] + acc[1][1];
    if (r == 1.2345f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename K>
void run(const char* name, K k, int threads, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)iters * 48;
    printf("%-44s %.3f ms  %.2f ns per MFMA per wave  (%.1f counter ticks)  => %.0f TFLOP/s chip\n", name, ms, ms * 1e6 / per_wave, (double)c / per_wave,
           per_wave * (threads / 64) * 256 * 16384.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 1024);
    unsigned long long* clk; hipMalloc(&clk, 64);
    run("1 accumulator, 1 wave / SIMD", k_chain<1, 4>, 256, out, clk);
    run("2 accumulators, 1 wave / SIMD", k_chain<2, 4>, 256, out, clk);
    run("4 accumulators, 1 wave / SIMD", k_chain<4, 4>, 256, out, clk);
    run("1 accumulator, 2 waves / SIMD", k_chain<1, 8>, 512, out, clk);
    run("2 accumulators, 2 waves / SIMD", k_chain<2, 8>, 512, out, clk);
    run("4 accumulators, 2 waves / SIMD", k_chain<4, 8>, 512, out, clk);
    printf("--- 32 fragment reads + 48 MFMAs per chunk (per-MFMA figures; 48 per chunk) ---\n");
    run("frag stream depth 8, 1 wave / SIMD", k_frag<8, 4, true>, 256, out, clk);
    run("frag stream depth 16, 1 wave / SIMD", k_frag<16, 4, true>, 256, out, clk);
    run("frag stream depth 8, 2 waves / SIMD", k_frag<8, 8, true>, 512, out, clk);
    run("frag stream depth 16, 2 waves / SIMD", k_frag<16, 8, true>, 512, out, clk);
    run("reads only depth 8, 1 wave / SIMD", k_frag<8, 4, false>, 256, out, clk);
    run("reads only depth 8, 2 waves / SIMD", k_frag<8, 8, false>, 512, out, clk);
    return 0;
}
