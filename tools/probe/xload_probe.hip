// Probe: HBM read rate of the chain kernels' token-per-lane x load pattern vs a row-per-wave pattern, at the chain
// kernels' occupancy (256-thread workgroups, 2 per CU through a 40 KB LDS reservation).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 256;
template <int PATTERN, int OCC_LDS>
__global__ __launch_bounds__(256, 2) void k_load(const float* __restrict__ x, float* __restrict__ out, int P) {
    __shared__ float pad[OCC_LDS / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hi = lane >> 5;
    float4 v[8][4];
    if (PATTERN == 0) {
        const int tok = min(blockIdx.x * 128 + wave * 32 + m, P - 1);
        const float* xp = x + (size_t)tok * C + 4 * hi;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = *reinterpret_cast<const float4*>(xp + 32 * b + 8 * j);
    } else {
        // row per instruction: 64 lanes x 16 B = one token row; 32 instructions = 32 tokens
        const int tok0 = blockIdx.x * 128 + wave * 32;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tok = min(tok0 + b * 4 + j, P - 1);
                v[b][j] = *reinterpret_cast<const float4*>(x + (size_t)tok * C + lane * 4);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[b][j].x + v[b][j].y + v[b][j].z + v[b][j].w;
    if (threadIdx.x == 0) pad[0] = s;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + pad[0] * 0.f;
}
template <typename K>
void run(const char* name, K k, const float* x, float* out, int P) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (P + 127) / 128;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, x, out, P);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, x, out, P);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, (double)P * C * 4 / ms / 1e9);
}
int main() {
    const int P = 5 * 140800;
    float* x; hipMalloc(&x, (size_t)P * C * 4); hipMemset(x, 0, (size_t)P * C * 4);
    float* out; hipMalloc(&out, (size_t)((P + 127) / 128) * 256 * 4);
    run("token-per-lane, 2 WG/CU", k_load<0, 40960>, x, out, P);
    run("row-per-instruction, 2 WG/CU", k_load<1, 40960>, x, out, P);
    run("token-per-lane, LDS-free", k_load<0, 16>, x, out, P);
    run("row-per-instruction, LDS-free", k_load<1, 16>, x, out, P);
    return 0;
}
