// Probe: HBM write rate of the chain kernels' store patterns (f16 rows of C = 256 per token).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int C = 256;
// PATTERN 0: row per instruction (64 lanes x 16 B = 2 tokens x 512 B): fully coalesced
// PATTERN 1: k_ln_qkv: lane (m, hi) stores 2 x 16 B per 32-channel tile, tiles 0..7 one after the other
// PATTERN 2: like 1 but 8-byte stores (the old 4 x dwordx2 per tile)
template <int PATTERN>
__global__ __launch_bounds__(256, 2) void k_store(_Float16* __restrict__ y, int P, int planes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hi = lane >> 5;
    half8 v;
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(lane + i);
    for (int pl = 0; pl < planes; ++pl) {
        _Float16* yp = y + (size_t)pl * P * C;
        if (PATTERN == 0) {
            const int tok0 = blockIdx.x * 128 + wave * 32;
            for (int i = 0; i < 16; ++i) {
                const int tok = tok0 + 2 * i + (lane >> 5);
                if (tok < P) *reinterpret_cast<half8*>(yp + (size_t)tok * C + (lane & 31) * 8) = v;
            }
        } else {
            const int tok = blockIdx.x * 128 + wave * 32 + m;
            if (tok < P)
                for (int t = 0; t < 8; ++t) {
                    _Float16* o = yp + (size_t)tok * C + 32 * t + 8 * hi;
                    if (PATTERN == 1) {
                        *reinterpret_cast<half8*>(o) = v;
                        *reinterpret_cast<half8*>(o + 16) = v;
                    } else {
                        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                        half4 h = {v[0], v[1], v[2], v[3]};
                        _Float16* o2 = yp + (size_t)tok * C + 32 * t + 4 * hi;
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<half4*>(o2 + 8 * j) = h;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
        }
    }
}
template <typename K>
void run(const char* name, K k, _Float16* y, int P, int planes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (P + 127) / 128;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, y, P, planes);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, y, P, planes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, (double)P * C * 2 * planes / ms / 1e9);
}
int main() {
    const int P = 5 * 140800, planes = 3;
    _Float16* y; hipMalloc(&y, (size_t)P * C * 2 * planes);
    run("row per instruction (coalesced)", k_store<0>, y, P, planes);
    run("tile pattern, 2 x 16 B per lane", k_store<1>, y, P, planes);
    run("tile pattern, 4 x 8 B per lane", k_store<2>, y, P, planes);
    return 0;
}
