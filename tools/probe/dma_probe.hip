// Probe: how fast can a CU stream an L2-resident weight image into LDS with global_load_lds_dwordx4 (the chain kernels' weight
// ring), and how fast can its waves read the fragments back (ds_read_b128)?  One 512-thread workgroup per CU, 32 KB chunks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int CHUNK_BYTES = 32768, N_CHUNKS = 48, WAVES = 8;

template <int DEPTH, bool READBACK, int THREADS>
__global__ __launch_bounds__(THREADS) void k_dma(const char* __restrict__ w, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char ring[3 * CHUNK_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int NW = THREADS / 64, PPT = CHUNK_BYTES / 16 / THREADS;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring;
    auto issue = [&](int c) {
        const char* src = w + (size_t)(c % N_CHUNKS) * CHUNK_BYTES;
        const unsigned slot = lds_base + (c % 3) * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int piece0 = (i * NW + wave) * 64;
            const uint4* g = reinterpret_cast<const uint4*>(src) + piece0 + lane;
            const unsigned dst = __builtin_amdgcn_readfirstlane(slot + piece0 * 16);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    };
    float acc = 0.f;
    const int total = iters * N_CHUNKS;
    for (int d = 0; d < DEPTH; ++d) issue(d);
    for (int c = 0; c < total; ++c) {
        if (c + DEPTH < total) issue(c + DEPTH);
        // chunk c must have landed: all but the newest DEPTH requests (each PPT instructions)
        if (c + DEPTH < total) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * PPT) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (READBACK) {
            // 32 fragments of 1 KB per wave, as the chain kernels read them (lane-linear ds_read_b128)
            const uint4* f = reinterpret_cast<const uint4*>(ring + (c % 3) * CHUNK_BYTES) + lane;
            uint4 x = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint4 v = f[k * 64];
                x.x ^= v.x; x.y ^= v.y; x.z ^= v.z; x.w ^= v.w;
            }
            acc += (float)(x.x ^ x.y ^ x.z ^ x.w);
            __syncthreads();
        }
    }
    if (acc == 12345.f) out[0] = acc;
}
template <typename K>
void run(const char* name, K k, int threads, const char* w, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, w, out, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * N_CHUNKS * CHUNK_BYTES;      // per CU
    printf("%-46s %.3f ms  %.1f GB/s per CU  %.1f B/clk @2.4GHz  (%.1f TB/s chip)\n", name, ms, bytes / ms / 1e6, bytes / (ms * 2.4e6), bytes * 256 / ms / 1e9);
}
int main() {
    char* w; hipMalloc(&w, (size_t)N_CHUNKS * CHUNK_BYTES); hipMemset(w, 0, (size_t)N_CHUNKS * CHUNK_BYTES);
    float* out; hipMalloc(&out, 1024);
    run("dma depth 1, 8 waves, no readback", k_dma<1, false, 512>, 512, w, out);
    run("dma depth 2, 8 waves, no readback", k_dma<2, false, 512>, 512, w, out);
    run("dma depth 2, 4 waves, no readback", k_dma<2, false, 256>, 256, w, out);
    run("dma depth 2, 8 waves, 32 KB readback per wave", k_dma<2, true, 512>, 512, w, out);
    run("dma depth 2, 4 waves, 32 KB readback per wave", k_dma<2, true, 256>, 256, w, out);
    return 0;
}
