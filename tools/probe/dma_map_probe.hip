// Probe (round 6): cost of one global_load_lds_dwordx4 (1 KB per wave instruction) by how its 64 lanes map onto memory, 8 waves per
// CU, rows drawn from an L2-resident set (1 KB rows, this wave's 128-byte head slice of each):
//   0  scattered: lane l -> row l & 7, 16-byte piece 2 g + e with g = 2 (l >> 5) + ((l >> 3) & 1), e = (l >> 4) & 1   (k_attention_patch, first form)
//   1  row-contiguous: lane l -> row l >> 3, piece l & 7                                 (8 rows x 128 B, 8 consecutive lanes per row)
//   2  pairs: lanes 2 u, 2 u + 1 -> the two pieces of (row u & 7, octet u >> 3)          (32 B per lane pair)
//   3  one 1 KB row per instruction                                                      (fully contiguous)
//   4  two 512-byte half rows per instruction                                            (the gather kernel's shape)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k_dma(const float* __restrict__ x, int iters, int row_mask, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8][8 * 1040];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds[wave];
    int rsel, piece;   // which of the instruction's rows this lane reads, and the float offset inside the 1 KB row
    if (MODE == 0) { const int g = ((lane >> 5) << 1) | ((lane >> 3) & 1), e = (lane >> 4) & 1; rsel = lane & 7; piece = wave * 32 + (2 * g + e) * 4; }
    else if (MODE == 1) { rsel = lane >> 3; piece = wave * 32 + (lane & 7) * 4; }
    else if (MODE == 2) { const int u = lane >> 1; rsel = u & 7; piece = wave * 32 + ((u >> 3) * 2 + (lane & 1)) * 4; }
    else if (MODE == 3) { rsel = 0; piece = lane * 4; }
    else { rsel = lane >> 5; piece = (wave & 1) * 128 + (lane & 31) * 4; }
    unsigned h = blockIdx.x * 977u + wave * 131u + 7u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            // 8 rows near each other (a patch): base row + small offsets
            const int row = ((h >> 8) + rsel * 3 + (rsel >> 2) * 700) & row_mask;
            const float* src = x + (size_t)row * 256 + piece;
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + k * 1040);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    if (lds[wave][lane] == 0x77 && iters < 0) clk[1] = 1;
}
template <typename K>
void run(const char* name, K k, const float* x, unsigned long long* clk, int row_mask, int grid = 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, x, 50, row_mask, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, x, iters, row_mask, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * 8;   // wave instructions per CU
    printf("grid %3d %-40s rows %7d: %.3f ms  %.1f ns per wave instruction per CU (%.0f cycles at 2.4 GHz), %.1f TB/s chip\n", grid, name, row_mask + 1, ms, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, n * grid * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    float* x; hipMalloc(&x, bytes); hipMemset(x, 0, bytes);
    unsigned long long* clk; hipMalloc(&clk, 64);
    for (int rm : {(1 << 12) - 1, (1 << 15) - 1, (1 << 20) - 1}) {     // 4 MB (L2 of all XCDs), 32 MB, 1 GB (HBM)
        run("0 scattered (patch kernel, first form)", k_dma<0>, x, clk, rm);
        run("1 row-contiguous 8 x 128 B", k_dma<1>, x, clk, rm);
        run("2 lane pairs 32 B", k_dma<2>, x, clk, rm);
        run("3 one 1 KB row", k_dma<3>, x, clk, rm);
        run("4 two 512 B half rows", k_dma<4>, x, clk, rm);
    }
    // is the HBM-resident cost a per-CU limit (outstanding misses x latency) or the chip's bandwidth?  Fewer CUs at work:
    for (int grid : {16, 64, 128, 256}) run("1 row-contiguous 8 x 128 B", k_dma<1>, x, clk, (1 << 20) - 1, grid);
    for (int grid : {16, 64, 128, 256}) run("3 one 1 KB row", k_dma<3>, x, clk, (1 << 20) - 1, grid);
    return 0;
}
