// does gfx950 execute scalar atomics (s_atomic_add ... glc)?  llvm-mc assembles them; this checks the hardware:
// 256 workgroups x 4 waves each draw 100 tickets from one counter; every ticket must come out exactly once.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/satomic_probe.hip -o tools/probe/satomic_probe.bin && timeout 60 tools/probe/satomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* ctr, int* seen, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 100; ++i) {
        int v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(&seen[v], 1);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = __builtin_amdgcn_s_memtime() - t0;
}
// the same with the result landing in flat_scratch_lo (SGPR 102: never allocated by the compiler; this kernel has no scratch), picked up
// later: the asynchronous form - nothing the register allocator does between issue and pick-up can see a stale value
template <int REG>
__global__ void k_async(int* ctr, int* seen) {
    for (int i = 0; i < 100; ++i) {
        if (REG == 0) asm volatile("s_mov_b32 flat_scratch_lo, 1\n\ts_atomic_add flat_scratch_lo, %0, 0x0 glc" :: "s"(ctr) : "memory");
        else asm volatile("s_mov_b32 xnack_mask_lo, 1\n\ts_atomic_add xnack_mask_lo, %0, 0x0 glc" :: "s"(ctr) : "memory");
        float a = (float)threadIdx.x;
        for (int j = 0; j < 50; ++j) a = a * 1.0001f + 0.5f;           // unrelated work in between
        int v;
        if (REG == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, flat_scratch_lo" : "=s"(v) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, xnack_mask_lo" : "=s"(v) :: "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(&seen[(unsigned)v < 102400u ? v : 102400], a > 1e30f ? 2 : 1);
    }
}
__global__ void k_lat(int* ctr, unsigned long long* clk) {     // one wave, nobody else: the latency of a pull
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int v = 1;
    for (int i = 0; i < 1000; ++i) { v = 1; asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory"); }
    if (threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memrealtime() - t0; clk[1] = v; }
}
int main() {
    const int n = 256 * 4 * 100;
    int *ctr, *seen; unsigned long long* clk;
    hipMalloc(&ctr, 4); hipMalloc(&seen, n * 4 + 4); hipMalloc(&clk, 8);
    hipMemset(ctr, 0, 4); hipMemset(seen, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, ctr, seen, clk);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<int> h(n); int c; unsigned long long hc;
    hipMemcpy(h.data(), seen, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != 1;
    printf("counter %d (expected %d), tickets not seen exactly once: %d, 100 pulls under contention: %llu clocks (100 MHz)\n", c, n, bad, hc);
    fflush(stdout);
    int bad2 = 0;
    for (int reg = 1; reg >= 1; --reg) {   // reg 0 (flat_scratch_lo as the landing register) ends in a memory fault on gfx950
        hipMemset(ctr, 0, 4); hipMemset(seen, 0, n * 4 + 4);
        if (reg == 0) hipLaunchKernelGGL(k_async<0>, dim3(256), dim3(256), 0, 0, ctr, seen);
        else hipLaunchKernelGGL(k_async<1>, dim3(256), dim3(256), 0, 0, ctr, seen);
        e = hipDeviceSynchronize();
        int out_of_range = 0;
        hipMemcpy(h.data(), seen, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost); hipMemcpy(&out_of_range, seen + n, 4, hipMemcpyDeviceToHost);
        int b2 = 0; for (int i = 0; i < n; ++i) b2 += h[i] != 1;
        printf("%s landing, picked up later: %s, counter %d, tickets not seen exactly once: %d, out of range: %d\n", reg ? "xnack_mask_lo" : "flat_scratch_lo", hipGetErrorString(e), c, b2, out_of_range);
        fflush(stdout);
        if (reg == 1) bad2 = b2;
    }
    unsigned long long* clk2; hipMalloc(&clk2, 16);
    hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, ctr, clk2);
    hipDeviceSynchronize();
    unsigned long long hl[2]; hipMemcpy(hl, clk2, 16, hipMemcpyDeviceToHost);
    printf("uncontended pull: %.0f ns each (1000 in a row, last ticket %llu)\n", hl[0] * 10.0 / 1000, hl[1]);
    return bad != 0 || c != n || bad2 != 0;
}
