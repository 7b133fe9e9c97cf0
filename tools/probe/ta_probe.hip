// Probe (round 4): cost of a vector-memory wave instruction on the texture path, 4 waves per CU (one per SIMD, the attention
// kernel's loader role): structured buffer loads of 16 / 8 / 4 bytes per lane, in range (L1-resident rows) or out of range
// (index -1, or an empty descriptor).  Reports cycles per wave instruction per CU (s_memtime) and ns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
__device__ uint4v ld4(int4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");
__device__ uint2v ld2(int4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2i32");
__device__ unsigned ld1(int4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.i32");

template <int MODE>
__global__ __launch_bounds__(256) void k_ta(const float* __restrict__ x, unsigned* out, int iters, int n_tok, unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long a = (unsigned long long)x;
    int4v rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((unsigned)(a >> 32) & 0xffffu) | (1024u << 16)));
    rs.z = (MODE == 3) ? 0 : n_tok;
    rs.w = 0x00020000;
    // 32 lanes per token row (16 B each, 512 B of the row), 2 tokens per wave instruction; the tokens walk a small L1-resident set
    const int off = (lane & 31) * 16;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint4v r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int tok = (MODE == 0 || MODE == 4 || MODE == 5) ? ((it * 8 + k) * 2 + (lane >> 5) + wave * 4) & 15 : -1;
            if (MODE == 4) { const uint2v v = ld2(rs, tok, off, 0, 0); r[k] = uint4v{v.x, v.y, 0, 0}; }
            else if (MODE == 2 || MODE == 5) { r[k] = uint4v{ld1(rs, tok, off, 0, 0), 0, 0, 0}; }
            else r[k] = ld4(rs, tok, off, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= r[k].x ^ r[k].y ^ r[k].z ^ r[k].w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345u) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <typename K>
void run(const char* name, K k, const float* x, unsigned* out, unsigned long long* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, x, out, 100, 64, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, x, out, iters, 64, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * 4;   // wave instructions per CU
    printf("%-52s %.3f ms  %.1f ns per wave instruction per CU   (%.1f counter ticks)\n", name, ms, ms * 1e6 / n, (double)c / n);
}
int main() {
    float* x; hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20);
    unsigned* out; hipMalloc(&out, 1024);
    unsigned long long* clk; hipMalloc(&clk, 64);
    run("dwordx4, in range (L1 hits)", k_ta<0>, x, out, clk);
    run("dwordx4, index -1", k_ta<1>, x, out, clk);
    run("dword, index -1", k_ta<2>, x, out, clk);
    run("dwordx4, empty descriptor", k_ta<3>, x, out, clk);
    run("dwordx2, in range (L1 hits)", k_ta<4>, x, out, clk);
    run("dword, in range (L1 hits)", k_ta<5>, x, out, clk);
    return 0;
}
