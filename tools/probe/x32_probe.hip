// Probe (round 6): can ONE wavefront per SIMD keep the matrix pipe busy while it does a step's other work itself?  The x16 tails run two
// 16-token waves per SIMD whose product phases alternate (48 x v_mfma_f32_16x16x32_f16 + 32 fragment reads per 32 KB weight chunk and wave,
// ~1120 cycles each, 2600 per step).  A 32-token wave needs 48 x v_mfma_f32_32x32x16_f16 for the same chunk (1536 cycles of matrix pipe for
// twice the tokens) and the same 32 fragment reads - if the step's VALU / LDS work can be issued BETWEEN its MFMAs (source order pinned by
// sched_barrier: MFMA, a slice of filler, MFMA, ...).  FILL = dependent-chain VALU instructions per MFMA issued in its shadow.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
template <int OFF>
__device__ __forceinline__ void lds_read_frag(half8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF)); }
template <int N>
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N)); __builtin_amdgcn_sched_barrier(0); }

template <int FILL, int CHAINS, bool LDSW>
__device__ __forceinline__ void filler(float (&v)[8], float* stg, int lane, int f) {
    // CHAINS independent dependency chains of FMAs (a real rest-half has both: GELU polynomials are chains, conversions are wide)
#pragma unroll
    for (int i = 0; i < FILL; ++i) v[i % CHAINS] = fmaf(v[i % CHAINS], 1.0001f, 0.5f);
    if (LDSW && (f & 7) == 7) *reinterpret_cast<float4*>(stg + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

template <int DEPTH, int FILL, int CHAINS, bool LDSW, int f = 0>
struct Steps {
    static __device__ __forceinline__ void run(float16v& acc, unsigned addr, const half8 (&ah)[16], const half8 (&al)[16], half8 (&w)[DEPTH], float (&v)[8], float* stg, int lane) {
        constexpr int F = 32, s = f / 2;
        constexpr int issued = (DEPTH + f) < F ? (DEPTH + f) : F;
        lgkm_wait<issued - (f + 1) + (LDSW ? 1 : 0)>();       // (+1: a staging store may sit in the queue: stricter would be lgkmcnt(n) exact)
        if constexpr ((f & 1) == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], al[s], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler<FILL, CHAINS, false>(v, stg, lane, f);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], ah[s], acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], ah[s], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (f + DEPTH < F) lds_read_frag<(f + DEPTH) * 1024>(w[f % DEPTH], addr);
        filler<FILL, CHAINS, LDSW>(v, stg, lane, f);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (f + 1 < F) Steps<DEPTH, FILL, CHAINS, LDSW, f + 1>::run(acc, addr, ah, al, w, v, stg, lane);
    }
};
template <int DEPTH, int i = 0>
struct Prologue {
    static __device__ __forceinline__ void run(unsigned addr, half8 (&w)[DEPTH]) {
        lds_read_frag<i * 1024>(w[i], addr);
        if constexpr (i + 1 < DEPTH) Prologue<DEPTH, i + 1>::run(addr, w);
    }
};

template <int FILL, int CHAINS, bool LDSW, bool BAR>
__global__ __launch_bounds__(256) void k_x32(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) _Float16 chunk[3][16384];
    __shared__ float stgs[4][256];
    for (int i = threadIdx.x; i < 3 * 16384; i += 256) (&chunk[0][0])[i] = (_Float16)(0.001f * (i & 7));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 ah[16], al[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { ah[s] = (half8)(_Float16)(0.5f + s); al[s] = (half8)(_Float16)(0.001f * s); }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * i + lane;
    float16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float r = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(&chunk[it % 3][0] + lane * 8);
        half8 w[8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        Prologue<8>::run(addr, w);
        Steps<8, FILL, CHAINS, LDSW>::run(acc, addr, ah, al, w, v, stgs[wave], lane);
        __builtin_amdgcn_sched_barrier(0);
        r += acc[it & 15];
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float q = r;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += v[i];
    if (q == 1.2345f) out[0] = q;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <typename K>
void run(const char* name, K k, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-64s %.3f ms  %.0f ns per chunk (128 tokens)  %.0f counter ticks per chunk  MFMA share %.0f %% of 2.5 PF\n", name, ms, ms * 1e6 / iters, (double)c / iters,
           100.0 * iters * 48 * 4 * 256 * 32768.0 / (ms * 1e-3) / 2.5e15);
}
int main() {
    float* out; hipMalloc(&out, 1024);
    unsigned long long* clk; hipMalloc(&clk, 64);
    printf("one 32-token wave per SIMD, 48 x mfma_32x32x16 + 32 fragment reads per chunk; filler = VALU FMAs per fragment step (x 32 + x 16 per chunk)\n");
    run("no filler", k_x32<0, 1, false, false>, out, clk);
    run("no filler, barrier per chunk", k_x32<0, 1, false, true>, out, clk);
    run("4 FMAs per slot (192 per chunk), 4 chains", k_x32<4, 4, false, false>, out, clk);
    run("8 FMAs per slot (384 per chunk), 4 chains", k_x32<8, 4, false, false>, out, clk);
    run("8 FMAs per slot, 1 chain", k_x32<8, 1, false, false>, out, clk);
    run("12 FMAs per slot (576 per chunk), 4 chains", k_x32<12, 4, false, false>, out, clk);
    run("8 FMAs per slot, 4 chains, staging stores, barrier per chunk", k_x32<8, 4, true, true>, out, clk);
    run("16 FMAs per slot (768 per chunk), 8 chains", k_x32<16, 8, false, false>, out, clk);
    return 0;
}
