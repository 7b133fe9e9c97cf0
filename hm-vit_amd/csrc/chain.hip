// Register-resident token chains (f16-operand mode): every wavefront owns 32 tokens and keeps
// their activations in registers across LayerNorm and up to three GEMMs; only weights go through
// LDS.  Two kernels:
//
//   k_ln_qkv   x (f32) -> HeteroLayerNorm -> Q | K' | V' projections -> f16 planes
//              (base_transformer.py:138-177 + hetero_fusion.py:111-140 with the relation folds)
//   k_out_ffn  O (f16), x (f32) -> a_linears + bias + residual -> HeteroLayerNorm ->
//              Linear + GELU -> Linear + bias + residual -> x (f32)
//              (hetero_fusion.py:142-152, :399-402 / :439-442, base_transformer.py:129-136,180-192);
//              with the flags off it is mlp_head (bevformer_point_pillar_hetero.py:37,47-48)
//              storing straight to NCHW.
//
// Layout trick: all products are computed transposed, D'[n][m] = sum_k W[n][k] act[m][k], with
// v_mfma_f32_32x32x16_f16 (A operand = weight rows, B operand = tokens).  Lane (m = lane & 31,
// hi = lane >> 5) then owns, for token m, the channels 32 b + 8 j + 4 hi + i (b = 32-channel
// block, j < 4, i < 4) of every activation -- inputs are loaded in that pattern, accumulators
// come out in it, and after a float->half conversion they ARE the B operand of the next GEMM
// (k-slot 4 jj + i of step kk = 2 b + s holds channel 32 b + 16 s + 8 jj + 4 hi + i; the host
// permutes the weight columns to match, hm-vit_amd/weights.py:weight_image).  No activation ever
// takes an LDS round trip, and NCHW input/output is a 128-byte-per-channel access in this layout.
//
// Weights stream through a 2-deep LDS ring in "chunks" of one 32-row tile x all K (KK KiB), stored
// in global memory in exactly fragment order, so staging is a linear copy and every ds_read_b128 is
// lane-linear (conflict-free).  Loads of chunk c+1 are issued before the MFMAs of chunk c and
// written to LDS after them (one barrier per chunk).
#include <cstring>
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

// 4 wavefronts x 32 tokens per workgroup: two workgroups share a CU (256 VGPRs each), so one
// group's HBM load / store phases overlap the other's MFMA phase
constexpr int CHAIN_THREADS_C = 256;

// SP ("split" precision mode): every fp32 product runs on the f16 matrix pipe as x = hi + lo (hi = f16(x), lo = f16(x - hi)),
// three MFMAs per product (w_hi a_hi + w_hi a_lo + w_lo a_hi, f32 accumulate; the lo x lo term is below 2^-22 relative).  A
// weight chunk then holds, per k-step, the hi fragment followed by the lo fragment (weights.py weight_image(split=True)), the
// activations live in registers as hi / lo operand pairs, and everything that goes to memory between kernels (Q, K', V', O,
// the residual stream) is f32.  The register budget doubles, so these instantiations run one workgroup per CU.
template <int C, bool SP = false>
struct ChainCfg {
    static constexpr int NB = C / 32;             // 32-channel blocks
    static constexpr int KK = C / 16;             // MFMA k-steps over C
    static constexpr int NT = C / 32;             // 32-row output tiles
    static constexpr int FR = SP ? 2 : 1;         // weight fragments per k-step
    static constexpr int CHUNK_HALVES = KK * 512 * FR; // one chunk: KK (x 2) fragments of 64 lanes x 8 halves
    static constexpr int PIECES = KK * 64 * FR;   // 16-byte pieces per chunk
    static constexpr int PPT = (PIECES + CHAIN_THREADS_C - 1) / CHAIN_THREADS_C;
};

constexpr int CHAIN_THREADS = CHAIN_THREADS_C;
constexpr int CHAIN_WAVES = CHAIN_THREADS / 64;
constexpr int CHAIN_TOKENS = CHAIN_WAVES * 32;

// ---- weight chunk ring: global -> LDS DMA (no staging registers) ----
// The chunk is stored in global memory in LDS-image order, so wave w copies pieces
// [(i*8 + w) * 64, +64) with one global_load_lds_dwordx4 each (LDS destination = wave-uniform
// base + lane * 16).  Completion: the __syncthreads() that ends the iteration (hipcc drains
// vmcnt(0) before a barrier while an LDS-DMA is in flight).
template <int C, bool SP = false>
__device__ __forceinline__ void stage_chunk(const half_t* __restrict__ chunk, half_t* lds_buf) {
    // Inline asm on purpose: hipcc must not count this DMA (it would drain vmcnt(0), i.e. also the
    // output stores in flight, before every ds_read of the ring).  Completion is waited for by
    // dma_wait() below, placed where the only other outstanding VMEM ops are long-issued stores.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_buf;
#pragma unroll
    for (int i = 0; i < ChainCfg<C, SP>::PPT; ++i) {
        const int piece0 = (i * CHAIN_WAVES + wave) * 64;
        if (piece0 < ChainCfg<C, SP>::PIECES) {
            const uint4* gsrc = reinterpret_cast<const uint4*>(chunk) + piece0 + lane;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + piece0 * 16);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
    }
}

// wait for this wave's DMA pieces (and any older store), then workgroup barrier: afterwards the
// whole chunk is visible to every wave and the previous buffer may be overwritten
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef HMVIT_EXP_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ half8 lds_frag(const half_t* buf, int frag, int lane) {
    return *reinterpret_cast<const half8*>(buf + (frag * 64 + lane) * 8);
}

// GELU(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|) with Phi(-t) = exp2(-Q(t)), Q = degree-7 Chebyshev fit of
// -log2 Phi(-t) on [0, 6] (Phi(-6) = 1e-9; t is clamped there).  |error| <= 6.4e-7 over the reals (f32 round-off class;
// the result is rounded to f16 next): seven multiply-adds and one v_exp_f32, no division, no select -- this sits between
// the two FFN matrix products of every hidden slice and was a third of the chain tail's VALU work in its erf form.
// two values at a time: the degree-7 polynomial as packed multiply-adds
__device__ __forceinline__ float2v gelu_f2(float2v x) {
    const float2v ax = {fabsf(x.x), fabsf(x.y)};
    const float2v t = {fminf(ax.x, 6.f), fminf(ax.y, 6.f)};
    float2v q = __builtin_elementwise_fma((float2v)(1.889626219e-06f), t, (float2v)(-6.268139987e-05f));
    q = __builtin_elementwise_fma(q, t, (float2v)(9.388679173e-04f));
    q = __builtin_elementwise_fma(q, t, (float2v)(-8.539461531e-03f));
    q = __builtin_elementwise_fma(q, t, (float2v)(5.402068794e-02f));
    q = __builtin_elementwise_fma(q, t, (float2v)(4.584097862e-01f));
    q = __builtin_elementwise_fma(q, t, (float2v)(1.151269197e+00f));
    q = __builtin_elementwise_fma(q, t, (float2v)(9.999943376e-01f));
    const float2v e = {__builtin_amdgcn_exp2f(-q.x), __builtin_amdgcn_exp2f(-q.y)};
    const float2v pos = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return __builtin_elementwise_fma(-ax, e, pos);
}
__device__ __forceinline__ float gelu_f(float x) {
    const float t = fminf(fabsf(x), 6.f);
    float q = fmaf(1.889626219e-06f, t, -6.268139987e-05f);
    q = fmaf(q, t, 9.388679173e-04f);
    q = fmaf(q, t, -8.539461531e-03f);
    q = fmaf(q, t, 5.402068794e-02f);
    q = fmaf(q, t, 4.584097862e-01f);
    q = fmaf(q, t, 1.151269197e+00f);
    q = fmaf(q, t, 9.999943376e-01f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(-q), fmaxf(x, 0.f));
}

// One chunk = KK weight fragments.  hipcc re-serialises a source-level prefetch into
// "ds_read x2, wait, mfma, wait, mfma" (it minimises registers), exposing the LDS latency on every
// pair of MFMAs; the reads are therefore issued from inline asm (invisible to its scheduler) DEPTH
// ahead of their MFMA, with hand-counted s_waitcnt lgkmcnt and a sched_barrier after every wait so that
// no compiler instruction (in particular no SMEM load, which shares the counter and returns out of
// order) can move into the counted region.
template <int OFF>
__device__ __forceinline__ void lds_read_frag(half8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
    __builtin_amdgcn_sched_barrier(0);
}

template <int KK, int DEPTH, int kk = 0>
struct MmaSteps {
    static __device__ __forceinline__ void run(float16v& acc, unsigned addr, const half8 (&act)[KK], half8 (&w)[DEPTH]) {
        constexpr int outstanding = (KK - kk - 1) < (DEPTH - 1) ? (KK - kk - 1) : (DEPTH - 1);
        lgkm_wait<outstanding>();
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[kk % DEPTH], act[kk], acc, 0, 0, 0);
        if constexpr (kk + DEPTH < KK) {
            __builtin_amdgcn_sched_barrier(0);
            lds_read_frag<(kk + DEPTH) * 1024>(w[kk % DEPTH], addr);
        }
        if constexpr (kk + 1 < KK) MmaSteps<KK, DEPTH, kk + 1>::run(acc, addr, act, w);
    }
};
// split mode: 2 KK fragments (hi, lo per k-step); the hi fragment feeds two MFMAs (x a_hi, x a_lo), the lo fragment one
template <int KK, int DEPTH, int f = 0>
struct MmaStepsSplit {
    static constexpr int F = 2 * KK;
    static __device__ __forceinline__ void run(float16v& acc, unsigned addr, const half8 (&ah)[KK], const half8 (&al)[KK],
                                               half8 (&w)[DEPTH]) {
        constexpr int outstanding = (F - f - 1) < (DEPTH - 1) ? (F - f - 1) : (DEPTH - 1);
        lgkm_wait<outstanding>();
        if constexpr ((f & 1) == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], al[f / 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], ah[f / 2], acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[f % DEPTH], ah[f / 2], acc, 0, 0, 0);
        }
        if constexpr (f + DEPTH < F) {
            __builtin_amdgcn_sched_barrier(0);
            lds_read_frag<(f + DEPTH) * 1024>(w[f % DEPTH], addr);
        }
        if constexpr (f + 1 < F) MmaStepsSplit<KK, DEPTH, f + 1>::run(acc, addr, ah, al, w);
    }
};
template <int DEPTH, int i = 0>
struct MmaPrologue {
    static __device__ __forceinline__ void run(unsigned addr, half8 (&w)[DEPTH]) {
        lds_read_frag<i * 1024>(w[i], addr);
        if constexpr (i + 1 < DEPTH) MmaPrologue<DEPTH, i + 1>::run(addr, w);
    }
};

template <int KK, int DEPTH>
__device__ __forceinline__ void mma_chunk(float16v& acc, const half_t* buf, const half8 (&act)[KK], int lane) {
    static_assert(KK % DEPTH == 0 && DEPTH <= 15, "");
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(buf + lane * 8);
    half8 w[DEPTH];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // nothing of the compiler's in the LGKM queue
    __builtin_amdgcn_sched_barrier(0);
    MmaPrologue<DEPTH>::run(addr, w);
    MmaSteps<KK, DEPTH>::run(acc, addr, act, w);
    __builtin_amdgcn_sched_barrier(0);
}
template <int KK, int DEPTH>
__device__ __forceinline__ void mma_chunk_split(float16v& acc, const half_t* buf, const half8 (&ah)[KK], const half8 (&al)[KK], int lane) {
    static_assert((2 * KK) % DEPTH == 0 && DEPTH <= 15, "");
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(buf + lane * 8);
    half8 w[DEPTH];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MmaPrologue<DEPTH>::run(addr, w);
    MmaStepsSplit<KK, DEPTH>::run(acc, addr, ah, al, w);
    __builtin_amdgcn_sched_barrier(0);
}
// operand pair of the current precision mode: hi always, lo only in split mode
template <int KK, bool SP>
struct Operands {
    half8 hi[KK];
    half8 lo[SP ? KK : 1];
};
template <int KK, int DEPTH, bool SP>
__device__ __forceinline__ void mma_chunk_op(float16v& acc, const half_t* buf, const Operands<KK, SP>& a, int lane) {
    if constexpr (SP) mma_chunk_split<KK, DEPTH>(acc, buf, a.hi, a.lo, lane);
    else mma_chunk<KK, DEPTH>(acc, buf, a.hi, lane);
}

__device__ __forceinline__ float pair_sum(float v) { return xor32_sum(v); }

// LayerNorm of the token owned by lane pair (m, hi = 0/1); v[b][j] holds channels 32b+8j+4hi+(0..3)
template <int C>
__device__ __forceinline__ void layer_norm_regs(float4 (&v)[C / 32][4], const float* __restrict__ lg,
                                                const float* __restrict__ lb, int hi) {
    constexpr int NB = C / 32;
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (v[b][j].x + v[b][j].y) + (v[b][j].z + v[b][j].w);
    const float mean = pair_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[b][j].x -= mean; v[b][j].y -= mean; v[b][j].z -= mean; v[b][j].w -= mean;
            q += (v[b][j].x * v[b][j].x + v[b][j].y * v[b][j].y) + (v[b][j].z * v[b][j].z + v[b][j].w * v[b][j].w);
        }
    const float rstd = rsqrtf(pair_sum(q) * (1.f / C) + 1e-5f);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 32 * b + 8 * j + 4 * hi;
            const float4 g = *reinterpret_cast<const float4*>(lg + c);
            const float4 be = *reinterpret_cast<const float4*>(lb + c);
            v[b][j].x = v[b][j].x * rstd * g.x + be.x;
            v[b][j].y = v[b][j].y * rstd * g.y + be.y;
            v[b][j].z = v[b][j].z * rstd * g.z + be.z;
            v[b][j].w = v[b][j].w * rstd * g.w + be.w;
        }
}

// f32 activations -> MFMA B operands
template <int C, bool SP>
__device__ __forceinline__ void to_operands(const float4 (&v)[C / 32][4], Operands<C / 16, SP>& act) {
#pragma unroll
    for (int b = 0; b < C / 32; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8 h, l;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4 f = v[b][2 * s + jj];
                if constexpr (SP) {
                    split_h(f.x, h[4 * jj + 0], l[4 * jj + 0]); split_h(f.y, h[4 * jj + 1], l[4 * jj + 1]);
                    split_h(f.z, h[4 * jj + 2], l[4 * jj + 2]); split_h(f.w, h[4 * jj + 3], l[4 * jj + 3]);
                } else {
                    h[4 * jj + 0] = (half_t)f.x; h[4 * jj + 1] = (half_t)f.y;
                    h[4 * jj + 2] = (half_t)f.z; h[4 * jj + 3] = (half_t)f.w;
                }
            }
            act.hi[2 * b + s] = h;
            if constexpr (SP) act.lo[2 * b + s] = l;
        }
}

// LayerNorm straight from the accumulator layout into MFMA B operands (no f32 copy is kept)
template <int C, bool SP>
__device__ __forceinline__ void ln_acc_to_operands(const float16v (&xa)[C / 32], const float* __restrict__ lg,
                                                   const float* __restrict__ lb, int hi, Operands<C / 16, SP>& act) {
    constexpr int NT = C / 32;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += xa[t][e];
    const float mean = pair_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float dlt = xa[t][e] - mean;
            q = fmaf(dlt, dlt, q);
        }
    const float rstd = rsqrtf(pair_sum(q) * (1.f / C) + 1e-5f);
    const float shift = -mean * rstd;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            __builtin_amdgcn_sched_barrier(0);
            half8 h, l;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * sx + jj, c = 32 * t + 8 * j + 4 * hi;
                const float4 g = *reinterpret_cast<const float4*>(lg + c);
                const float4 be = *reinterpret_cast<const float4*>(lb + c);
                // x * rstd + shift, not (x - mean) * rstd: the latter shares (x - mean) with the variance pass and
                // hipcc then keeps all C centred values alive next to the accumulators (32 spilled registers)
                const float n0 = fmaf(xa[t][4 * j + 0], rstd, shift) * g.x + be.x;
                const float n1 = fmaf(xa[t][4 * j + 1], rstd, shift) * g.y + be.y;
                const float n2 = fmaf(xa[t][4 * j + 2], rstd, shift) * g.z + be.z;
                const float n3 = fmaf(xa[t][4 * j + 3], rstd, shift) * g.w + be.w;
                if constexpr (SP) {
                    split_h(n0, h[4 * jj + 0], l[4 * jj + 0]); split_h(n1, h[4 * jj + 1], l[4 * jj + 1]);
                    split_h(n2, h[4 * jj + 2], l[4 * jj + 2]); split_h(n3, h[4 * jj + 3], l[4 * jj + 3]);
                } else {
                    h[4 * jj + 0] = (half_t)n0; h[4 * jj + 1] = (half_t)n1;
                    h[4 * jj + 2] = (half_t)n2; h[4 * jj + 3] = (half_t)n3;
                }
            }
            asm volatile("" : "+v"(h));   // pin: the optimizer otherwise sinks the normalisation to the first use
            act.hi[2 * t + sx] = h;
            if constexpr (SP) {
                asm volatile("" : "+v"(l));
                act.lo[2 * t + sx] = l;
            }
        }
    // (the fences below keep hipcc from hoisting all 64 gamma / beta reads ahead of the arithmetic)
}

// load the 128 / 64 / 32 channels of token `tok` this lane owns (tok already clamped into range:
// one base pointer + compile-time offsets, no per-load branches)
template <int C>
__device__ __forceinline__ void load_token(const float* __restrict__ x, bool nchw, int P, int tok, int hi,
                                           float4 (&v)[C / 32][4]) {
    if (nchw) {
        const float* xp = x + (size_t)(4 * hi) * P + tok;
#pragma unroll
        for (int b = 0; b < C / 32; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* q = xp + (size_t)(32 * b + 8 * j) * P;
                v[b][j] = make_float4(q[0], q[P], q[2 * (size_t)P], q[3 * (size_t)P]);
            }
    } else {
        const float* xp = x + (size_t)tok * C + 4 * hi;
#pragma unroll
        for (int b = 0; b < C / 32; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b][j] = *reinterpret_cast<const float4*>(xp + 32 * b + 8 * j);
    }
}

// Coalesced stores of the projected tiles.  An accumulator lane owns one token and 2 x 8 channels of a 32-channel tile:
// stored directly, a wave instruction writes 32 separate 32-byte pieces (measured: 3.9 TB/s for this pattern against
// 6.0 TB/s for whole rows, tools/probe/xstore_probe.hip).  Instead four consecutive tiles (128 channels = 256 bytes per
// token) are parked in a per-wave LDS buffer and written out as rows: every store instruction then covers 4 tokens x 256
// contiguous bytes.  The buffer is private to the wave (LDS executes a wave's accesses in order: no barrier), rows are
// padded by 16 bytes so that the 32 token rows of a write do not fall into the same banks.
constexpr int STG_ROW = 128 + 8;              // halves per staged token row
constexpr int STG_WAVE = 32 * STG_ROW;        // halves per wave

__device__ __forceinline__ void stage_tile(half_t* stg, int m, int hi, int t4, const float16v& acc) {
    half_t* d = stg + m * STG_ROW + 32 * t4 + 8 * hi;   // rows in store order (weights.py store_row_order)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (half_t)acc[8 * s + i];
        *reinterpret_cast<half8*>(d + 16 * s) = h;
    }
}

// split mode: the projected tiles go to memory as f32.  Two consecutive tiles (64 channels = 256 bytes per token) share the same
// per-wave buffer: rows of 64 floats + 4 floats of padding (= STG_ROW halves), rows in store order as above.
__device__ __forceinline__ void stage_tile_f32(half_t* stg_h, int m, int hi, int t2, const float16v& acc) {
    float* d = reinterpret_cast<float*>(stg_h) + m * (STG_ROW / 2) + 32 * t2 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<float4*>(d + 16 * s) = make_float4(acc[8 * s], acc[8 * s + 1], acc[8 * s + 2], acc[8 * s + 3]);
        *reinterpret_cast<float4*>(d + 16 * s + 4) = make_float4(acc[8 * s + 4], acc[8 * s + 5], acc[8 * s + 6], acc[8 * s + 7]);
    }
}
// y: first token row of the wave in the output plane, already offset to the 64-channel group
__device__ __forceinline__ void flush_tiles_f32(const half_t* stg_h, float* y, int lane, int n_tok, int C) {
    const float* stg = reinterpret_cast<const float*>(stg_h);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int token = 4 * k + (lane >> 4), piece = lane & 15;
        const float4 v = *reinterpret_cast<const float4*>(stg + token * (STG_ROW / 2) + piece * 4);
        if (token < n_tok) *reinterpret_cast<float4*>(y + (size_t)token * C + piece * 4) = v;
    }
}

// y: first token row of the wave in the output plane, already offset to the 128-channel group
__device__ __forceinline__ void flush_tiles(const half_t* stg, half_t* y, int lane, int n_tok, int C) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int token = 4 * k + (lane >> 4), piece = lane & 15;
        const half8 v = *reinterpret_cast<const half8*>(stg + token * STG_ROW + piece * 8);
        if (token < n_tok) *reinterpret_cast<half8*>(y + (size_t)token * C + piece * 8) = v;
    }
}

// One projected 32-channel tile `t` of matrix output `y` (f16 plane, or f32 plane in split mode) for the wave's 32 tokens.
template <int C, bool SP, bool STAGED>
__device__ __forceinline__ void store_proj_tile(half_t* stg, void* y, int t, const float16v& acc, int m, int hi, int lane, int tok,
                                                bool valid, int tok_w, int P) {
    if constexpr (STAGED && SP) {
        stage_tile_f32(stg, m, hi, t & 1, acc);
        if ((t & 1) == 1) flush_tiles_f32(stg, reinterpret_cast<float*>(y) + (size_t)tok_w * C + 64 * (t >> 1), lane, P - tok_w, C);
    } else if constexpr (STAGED) {
        stage_tile(stg, m, hi, t & 3, acc);
        if ((t & 3) == 3) flush_tiles(stg, reinterpret_cast<half_t*>(y) + (size_t)tok_w * C + 128 * (t >> 2), lane, P - tok_w, C);
    } else if (valid) {
        // the images of these kernels order the rows of a tile so that the lane's 16 results are two runs
        // of 8 consecutive channels (weights.py store_row_order): 2 x 16-byte stores (f16), adjacent for the
        // lane pair of a token, instead of 4 x 8 (the store path is issue-bound)
        if constexpr (SP) {
            float* o = reinterpret_cast<float*>(y) + ((size_t)tok * C + 32 * t + 8 * hi);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                *reinterpret_cast<float4*>(o + 16 * s) = make_float4(acc[8 * s], acc[8 * s + 1], acc[8 * s + 2], acc[8 * s + 3]);
                *reinterpret_cast<float4*>(o + 16 * s + 4) = make_float4(acc[8 * s + 4], acc[8 * s + 5], acc[8 * s + 6], acc[8 * s + 7]);
            }
        } else {
            half_t* o = reinterpret_cast<half_t*>(y) + ((size_t)tok * C + 32 * t + 8 * hi);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 h;
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] = (half_t)acc[8 * s + i];
                *reinterpret_cast<half8*>(o + 16 * s) = h;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_ln_qkv
// ------------------------------------------------------------------------------------------
template <int C, bool SP>
__global__ __launch_bounds__(CHAIN_THREADS, SP ? 1 : 2) void k_ln_qkv(QkvParams p) {
    using Cfg = ChainCfg<C, SP>;
    constexpr int KK = Cfg::KK, NT = Cfg::NT;
    // ONE LDS object (a second one would make hipcc wait for the DMA before every ds_read)
    constexpr bool STAGED = C == 256;   // coalesced stores through LDS (stage_tile / flush_tiles)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * Cfg::CHUNK_HALVES + 4 * C + (STAGED ? CHAIN_WAVES * STG_WAVE : 0)];
    half_t* ring0 = smem;
    half_t* ring1 = smem + Cfg::CHUNK_HALVES;
    float* lnp = reinterpret_cast<float*>(smem + 2 * Cfg::CHUNK_HALVES);   // gamma[C], beta[C]

    const QkvJob& J = p.job[blockIdx.y];
    const int P = p.P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 31, hi = lane >> 5;
    const int tok = blockIdx.x * CHAIN_TOKENS + wave * 32 + m;
    const bool valid = tok < P;

    for (int i = threadIdx.x; i < C; i += CHAIN_THREADS) {
        lnp[i] = p.gamma[J.type * C + i];
        lnp[C + i] = p.beta[J.type * C + i];
    }
    const int n_chunks = J.n_mat * NT;
    stage_chunk<C, SP>(J.w[0], ring0);

    float4 v[C / 32][4];
    load_token<C>(J.x, p.in_nchw != 0, P, min(tok, P - 1), hi, v);
    if (p.in_nchw && valid && J.xs_out) {
        // first stage: also emit the token-major f32 residual stream (unless the stage's tail reads the input map itself)
        float* xo = J.xs_out + (size_t)tok * C + 4 * hi;
#pragma unroll
        for (int b = 0; b < C / 32; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(xo + 32 * b + 8 * j) = v[b][j];
    }
    dma_wait();
    __syncthreads();
    layer_norm_regs<C>(v, lnp, lnp + C, hi);
    Operands<KK, SP> act;
    to_operands<C, SP>(v, act);
    half_t* stg = smem + 2 * Cfg::CHUNK_HALVES + 4 * C + wave * STG_WAVE;
    const int tok_w = blockIdx.x * CHAIN_TOKENS + wave * 32;

    for (int c = 0; c < n_chunks; ++c) {
        const int mat = c / NT, t = c - mat * NT;
        const half_t* buf = (c & 1) ? ring1 : ring0;
        if (c + 1 < n_chunks) {
            const int c1 = c + 1, mat1 = c1 / NT, t1 = c1 - mat1 * NT;
            stage_chunk<C, SP>(J.w[mat1] + (size_t)t1 * Cfg::CHUNK_HALVES, (c & 1) ? ring0 : ring1);
        }
        float16v acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mma_chunk_op<KK, (KK < 8 ? KK : 8), SP>(acc, buf, act, lane);
        if constexpr (SP) acc *= J.c[mat];
#ifndef HMVIT_EXP_NOWAIT
        dma_wait();
#endif
#ifndef HMVIT_EXP_NOSTORE
        store_proj_tile<C, SP, STAGED>(stg, J.y[mat], t, acc, m, hi, lane, tok, valid, tok_w, P);
#endif
        wg_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// k_out_ffn
// ------------------------------------------------------------------------------------------
// The body serves two kernels: k_out_ffn (one stage's tail) and k_out_ffn_qkv, which appends the NEXT stage's
// LayerNorm + Q / K' / V' projections while the updated residual row x'' is still in the accumulators: the
// residual stream is then read once instead of twice per stage (and not written at all for agents that the
// pruned last stage only uses as K / V sources).  `qp` = the k_ln_qkv parameters of the next stage, job j of
// both lists is the same agent.
typedef int int4v __attribute__((ext_vector_type(4)));
__device__ float llvm_raw_buffer_load_f32(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

template <int C, bool OUTPROJ, bool LN, bool RESID, bool OUT_NCHW, int TAIL, bool XN = false, bool SP = false>
__device__ __forceinline__ void out_ffn_body(const FfnParams& p, const QkvParams* qp) {
    constexpr bool QKV = TAIL == 1, HEAD = TAIL == 2;
    using Cfg = ChainCfg<C, SP>;
    constexpr int KK = Cfg::KK, NT = Cfg::NT, NH = C / 32;   // hidden width == C
    constexpr bool STAGED = QKV && C == 256;   // coalesced Q / K' / V' stores through LDS (stage_tile / flush_tiles)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * Cfg::CHUNK_HALVES + 10 * C + (STAGED ? CHAIN_WAVES * STG_WAVE : 0)];
    half_t* ring0 = smem;
    half_t* ring1 = smem + Cfg::CHUNK_HALVES;
    float (*vec)[C] = reinterpret_cast<float (*)[C]>(smem + 2 * Cfg::CHUNK_HALVES);   // b_o, ln g, ln b, b_1, b_2

    const FfnJob& J = p.job[blockIdx.y];
    const int P = p.P;
    if (J.need) {   // none of this workgroup's tokens is read by a later stage (k_window_need): nothing to do
        const int t = blockIdx.x * CHAIN_TOKENS + (threadIdx.x & (CHAIN_TOKENS - 1));
        const int r = t / p.W, c = t - r * p.W;
        const int live = (t < P) ? J.need[(r >> 3) * (p.W >> 3) + (c >> 3)] : 0;
        if (!__syncthreads_or(live)) return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 31, hi = lane >> 5;
    const int tok = blockIdx.x * CHAIN_TOKENS + wave * 32 + m;
    const bool valid = tok < P;
    const int ty = J.type;

    for (int i = threadIdx.x; i < C; i += CHAIN_THREADS) {
        vec[0][i] = OUTPROJ ? p.b_o[ty * C + i] : 0.f;
        vec[1][i] = LN ? p.ln_g[ty * C + i] : 1.f;
        vec[2][i] = LN ? p.ln_b[ty * C + i] : 0.f;
        vec[3][i] = p.b_1[ty * C + i];
        vec[4][i] = p.b_2[ty * C + i];
    }
    const half_t* wo = OUTPROJ ? p.w_o + (size_t)ty * NT * Cfg::CHUNK_HALVES : nullptr;
    const half_t* wf = p.w_ffn + (size_t)ty * 2 * NH * Cfg::CHUNK_HALVES;
    constexpr int N_OUT = OUTPROJ ? NT : 0;
    constexpr int N_CHUNKS = N_OUT + 2 * NH;
    auto chunk_ptr = [&](int c) -> const half_t* {
        return (c < N_OUT) ? wo + (size_t)c * Cfg::CHUNK_HALVES : wf + (size_t)(c - N_OUT) * Cfg::CHUNK_HALVES;
    };
    stage_chunk<C, SP>(chunk_ptr(0), ring0);

    // residual stream (f32) in accumulator layout: xacc[t][4j+i] = channel 32t + 8j + 4hi + i
    float16v xacc[NT];
    Operands<KK, SP> act;
    const int tok_c = min(tok, P - 1);   // out-of-range lanes read a valid token and never store
    if constexpr (XN) {
        // tail of the first stage: the residual is the module's (C, P) input itself (k_ln_qkv wrote no token-major copy).
        // Raw buffer loads: the channel offset rides in the scalar offset, one VGPR of address for all 128 loads.
        const unsigned long long a = (unsigned long long)J.x;
        int4v rs;
        rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        rs.z = C * P * 4;
        rs.w = 0x00020000;
        const int voff = (4 * hi * P + tok_c) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                xacc[t][e] = llvm_raw_buffer_load_f32(rs, voff, (32 * t + 8 * (e >> 2) + (e & 3)) * P * 4, 0);
    } else {
        const float* xp = J.x + (size_t)tok_c * C + 4 * hi;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 f = *reinterpret_cast<const float4*>(xp + 32 * t + 8 * j);
                xacc[t][4 * j + 0] = f.x; xacc[t][4 * j + 1] = f.y; xacc[t][4 * j + 2] = f.z; xacc[t][4 * j + 3] = f.w;
            }
    }
    // split mode: range normalisation (HmvitStageScales), see tail16_body; all 1 / unused in f16 mode
    float c_1 = p.c_1[ty], s_g = p.s_g[ty], k_2 = p.k_2[ty], b1_pre = 1.f;
    const float c_o = p.c_o[ty];
    // accumulator-layout f32 rows -> operands (x itself as the operand of mlp_head; split mode: scaled per token)
    auto acc_to_operands = [&]() {
        float s_tok = 1.f;
        if constexpr (SP) {
            c_1 = 1.f; s_g = 1.f; k_2 = 1.f;
            if (p.dyn_head) {
                float r = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) r = fmaxf(r, fabsf(xacc[t][e]));
                float lo_, hi_;
                xor32_pair(r, lo_, hi_);
                r = fmaxf(lo_, hi_);
                s_tok = pow2_scale(r);
                b1_pre = p.head.w1[ty] * s_tok;
                c_1 = pow2_inv(b1_pre);
                s_g = pow2_scale(fmaf(p.head.l1[ty], r, p.head.b1max[ty]));
                k_2 = p.head.w2[ty] * s_g;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (SP) split_h(xacc[t][8 * sx + q] * s_tok, act.hi[2 * t + sx][q], act.lo[2 * t + sx][q]);
                    else act.hi[2 * t + sx][q] = (half_t)xacc[t][8 * sx + q];
                }
    };
    if constexpr (!OUTPROJ && !LN) acc_to_operands();   // mlp_head: x itself is the operand
    if constexpr (OUTPROJ) {
        // attention output of this token as B operands: img_o is built with the linear K order (weights.py
        // weight_image(linear_k)), so fragment kk of lane (m, hi) is the 8 channels 16 kk + 8 hi + (0..7)
        if constexpr (SP) {
            const float* op = reinterpret_cast<const float*>(J.o) + (size_t)tok_c * C + 8 * hi;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const float4 a = *reinterpret_cast<const float4*>(op + 16 * kk);
                const float4 b = *reinterpret_cast<const float4*>(op + 16 * kk + 4);
                const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) split_h(f[q], act.hi[kk][q], act.lo[kk][q]);
            }
        } else {
            const half_t* op = reinterpret_cast<const half_t*>(J.o) + (size_t)tok_c * C + 8 * hi;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) act.hi[kk] = *reinterpret_cast<const half8*>(op + 16 * kk);
        }
    }
    dma_wait();
    __syncthreads();

    // ---- phase 1: x' = x + b_o + W_o . O, one 32-channel tile per chunk (static accumulator index) ----
    if constexpr (OUTPROJ) {
        // rolled loop (keeps the scheduler's live ranges short); the tile result is added into the
        // statically indexed accumulator through a wave-uniform branch chain
#pragma unroll 1
        for (int c = 0; c < N_OUT; ++c) {
            const half_t* buf = (c & 1) ? ring1 : ring0;
            stage_chunk<C, SP>(chunk_ptr(c + 1), (c & 1) ? ring0 : ring1);   // there is always a next chunk
            float16v acc;   // starts from the out-projection bias of the tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bo = *reinterpret_cast<const float4*>(&vec[0][32 * c + 8 * j + 4 * hi]);
                acc[4 * j + 0] = bo.x; acc[4 * j + 1] = bo.y; acc[4 * j + 2] = bo.z; acc[4 * j + 3] = bo.w;
            }
            mma_chunk_op<KK, 4, SP>(acc, buf, act, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t == c) {
                    if constexpr (SP) xacc[t] = acc * c_o + xacc[t];   // b_o arrives pre-divided by c_o
                    else xacc[t] += acc;
                }
            dma_wait();
            wg_barrier();
        }
    }
    if constexpr (LN) ln_acc_to_operands<C, SP>(xacc, vec[1], vec[2], hi, act);

    // accumulator of the second Linear starts from (residual +) b_2
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b2 = *reinterpret_cast<const float4*>(&vec[4][32 * t + 8 * j + 4 * hi]);
            if constexpr (RESID && SP) {   // the row is carried as x k_2 through W_2's products; b_2 arrives as b_2 k_2
                xacc[t][4 * j + 0] = fmaf(xacc[t][4 * j + 0], k_2, b2.x); xacc[t][4 * j + 1] = fmaf(xacc[t][4 * j + 1], k_2, b2.y);
                xacc[t][4 * j + 2] = fmaf(xacc[t][4 * j + 2], k_2, b2.z); xacc[t][4 * j + 3] = fmaf(xacc[t][4 * j + 3], k_2, b2.w);
            } else if constexpr (RESID) {
                xacc[t][4 * j + 0] += b2.x; xacc[t][4 * j + 1] += b2.y;
                xacc[t][4 * j + 2] += b2.z; xacc[t][4 * j + 3] += b2.w;
            } else if constexpr (SP) {     // mlp_head: per-token k_2, b_2 at its true scale
                xacc[t][4 * j + 0] = b2.x * k_2; xacc[t][4 * j + 1] = b2.y * k_2;
                xacc[t][4 * j + 2] = b2.z * k_2; xacc[t][4 * j + 3] = b2.w * k_2;
            } else {
                xacc[t][4 * j + 0] = b2.x; xacc[t][4 * j + 1] = b2.y;
                xacc[t][4 * j + 2] = b2.z; xacc[t][4 * j + 3] = b2.w;
            }
            if (j == 3) __builtin_amdgcn_sched_barrier(0);
        }

    // ---- phase 2: per hidden tile hc: h = GELU(W_1[hc] . xn + b_1[hc]);  x'' += W_2[:, hc] . h ----
    // (N_OUT is even, so the W_1 chunk of every iteration sits in ring0 and the W_2 chunk in ring1)
    auto ffn_pass = [&](const half_t* wfp) {
#pragma unroll 1
    for (int hc = 0; hc < NH; ++hc) {
        const half_t* w1c = wfp + (size_t)(2 * hc) * Cfg::CHUNK_HALVES;
        float16v hacc;                 // f16 mode: starts from b_1; split mode: from 0, h = hacc c_1 + b_1 below
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b1 = *reinterpret_cast<const float4*>(&vec[3][32 * hc + 8 * j + 4 * hi]);
            // split mode: b_1 / c_1 (the stage's own chain: pre-divided on the host, b1_pre = 1; mlp_head: per token)
            hacc[4 * j + 0] = b1.x; hacc[4 * j + 1] = b1.y; hacc[4 * j + 2] = b1.z; hacc[4 * j + 3] = b1.w;
            if constexpr (SP) {
                hacc[4 * j + 0] *= b1_pre; hacc[4 * j + 1] *= b1_pre; hacc[4 * j + 2] *= b1_pre; hacc[4 * j + 3] *= b1_pre;
            }
        }
        stage_chunk<C, SP>(w1c + Cfg::CHUNK_HALVES, ring1);          // W_2 slice hc
        mma_chunk_op<KK, 4, SP>(hacc, ring0, act, lane);
        dma_wait();
        wg_barrier();

        // W_1 tile hc + 1 is requested before the GELU: ring0 is free since the barrier, and the activation's VALU work
        // covers the DMA's flight on top of the W_2 products
        if (hc + 1 < NH) stage_chunk<C, SP>(w1c + 2 * Cfg::CHUNK_HALVES, ring0);
        half8 hop[2], hopl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (SP) split_h(gelu_f(hacc[8 * s + q] * c_1) * s_g, hop[s][q], hopl[s][q]);
                else hop[s][q] = (half_t)gelu_f(hacc[8 * s + q]);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (SP) {
                    const half8 wh = lds_frag(ring1, (t * 2 + s) * 2, lane), wl = lds_frag(ring1, (t * 2 + s) * 2 + 1, lane);
                    xacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hopl[s], xacc[t], 0, 0, 0);
                    xacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hop[s], xacc[t], 0, 0, 0);
                    xacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hop[s], xacc[t], 0, 0, 0);
                } else {
                    xacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(ring1, t * 2 + s, lane), hop[s], xacc[t], 0, 0, 0);
                }
            }
        }
        dma_wait();
        wg_barrier();
    }
    };
    ffn_pass(wf);
    if constexpr (SP) {
        const float k_inv = pow2_inv(k_2);
#pragma unroll
        for (int t = 0; t < NT; ++t) xacc[t] *= k_inv;
    }

    auto store_x = [&]() {
        if (valid && !(QKV && J.pad)) {   // pad = 1: x'' is not needed in memory (fused launch before the pruned stage)
            if constexpr (OUT_NCHW) {
                float* op = J.out + (size_t)(4 * hi) * P + tok;
    #pragma unroll
                for (int t = 0; t < NT; ++t)
    #pragma unroll
                    for (int e = 0; e < 16; ++e) op[(size_t)(32 * t + 8 * (e >> 2) + (e & 3)) * P] = xacc[t][e];
            } else {
                float* op = J.out + (size_t)tok * C + 4 * hi;
    #pragma unroll
                for (int t = 0; t < NT; ++t)
    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4*>(op + 32 * t + 8 * j) =
                            make_float4(xacc[t][4 * j], xacc[t][4 * j + 1], xacc[t][4 * j + 2], xacc[t][4 * j + 3]);
            }
        }
    };
    // the same rows, two 32-channel tiles (256 bytes per token) at a time through the wave's staging buffer: every
    // store instruction covers 4 tokens x 256 contiguous bytes (see stage_tile / flush_tiles)
    auto store_x_staged = [&](half_t* stg_h) {
        if (J.pad) return;
        float* stg = reinterpret_cast<float*>(stg_h);            // 32 rows of 64 floats + 4 floats of padding
        constexpr int ROW = STG_ROW / 2;
        const int tok_w = blockIdx.x * CHAIN_TOKENS + wave * 32, n_tok = P - tok_w;
    #pragma unroll
        for (int t2 = 0; t2 < NT / 2; ++t2) {
    #pragma unroll
            for (int tt = 0; tt < 2; ++tt)
    #pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(stg + m * ROW + 32 * tt + 8 * j + 4 * hi) =
                        make_float4(xacc[2 * t2 + tt][4 * j], xacc[2 * t2 + tt][4 * j + 1], xacc[2 * t2 + tt][4 * j + 2],
                                    xacc[2 * t2 + tt][4 * j + 3]);
    #pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int token = 4 * k + (lane >> 4), piece = lane & 15;
                const float4 v = *reinterpret_cast<const float4*>(stg + token * ROW + piece * 4);
                if (token < n_tok) *reinterpret_cast<float4*>(J.out + (size_t)(tok_w + token) * C + 64 * t2 + piece * 4) = v;
            }
        }
    };
    if constexpr (TAIL == 0) store_x();

    if constexpr (HEAD) {
        // ---- mlp_head on the ego's x'' (bevformer_point_pillar_hetero.py:48): Linear -> GELU -> Linear, no norm, no
        // residual; x'' itself never goes to memory, the output is the NCHW map the model returns ----
        acc_to_operands();
        // vec[3..4] and both ring buffers are free (last use before the barrier that ended phase 2)
        for (int i = threadIdx.x; i < C; i += CHAIN_THREADS) {
            vec[3][i] = p.hb_1[ty * C + i];
            vec[4][i] = p.hb_2[ty * C + i];
        }
        const half_t* wh = p.w_head + (size_t)ty * 2 * NH * Cfg::CHUNK_HALVES;
        stage_chunk<C, SP>(wh, ring0);
        dma_wait();
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b2 = *reinterpret_cast<const float4*>(&vec[4][32 * t + 8 * j + 4 * hi]);
                xacc[t][4 * j + 0] = b2.x; xacc[t][4 * j + 1] = b2.y; xacc[t][4 * j + 2] = b2.z; xacc[t][4 * j + 3] = b2.w;
                if constexpr (SP) {
                    xacc[t][4 * j + 0] *= k_2; xacc[t][4 * j + 1] *= k_2; xacc[t][4 * j + 2] *= k_2; xacc[t][4 * j + 3] *= k_2;
                }
            }
        ffn_pass(wh);
        if constexpr (SP) {
            const float k_inv = pow2_inv(k_2);
#pragma unroll
            for (int t = 0; t < NT; ++t) xacc[t] *= k_inv;
        }
        if (valid) {
            float* op = J.out + (size_t)(4 * hi) * P + tok;      // (C, P) map
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) op[(size_t)(32 * t + 8 * (e >> 2) + (e & 3)) * P] = xacc[t][e];
        }
    }

    if constexpr (QKV) {
        // ---- next stage: LayerNorm(x'') -> Q / K' / V' tiles, exactly k_ln_qkv's loop ----
        const QkvJob& Q = qp->job[blockIdx.y];
        const int n_chunks = Q.n_mat * NT;
        if (n_chunks > 0) {
            // both ring buffers and vec[1..2] are free here (last use before the barrier that ended phase 2)
            for (int i = threadIdx.x; i < C; i += CHAIN_THREADS) {
                vec[1][i] = qp->gamma[ty * C + i];
                vec[2][i] = qp->beta[ty * C + i];
            }
            stage_chunk<C, SP>(Q.w[0], ring0);
            dma_wait();
            __syncthreads();
            ln_acc_to_operands<C, SP>(xacc, vec[1], vec[2], hi, act);
            // x'' leaves while the first tiles are computed (the stores are not waited for here)
            if constexpr (STAGED) store_x_staged(smem + 2 * Cfg::CHUNK_HALVES + 10 * C + wave * STG_WAVE);
            else store_x();
            half_t* stg = smem + 2 * Cfg::CHUNK_HALVES + 10 * C + wave * STG_WAVE;
            const int tok_w = blockIdx.x * CHAIN_TOKENS + wave * 32;
            for (int c = 0; c < n_chunks; ++c) {
                const int mat = c / NT, t = c - mat * NT;
                const half_t* buf = (c & 1) ? ring1 : ring0;
                if (c + 1 < n_chunks) {
                    const int c1 = c + 1, mat1 = c1 / NT, t1 = c1 - mat1 * NT;
                    stage_chunk<C, SP>(Q.w[mat1] + (size_t)t1 * Cfg::CHUNK_HALVES, (c & 1) ? ring0 : ring1);
                }
                float16v acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                mma_chunk_op<KK, 4, SP>(acc, buf, act, lane);
                if constexpr (SP) acc *= Q.c[mat];
#ifndef HMVIT_EXP_NOWAIT
                dma_wait();
#endif
#ifndef HMVIT_EXP_NOSTORE
                store_proj_tile<C, SP, STAGED>(stg, Q.y[mat], t, acc, m, hi, lane, tok, valid, tok_w, P);
#endif
                wg_barrier();
            }
        } else {
            store_x();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Split mode, C = 256: 16 tokens per wavefront, 8 wavefronts per workgroup ("x16" kernels).
//
// The 32-token split kernels above hold a token's residual row, both operand halves and the accumulators of one lane pair in
// ~470 registers: one wavefront per SIMD, so VALU work (LayerNorm, GELU, hi / lo splits), LDS staging, global loads / stores
// and the MFMA stream of a workgroup run one after the other.  Here a token is spread over FOUR lanes
// (v_mfma_f32_16x16x32_f16: lane (tk = lane & 15, g = lane >> 4) supplies B[k = 8 g + j][token tk] and receives
// D[row 4 g + r][token tk]), 128 + ~60 registers per lane, two wavefronts per SIMD in one 512-thread workgroup: same 128 tokens,
// same weight chunks (32 output rows x K = 256 x (hi, lo) = 32 KB) and ring as before, but while one wavefront of a SIMD
// converts, stages or waits, the other one feeds the matrix pipe.
//
// Layout: lane (tk, g) owns channels 16 t + 4 g + r of token tk (t < 16, r < 4): xacc[t] (float4).  k-step s (channels
// 32 s .. 32 s + 31) takes operand slot j of that lane from channel 32 s + 16 (j >> 2) + 4 g + (j & 3), i.e. from
// xacc[2 s + (j >> 2)][j & 3]; the weight images are built in that k order (weights.py weight_image16): fragment
// (row tile T, k-step s, half) = 64 lanes x 8 halves with lane (l, g) holding W[16 T + l][32 s + 16 (j >> 2) + 4 g + (j & 3)].
// A chunk = row tiles (2 c, 2 c + 1) x 8 k-steps x (hi, lo); a W_2 slice chunk = 16 row tiles x k-step hc x (hi, lo).
// ------------------------------------------------------------------------------------------
constexpr int X16_WAVES = 8, X16_THREADS = 512, X16_TOKENS = 128;
constexpr int X16_CHUNK = 16384;                    // halves per chunk (32 KB)
constexpr int X16_STG_ROW = 64 + 4;                 // floats per staged token row (64 channels + padding)
constexpr int X16_STG_WAVE = 16 * X16_STG_ROW;      // floats per wave

__device__ __forceinline__ void stage_chunk16(const half_t* __restrict__ chunk, half_t* lds_buf) {
#ifdef HMVIT_EXP_X16_NODMA
    return;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_buf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                   // 2048 pieces of 16 bytes, 512 threads
        const int piece0 = (i * X16_WAVES + wave) * 64;
        const uint4* gsrc = reinterpret_cast<const uint4*>(chunk) + piece0 + lane;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + piece0 * 16);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    }
}

__device__ __forceinline__ float quad_sum(float v) { return xor32_sum(xor16_sum(v)); }   // over the 4 lanes of a token

// (Measured and dropped: issuing the products in pairs over two accumulators so that no MFMA waits for its predecessor's result -
// the lone wave's 48 products still took ~1120 cycles (projection chunk) / ~1640 (W_2 slice), i.e. the dependent accumulator is
// not what holds them above the 48 x 17 cycles of bare back-to-back MFMAs; neither is the fragment prefetch depth (8 / 12 / 16
// measured equal).  What remains is the issue of the 32 ds_read_b128 of a chunk between the products.)
// HMVIT_X16_REFILL_LAG: after the products of fragment f the ring slot of fragment f - LAG is refilled (0: the slot just
// consumed).  Measured 0 / 1 / 2: tails 5.25 / 5.21 / 5.16 ms - the refill does not wait on the MFMA that read the slot.
#ifndef HMVIT_X16_DEFER
#define HMVIT_X16_DEFER 0               // group B's barrier behind a step that is followed by non-ring work moves behind that work
#endif
#ifndef HMVIT_X16_REFILL_LAG
#define HMVIT_X16_REFILL_LAG 0
#endif
// fragment f of a chunk: hi fragments feed two MFMAs (x a_lo, x a_hi), lo fragments one (x a_hi)
// NT16 = 2: projection chunk, fragment f = (row tile f / 16, k-step (f % 16) / 2, half f & 1), accumulators acc[2]
template <int DEPTH, bool HAS_LO, int f = 0>
struct MmaProj16 {
    static __device__ __forceinline__ void run(float4v (&acc)[2], unsigned addr, const half8 (&ah)[8], const half8 (&al)[8], half8 (&w)[DEPTH]) {
        constexpr int F = 32, T = f / 16, s = (f % 16) / 2;
        constexpr int LAG = HMVIT_X16_REFILL_LAG;             // the slot refilled after fragment f is the one fragment f - LAG used
        constexpr int issued = (DEPTH + (f > LAG ? f - LAG : 0)) < F ? (DEPTH + (f > LAG ? f - LAG : 0)) : F;
        constexpr int outstanding = issued - (f + 1);
        lgkm_wait<outstanding>();
        if constexpr ((f & 1) == 0) {
            if constexpr (HAS_LO) acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], al[s], acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], ah[s], acc[T], 0, 0, 0);
        } else {
            acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], ah[s], acc[T], 0, 0, 0);
        }
        if constexpr (f >= LAG && f - LAG + DEPTH < F) {
            __builtin_amdgcn_sched_barrier(0);
            lds_read_frag<(f - LAG + DEPTH) * 1024>(w[(f - LAG) % DEPTH], addr);
        }
        if constexpr (f + 1 < F) MmaProj16<DEPTH, HAS_LO, f + 1>::run(acc, addr, ah, al, w);
    }
};
#ifndef HMVIT_X16_DEPTH
#define HMVIT_X16_DEPTH 8
#endif
template <bool HAS_LO = true>
__device__ __forceinline__ void mma_proj16(float4v (&acc)[2], const half_t* buf, const half8 (&ah)[8], const half8 (&al)[8], int lane) {
#ifdef HMVIT_EXP_NOMMA
    return;
#endif
    constexpr int DEPTH = HMVIT_X16_DEPTH;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(buf + lane * 8);
    half8 w[DEPTH];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MmaPrologue<DEPTH>::run(addr, w);
    MmaProj16<DEPTH, HAS_LO>::run(acc, addr, ah, al, w);
    __builtin_amdgcn_sched_barrier(0);
}
// W_2 slice chunk: fragment f = (row tile f / 2, half f & 1), one k-step (the hidden tile), accumulators xacc[16]
template <int DEPTH, int f = 0>
struct MmaSlice16 {
    static __device__ __forceinline__ void run(float4v (&xacc)[16], unsigned addr, const half8& hh, const half8& hl, half8 (&w)[DEPTH]) {
        constexpr int F = 32, T = f / 2;
        constexpr int LAG = HMVIT_X16_REFILL_LAG;
        constexpr int issued = (DEPTH + (f > LAG ? f - LAG : 0)) < F ? (DEPTH + (f > LAG ? f - LAG : 0)) : F;
        constexpr int outstanding = issued - (f + 1);
        lgkm_wait<outstanding>();
        if constexpr ((f & 1) == 0) {
            xacc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], hl, xacc[T], 0, 0, 0);
            xacc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], hh, xacc[T], 0, 0, 0);
        } else {
            xacc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[f % DEPTH], hh, xacc[T], 0, 0, 0);
        }
        if constexpr (f >= LAG && f - LAG + DEPTH < F) {
            __builtin_amdgcn_sched_barrier(0);
            lds_read_frag<(f - LAG + DEPTH) * 1024>(w[(f - LAG) % DEPTH], addr);
        }
        if constexpr (f + 1 < F) MmaSlice16<DEPTH, f + 1>::run(xacc, addr, hh, hl, w);
    }
};
__device__ __forceinline__ void mma_slice16(float4v (&xacc)[16], const half_t* buf, const half8& hh, const half8& hl, int lane) {
#ifdef HMVIT_EXP_NOMMA
    return;
#endif
    constexpr int DEPTH = HMVIT_X16_DEPTH;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(buf + lane * 8);
    half8 w[DEPTH];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    MmaPrologue<DEPTH>::run(addr, w);
    MmaSlice16<DEPTH>::run(xacc, addr, hh, hl, w);
    __builtin_amdgcn_sched_barrier(0);
}

// f32 rows in accumulator layout -> operand halves: slot j of k-step s <- x[2 s + (j >> 2)][j & 3]
__device__ __forceinline__ void rows_to_operands16(const float4v (&x)[16], half8 (&ah)[8], half8 (&al)[8], float sc) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[2 * s + (j >> 2)][j & 3] * sc;
        split_pk8(v, ah[s], al[s]);
    }
}
// max |x| over the token's 256 channels (the 4 lanes of the token each hold 64)
__device__ __forceinline__ float row_absmax16(const float4v (&x)[16]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(x[t][r]));
    return max_over_lane_groups(m);
}
// LayerNorm of the token held by 4 lanes, straight into operand halves (gamma / beta from LDS).  Written on 2-vectors: every
// step is a packed f32 instruction (v_pk_add / v_pk_fma_f32, two channels each) - the wave runs this while its SIMD partner
// owns the matrix pipe, and a lone wave is bound by instruction issue, not by the VALU's width
__device__ __forceinline__ void ln_to_operands16(const float4v (&x)[16], const float* __restrict__ lg, const float* __restrict__ lb, int g,
                                                 half8 (&ah)[8], half8 (&al)[8]) {
    constexpr int C = 256;
    float2v s2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) s2 += x[t].xy + x[t].zw;
    const float mean = quad_sum(s2.x + s2.y) * (1.f / C);
    const float2v m2 = {mean, mean};
    float2v q2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float2v d0 = x[t].xy - m2, d1 = x[t].zw - m2;
        q2 = __builtin_elementwise_fma(d0, d0, q2);
        q2 = __builtin_elementwise_fma(d1, d1, q2);
    }
    const float rstd = rsqrtf(quad_sum(q2.x + q2.y) * (1.f / C) + 1e-5f);
    const float2v r2 = {rstd, rstd}, sh2 = {-mean * rstd, -mean * rstd};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int t = 2 * s + jj, c = 16 * t + 4 * g;
            const float4v ga = *reinterpret_cast<const float4v*>(lg + c);
            const float4v be = *reinterpret_cast<const float4v*>(lb + c);
            const float2v n0 = __builtin_elementwise_fma(__builtin_elementwise_fma(x[t].xy, r2, sh2), ga.xy, be.xy);
            const float2v n1 = __builtin_elementwise_fma(__builtin_elementwise_fma(x[t].zw, r2, sh2), ga.zw, be.zw);
            v[4 * jj + 0] = n0.x; v[4 * jj + 1] = n0.y; v[4 * jj + 2] = n1.x; v[4 * jj + 3] = n1.y;
        }
        split_pk8(v, ah[s], al[s]);
    }
}
// LayerNorm of the token held by 4 lanes, in place (f32 rows; k_linear16 scales and splits them afterwards)
__device__ __forceinline__ void ln_rows16(float4v (&x)[16], const float* __restrict__ lg, const float* __restrict__ lb, int g) {
    constexpr int C = 256;
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) sm += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    const float mean = quad_sum(sm) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x[t][r] - mean;
            q = fmaf(d, d, q);
        }
    const float rstd = rsqrtf(quad_sum(q) * (1.f / C) + 1e-5f);
    const float shift = -mean * rstd;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float4 ga = *reinterpret_cast<const float4*>(lg + 16 * t + 4 * g);
        const float4 be = *reinterpret_cast<const float4*>(lb + 16 * t + 4 * g);
        x[t][0] = fmaf(x[t][0], rstd, shift) * ga.x + be.x;
        x[t][1] = fmaf(x[t][1], rstd, shift) * ga.y + be.y;
        x[t][2] = fmaf(x[t][2], rstd, shift) * ga.z + be.z;
        x[t][3] = fmaf(x[t][3], rstd, shift) * ga.w + be.w;
    }
}

// two projected row tiles (32 channels 32 c .. 32 c + 31) of the wave's 16 tokens -> f32 plane, through the wave's staging rows:
// chunks (2 k, 2 k + 1) fill 64 channels = 256 bytes per token, then 4 store instructions write 4 x (4 tokens x 256 bytes)
__device__ __forceinline__ void store_proj16(float* stg, float* y, int c, const float4v (&acc)[2], float cm, int tk, int g, int lane, int tok_w, int P) {
    constexpr int C = 256;
    float* d = stg + tk * X16_STG_ROW + 32 * (c & 1) + 4 * g;
    *reinterpret_cast<float4*>(d) = make_float4(acc[0][0] * cm, acc[0][1] * cm, acc[0][2] * cm, acc[0][3] * cm);
    *reinterpret_cast<float4*>(d + 16) = make_float4(acc[1][0] * cm, acc[1][1] * cm, acc[1][2] * cm, acc[1][3] * cm);
    if (c & 1) {
        float* yo = y + (size_t)tok_w * C + 64 * (c >> 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int token = 4 * k + (lane >> 4), piece = lane & 15;
            const float4 v = *reinterpret_cast<const float4*>(stg + token * X16_STG_ROW + piece * 4);
#ifndef HMVIT_EXP_X16_NOSTORE
            if (tok_w + token < P) *reinterpret_cast<float4*>(yo + (size_t)token * C + piece * 4) = v;
#else
            if (v.x == 1.2345e-30f) *reinterpret_cast<float4*>(yo + (size_t)token * C + piece * 4) = v;
#endif
        }
    }
}

// Weight ring of the x16 kernels: THREE chunk slots.  Chunk i is requested two steps ahead (into the slot chunk i - 3 left at
// the barrier that ended its step) and must have landed when step i - 1 ends: the LDS-DMA of a 32 KB chunk takes longer from
// issue to landing (~1.1-1.3 us) than a step's matrix products (~0.65 us), so with a single chunk of lookahead every step
// stalled on its successor.  ring_wait(n): all of this wave's memory operations except the newest n requests are complete.
constexpr int X16_RING = 3;
#ifndef HMVIT_X16_DEPHASE
#define HMVIT_X16_DEPHASE 1
#endif
constexpr bool X16_DEPHASE = HMVIT_X16_DEPHASE != 0;
#ifdef HMVIT_PROBE
// cycle stamps of one workgroup of the tail kernel, lane 0 of wave 0 (group A, first 64 rows) and of wave 4 (group B, next 64):
// [step][0 begin, 1 products done, 2 chunk confirmed (B), 3 mid barrier passed, 4 rest of the step + request done, 5 end barrier
// passed]; read back with hmvit_debug_x16_trace (tools/probe/x16_trace.py).
__device__ unsigned long long g_x16_trace[2 * 64 * 8];
#ifndef HMVIT_X16_STAMP_TAIL
#define HMVIT_X16_STAMP_TAIL 1          // which kernel stamps: 1 = k_out_ffn_qkv16, 2 = k_out_ffn_head16
#endif
#define X16_STAMP(step, slot)                                                                             \
    do {                                                                                                  \
        if (TAIL == HMVIT_X16_STAMP_TAIL && blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && (step) < 64) \
            g_x16_trace[((threadIdx.x >> 8) * 64 + (step)) * 8 + (slot)] = __builtin_readcyclecounter();  \
    } while (0)
#else
#define X16_STAMP(step, slot) do {} while (0)
#endif
__device__ __forceinline__ void ring_wait_newest4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
// the same when this wave has issued the 4 global stores of a tile flush (store_proj16) between the request it waits for and
// the newest request: loads and stores share the in-order vmcnt counter, so "newest 8" keeps those stores in flight too - a
// store's write acknowledgement takes longer than a step, and waiting for it at every flush stalled the whole workgroup
__device__ __forceinline__ void ring_wait_newest8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

// "mixed" precision mode (A16): the projected Q / K' / V' planes are f16 (they are attention operands, rounded once; the f16
// attention kernels consume them), 128 channels = 256 bytes per token and flush
__device__ __forceinline__ void store_proj16_h(float* stg_f, half_t* y, int c, const float4v (&acc)[2], float cm, int tk, int g, int lane, int tok_w, int P) {
    constexpr int C = 256, ROW = 2 * X16_STG_ROW;          // halves per staged token row (128 channels + padding)
    half_t* stg = reinterpret_cast<half_t*>(stg_f);
    half_t* d = stg + tk * ROW + 32 * (c & 3) + 4 * g;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[T][r] * cm);
        *reinterpret_cast<half4*>(d + 16 * T) = h;
    }
    if ((c & 3) == 3) {
        half_t* yo = y + (size_t)tok_w * C + 128 * (c >> 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int token = 4 * k + (lane >> 4), piece = lane & 15;
            const half8 v = *reinterpret_cast<const half8*>(stg + token * ROW + piece * 8);
            if (tok_w + token < P) *reinterpret_cast<half8*>(yo + (size_t)token * C + piece * 8) = v;
        }
    }
}

template <bool A16>
__global__ __launch_bounds__(X16_THREADS, 2) void k_ln_qkv16(QkvParams p) {
    constexpr int C = 256, NCH = 8;                 // chunks (32 rows) per matrix
    __shared__ __attribute__((aligned(16))) half_t smem[X16_RING * X16_CHUNK + 4 * C + 2 * X16_WAVES * X16_STG_WAVE];
    float* lnp = reinterpret_cast<float*>(smem + X16_RING * X16_CHUNK);
    const QkvJob& J = p.job[blockIdx.y];
    const int P = p.P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tk = lane & 15, g = lane >> 4;
    const int tok_w = blockIdx.x * X16_TOKENS + wave * 16, tok = tok_w + tk;
    float* stg = reinterpret_cast<float*>(smem + X16_RING * X16_CHUNK + 4 * C) + wave * X16_STG_WAVE;
    for (int i = threadIdx.x; i < C; i += X16_THREADS) {
        lnp[i] = p.gamma[J.type * C + i];
        lnp[C + i] = p.beta[J.type * C + i];
    }
    const int n_chunks = J.n_mat * NCH;
    auto chunk_ptr = [&](int i) -> const half_t* { return J.w[i / NCH] + (size_t)(i % NCH) * X16_CHUNK; };
    auto slot = [&](int i) -> half_t* { return smem + (i % X16_RING) * X16_CHUNK; };
    // step protocol and the one-barrier skew of waves 4-7: see tail16_body
    const bool grp_b = X16_DEPHASE && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= 4;
    const bool full_wave = tok_w + 16 <= P;
    stage_chunk16(chunk_ptr(0), slot(0));
    if (n_chunks > 1) stage_chunk16(chunk_ptr(1), slot(1));
    if (grp_b && n_chunks > 2) stage_chunk16(chunk_ptr(2), slot(2));
    float4v x[16];
    const int tok_c = min(tok, P - 1);
    if (p.in_nchw) {
        const float* xp = J.x + (size_t)(4 * g) * P + tok_c;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] = xp[(size_t)(16 * t + r) * P];
        if (tok < P && J.xs_out) {
            float* xo = J.xs_out + (size_t)tok * C + 4 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<float4*>(xo + 16 * t) = make_float4(x[t][0], x[t][1], x[t][2], x[t][3]);
        }
    } else {
        const float* xp = J.x + (size_t)tok_c * C + 4 * g;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 f = *reinterpret_cast<const float4*>(xp + 16 * t);
            x[t][0] = f.x; x[t][1] = f.y; x[t][2] = f.z; x[t][3] = f.w;
        }
    }
    dma_wait();
    __syncthreads();
    if (grp_b) wg_barrier();
    half8 ah[8], al[8];
    ln_to_operands16(x, lnp, lnp + C, g, ah, al);
    auto flushes = [](int k) { return k >= 0 && (A16 ? (k & 3) == 3 : (k & 1) != 0); };
    const float cq0 = J.c[0], cq1 = J.c[1], cq2 = J.c[2], cq3 = J.c[3], cq4 = J.c[4];
    for (int c = 0; c < n_chunks; ++c) {
        const int mat = c / NCH, t = c - mat * NCH;
        const float cm = mat == 0 ? cq0 : mat == 1 ? cq1 : mat == 2 ? cq2 : mat == 3 ? cq3 : cq4;
        auto ring_wait = [&](bool flush_since) {
            if (c + 2 < n_chunks && full_wave) { if (flush_since) ring_wait_newest8(); else ring_wait_newest4(); }
            else dma_wait();
        };
        float4v acc[2] = {(float4v)(0.f), (float4v)(0.f)};
        mma_proj16(acc, slot(c), ah, al, lane);
        if constexpr (X16_DEPHASE) {
            if (grp_b) ring_wait(flushes(c - 1));
            wg_barrier();
        }
        if constexpr (A16) store_proj16_h(stg, reinterpret_cast<half_t*>(J.y[mat]), t, acc, cm, tk, g, lane, tok_w, P);
        else store_proj16(stg, reinterpret_cast<float*>(J.y[mat]), t, acc, cm, tk, g, lane, tok_w, P);
        const int ahead = c + 2 + (grp_b ? 1 : 0);
        if (ahead < n_chunks) stage_chunk16(chunk_ptr(ahead), slot(ahead));
        if (!grp_b) ring_wait(flushes(c));
        wg_barrier();
    }
    if (X16_DEPHASE && !grp_b) wg_barrier();
}

// ------------------------------------------------------------------------------------------
// k_linear16: y_m = a W_m^T (+ bias_m) (+ residual), a (M, 256) f32, up to three (256, 256) matrices per job - the skinny GEMMs of
// the training path (capi_train.hip).  k_ln_qkv16 without the LayerNorm: 128 rows per workgroup, every row read once, all 256
// n_mat output columns formed from the same operand registers, weight chunks through the three-slot LDS-DMA ring.  The generic tile
// kernel it replaces (gemm.hip k_gemm_split, 128 x 128 tiles) fetched every row of `a` twice and the weights once per tile: 1.5 GB
// fetched for 0.7 GB of rows per launch, 0.73 ms where the bytes take 0.3 ms (profiles/r03_train_pmc.txt).
// Range: the row is scaled per token to [2^13, 2^14) before the hi / lo split (pow2_scale), the image carries the matrix at a
// power of two (launch_weight_images16): exact, and gradients of any magnitude keep their low halves.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_lin16(float* stg, float* y, int ldy, int c, const float4v (&v)[2], int tk, int g, int lane, int tok_w, int M) {
    float* d = stg + tk * X16_STG_ROW + 32 * (c & 1) + 4 * g;
    *reinterpret_cast<float4*>(d) = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
    *reinterpret_cast<float4*>(d + 16) = make_float4(v[1][0], v[1][1], v[1][2], v[1][3]);
    if (c & 1) {
        float* yo = y + (size_t)tok_w * ldy + 64 * (c >> 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int token = 4 * k + (lane >> 4), piece = lane & 15;
            const float4 q = *reinterpret_cast<const float4*>(stg + token * X16_STG_ROW + piece * 4);
            if (tok_w + token < M) *reinterpret_cast<float4*>(yo + (size_t)token * ldy + piece * 4) = q;
        }
    }
}

__global__ __launch_bounds__(X16_THREADS, 2) void k_linear16(LinJobs jobs) {
    constexpr int C = 256, NCH = 8;                 // chunks (32 output rows) per matrix
    __shared__ __attribute__((aligned(16))) half_t smem[X16_RING * X16_CHUNK + 2 * (kMaxLinMats + 2) * C + 2 * X16_WAVES * X16_STG_WAVE];
    float* bs = reinterpret_cast<float*>(smem + X16_RING * X16_CHUNK);
    float* lnp = bs + kMaxLinMats * C;              // gamma, beta
    const LinJob& J = jobs.j[blockIdx.y];
    const int M = J.M;
    if ((int)blockIdx.x * X16_TOKENS >= M) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tk = lane & 15, g = lane >> 4;
    const int tok_w = blockIdx.x * X16_TOKENS + wave * 16, tok = tok_w + tk;
    float* stg = reinterpret_cast<float*>(smem + X16_RING * X16_CHUNK + 2 * (kMaxLinMats + 2) * C) + wave * X16_STG_WAVE;
    for (int i = threadIdx.x; i < J.n_mat * C; i += X16_THREADS) bs[i] = J.bias[i / C] ? J.bias[i / C][i % C] : 0.f;
    if (J.ln_gamma)
        for (int i = threadIdx.x; i < C; i += X16_THREADS) {
            lnp[i] = J.ln_gamma[i];
            lnp[C + i] = J.ln_beta[i];
        }
    const int n_chunks = J.n_mat * NCH;
    auto chunk_ptr = [&](int i) -> const half_t* { return J.wimg[i / NCH] + (size_t)(i % NCH) * X16_CHUNK; };
    auto slot = [&](int i) -> half_t* { return smem + (i % X16_RING) * X16_CHUNK; };
    // step protocol and the one-barrier skew of waves 4-7: see tail16_body
    const bool grp_b = X16_DEPHASE && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= 4;
    const bool full_wave = tok_w + 16 <= M;
    stage_chunk16(chunk_ptr(0), slot(0));
    stage_chunk16(chunk_ptr(1), slot(1));
    if (grp_b) stage_chunk16(chunk_ptr(2), slot(2));
    float4v x[16];
    const int tok_c = min(tok, M - 1);
    {
        const float* xp = J.a + (size_t)tok_c * C + 4 * g;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 f = *reinterpret_cast<const float4*>(xp + 16 * t);
            x[t][0] = f.x; x[t][1] = f.y; x[t][2] = f.z; x[t][3] = f.w;
        }
    }
    dma_wait();
    __syncthreads();
    if (grp_b) wg_barrier();
    half8 ah[8], al[8];
    if (J.ln_gamma) ln_rows16(x, lnp, lnp + C, g);
    const float s_tok = pow2_scale(row_absmax16(x));
    rows_to_operands16(x, ah, al, s_tok);
    const float inv_tok = pow2_inv(s_tok);
    // the residual rows take the registers of `a`, in the accumulator layout (channel 16 t + 4 g + r of the lane's token)
    if (J.residual) {
        const float* rp = J.residual + (size_t)tok_c * J.ldy + 4 * g;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 f = *reinterpret_cast<const float4*>(rp + 16 * t);
            x[t][0] = f.x; x[t][1] = f.y; x[t][2] = f.z; x[t][3] = f.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = (float4v)(0.f);
    }
    const bool drop_on = J.drop.p > 0.f;
    const float inv_keep = drop_on ? 1.f / (1.f - J.drop.p) : 1.f;
    const unsigned dkey = drop_key(J.drop);
    // sum_inputs: y[0] = sum over the job's inputs; x[] (the residual rows, or zero) is the accumulator and leaves once, at the end
    const bool sum = J.sum_inputs != 0;
    auto flushes = [&](int k) { return !sum && k >= 0 && (k & 1) != 0; };
    float inv_in = inv_tok;
    for (int mat = 0; mat < J.n_mat; ++mat) {
        if (sum && mat > 0) {
            float4v r[16];
            const float* xp = J.a_more[mat - 1] + (size_t)tok_c * C + 4 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 f = *reinterpret_cast<const float4*>(xp + 16 * t);
                r[t][0] = f.x; r[t][1] = f.y; r[t][2] = f.z; r[t][3] = f.w;
            }
            const float s_in = pow2_scale(row_absmax16(r));
            rows_to_operands16(r, ah, al, s_in);
            inv_in = pow2_inv(s_in);
        }
        const float cm = inv_in * J.w_inv[mat][0];
        const float* bm = bs + (sum ? 0 : mat) * C + 4 * g;
        const bool res = mat == 0;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const int c = mat * NCH + t;
            auto ring_wait = [&](bool flush_since) {
                if (c + 2 < n_chunks && full_wave) { if (flush_since) ring_wait_newest8(); else ring_wait_newest4(); }
                else dma_wait();
            };
            float4v acc[2] = {(float4v)(0.f), (float4v)(0.f)};
            mma_proj16(acc, slot(c), ah, al, lane);
            if constexpr (X16_DEPHASE) {
                if (grp_b) ring_wait(flushes(c - 1));
                wg_barrier();
            }
            if (sum) {
                x[2 * t] += acc[0] * cm;
                x[2 * t + 1] += acc[1] * cm;
            } else {
            float4v v[2];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const float4 b4 = *reinterpret_cast<const float4*>(bm + 32 * t + 16 * T);
                const float4v r4 = res ? x[2 * t + T] : (float4v)(0.f);
                float4v lin = {fmaf(acc[T][0], cm, b4.x), fmaf(acc[T][1], cm, b4.y), fmaf(acc[T][2], cm, b4.z), fmaf(acc[T][3], cm, b4.w)};
                if (drop_on) {      // (uniform) the training forward's y = residual + Dropout(a W^T + b): the lane's four columns of its token
                    const unsigned long long i0 = (unsigned long long)tok * J.ldy + 32 * t + 16 * T + 4 * g;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lin[e] *= drop_scale(J.drop, dkey, i0 + e, inv_keep);
                }
                v[T] = lin + r4;
            }
            store_lin16(stg, J.y[mat], J.ldy, t, v, tk, g, lane, tok_w, M);
            }
            const int ahead = c + 2 + (grp_b ? 1 : 0);
            if (ahead < n_chunks) stage_chunk16(chunk_ptr(ahead), slot(ahead));
            if (!grp_b) ring_wait(flushes(c));
            wg_barrier();
        }
    }
    if (X16_DEPHASE && !grp_b) wg_barrier();
    if (sum) {
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            float4v v[2];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const float4 b4 = *reinterpret_cast<const float4*>(bs + 4 * g + 32 * t + 16 * T);
                v[T] = x[2 * t + T] + float4v{b4.x, b4.y, b4.z, b4.w};
            }
            store_lin16(stg, J.y[0], J.ldy, t, v, tk, g, lane, tok_w, M);
        }
    }
}

// (256, 256) f32 matrices -> x16 split images at a power of two: one workgroup per matrix (absmax, then the fragments;
// weights.py weight_image16 is the host-side statement of the layout)
__global__ __launch_bounds__(1024) void k_weight_image16(const float* __restrict__ w, half_t* __restrict__ img, float* __restrict__ w_inv) {
    constexpr int C = 256;
    __shared__ float red[16];
    const float* W = w + (size_t)blockIdx.x * C * C;
    half_t* I = img + (size_t)blockIdx.x * 2 * C * C;
    float m = 0.f;
    for (int i = threadIdx.x; i < C * C / 4; i += 1024) {
        const float4 f = reinterpret_cast<const float4*>(W)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(f.x), fabsf(f.y))), fmaxf(fabsf(f.z), fabsf(f.w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    const float sc = pow2_scale(m);
    if (threadIdx.x == 0) w_inv[blockIdx.x] = pow2_inv(sc);
    // fragment (T, s): lane (l, g) holds W[16 T + l][32 s + 16 (j >> 2) + 4 g + (j & 3)], hi fragment then lo fragment
    for (int f = threadIdx.x; f < 16 * 8 * 64; f += 1024) {
        const int lane = f & 63, s = (f >> 6) & 7, T = f >> 9;
        const float* row = W + (size_t)(16 * T + (lane & 15)) * C + 32 * s + 4 * (lane >> 4);
        const float4 f0 = *reinterpret_cast<const float4*>(row), f1 = *reinterpret_cast<const float4*>(row + 16);
        const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) split_h(v[j] * sc, h[j], l[j]);
        half_t* dst = I + ((size_t)(T * 8 + s) * 2) * 512 + lane * 8;
        *reinterpret_cast<half8*>(dst) = h;
        *reinterpret_cast<half8*>(dst + 512) = l;
    }
}

// TAIL: 0 = stage tail only (x'' stored), 1 = + next stage's LayerNorm / Q / K' / V', 2 = + mlp_head (output (C, P) map).
// OUTPROJ / LN / RESID as in out_ffn_body; OUT_NCHW (TAIL 0): x'' goes to a (C, P) map (the stand-alone mlp_head launch).
//
// PERSISTENT (round 4): a workgroup walks the 128-token tiles blockIdx.x + k gridDim.x of its job (k < 32; the launch sizes the
// grid to about one workgroup per CU) with ONE uninterrupted chunk sequence: the weight ring keeps turning across tile
// boundaries (the last steps of a tile request the first chunks of the next), group B stays one barrier behind group A from the
// first tile to the last, and the attention rows of the next tile are requested into the (by then idle) residual registers
// during the Q / K' / V' phase.  With one workgroup per CU and one tile per workgroup, 20-33 k of a tile's ~150 k cycles passed
// before the first MFMA (inputs + the first two chunks arriving with nothing to overlap them); now only the first tile pays.
// Tiles none of whose tokens a later stage reads (FfnJob::need) are dropped from the walk up front.
constexpr int X16_MAX_TILES = 32;      // tiles per workgroup (bits of the live mask)

// PULLED TILES (round 6, FfnParams::pull): one workgroup per CU, every workgroup draws (job, tile) tickets from ONE counter of the
// launch (ticket t = job t / n_tiles, tile t % n_tiles) - the walk's uninterrupted chunk ring without the static partition that lost
// more to the spread of the CUs' speeds than the walk saved (x16_grid_x).  The draw is a scalar atomic and the reachability lookups
// are scalar loads (lgkmcnt: the counted vmcnt waits of the ring never see them); wavefront 0 draws the ticket of tile n + 1 in the
// rest-half of the first step of tile n, where it would otherwise wait ~1 k cycles at the barrier for group B's products, and leaves
// it in an LDS word every wavefront reads after the out-projection.  Tickets of dead tiles (FfnJob::need) are skipped by the drawer.
__device__ __forceinline__ int tail16_pull(const FfnParams& p, int n_tiles) {
    const int total = p.n_jobs * n_tiles;
    for (;;) {
        int t = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(p.pull) : "memory");
        if (t >= total) return -1;
        const int job = t / n_tiles, tile = t - job * n_tiles;
        const unsigned char* need = p.job[job].need;
        if (!need) return t;
        unsigned long long last = ~0ull;
        unsigned word = 0;
        for (int i = 0; i < 16; ++i) {                   // the tile's 16 runs of 8 tokens: a run lies in one window (W % 8 == 0)
            const int tok0 = tile * X16_TOKENS + i * 8;
            if (tok0 >= p.P) break;
            const int r = tok0 / p.W, c = tok0 - r * p.W;
            const unsigned long long a = (unsigned long long)(need + (r >> 3) * (p.W >> 3) + (c >> 3)), a4 = a & ~3ull;
            if (a4 != last) {
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(word) : "s"(a4) : "memory");
                last = a4;
            }
            if ((word >> (8 * (unsigned)(a & 3))) & 0xffu) return t;
        }
    }
}

template <bool OUTPROJ, bool LN, bool RESID, bool OUT_NCHW, int TAIL, bool XN, bool A16>
__device__ __forceinline__ void tail16_body(const FfnParams& p, const QkvParams* qp) {
    constexpr int C = 256, NCH = 8, NH = 8;
    constexpr bool QKV = TAIL == 1 || TAIL == 3, HEAD = TAIL == 2;
    constexpr bool FFN = TAIL != 3;                 // TAIL 3 (round 6): LayerNorm + Q / K' / V' only - k_ln_qkv16 as pulled tiles
    static_assert(FFN || (!OUTPROJ && !LN && !RESID && !OUT_NCHW), "TAIL 3 is the projection phase alone");
    __shared__ __attribute__((aligned(16))) half_t smem[X16_RING * X16_CHUNK + 14 * C + 2 * X16_WAVES * X16_STG_WAVE + 8];
    // layout: [7 vector rows][per-wave store staging][weight ring]: the rows and the staging sit below 64 KB, where the 16-bit
    // offset field of the DS instructions reaches them from one base register (behind the ring every row position needed an
    // address register of its own: 32 of them, spilled, for the LayerNorm parameters alone)
    // rows: b_o, ln g, ln b, b_1, b_2, then [5], [6]: the next stage's LayerNorm (QKV) or mlp_head's b_1 / b_2 (HEAD)
    constexpr int X16_VEC = 14 * C, X16_STG = 2 * X16_WAVES * X16_STG_WAVE;      // halves
    half_t* ring = smem + X16_VEC + X16_STG;
    float (*vec)[C] = reinterpret_cast<float (*)[C]>(smem);
    const int P = p.P, n_tiles = (P + X16_TOKENS - 1) / X16_TOKENS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tk = lane & 15, g = lane >> 4;
    float* stg = reinterpret_cast<float*>(smem + 14 * C) + wave * X16_STG_WAVE;
    const bool dyn = (OUTPROJ || !FFN) && p.pull != nullptr;                        // pulled tiles (tail16_pull)
    const bool drawer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
    volatile int* nx = reinterpret_cast<volatile int*>(smem + 14 * C + 2 * X16_WAVES * X16_STG_WAVE + X16_RING * X16_CHUNK);   // the next ticket
    int job = blockIdx.y, tile = 0;                                       // current tile (dyn: from the ticket)
    int next_job = 0, next_tile = 0;
    bool pending = false;                                                 // dyn: a tile of another job class follows (new segment)

    // ---- the live tiles of this workgroup (thread = (tile k, 8-token run): W is a multiple of 8, so a run lies in one window) ----
    unsigned live_mask = 0;
    if (dyn) {
        if (drawer) {
            const int t = tail16_pull(p, n_tiles);
            if (lane == 0) *nx = t;
        }
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(*nx);
        __syncthreads();
        if (t < 0) return;
        job = t / n_tiles;
        tile = t - job * n_tiles;
    } else {
        const FfnJob& J = p.job[blockIdx.y];
        unsigned* sl = reinterpret_cast<unsigned*>(smem + 14 * C);
        if (threadIdx.x == 0) *sl = 0;
        __syncthreads();
        const int k = threadIdx.x >> 4, t = blockIdx.x + k * gridDim.x, tok0 = t * X16_TOKENS + (threadIdx.x & 15) * 8;
        if (k < X16_MAX_TILES && t < n_tiles && tok0 < P) {
            int live = 1;
            if (J.need) {
                const int r = tok0 / p.W, c = tok0 - r * p.W;
                live = J.need[(r >> 3) * (p.W >> 3) + (c >> 3)];
            }
            if (live) atomicOr(sl, 1u << k);
        }
        __syncthreads();
        live_mask = __builtin_amdgcn_readfirstlane(*sl);
        __syncthreads();
        if (live_mask == 0) return;
    }

    // a SEGMENT = consecutive tiles of one job class (FfnJob::cls: same type, matrices and scales - everything the vector rows, the
    // chunk sequence and the ring depend on); without the pull there is one segment (the workgroup's job)
#pragma unroll 1
    for (;;) {
    const int ty = p.job[job].type, cls = p.job[job].cls;
    const QkvJob* Qj = QKV ? &qp->job[job] : nullptr;
    for (int i = threadIdx.x; i < C; i += X16_THREADS) {
        vec[0][i] = OUTPROJ ? p.b_o[ty * C + i] : 0.f;
        vec[1][i] = LN ? p.ln_g[ty * C + i] : 1.f;
        vec[2][i] = LN ? p.ln_b[ty * C + i] : 0.f;
        vec[3][i] = FFN ? p.b_1[ty * C + i] : 0.f;
        vec[4][i] = FFN ? p.b_2[ty * C + i] : 0.f;
        if constexpr (QKV) { vec[5][i] = qp->gamma[ty * C + i]; vec[6][i] = qp->beta[ty * C + i]; }
        if constexpr (HEAD) { vec[5][i] = p.hb_1[ty * C + i]; vec[6][i] = p.hb_2[ty * C + i]; }
    }
    // a tile as ONE chunk sequence: [out-projection 8] [FFN 16: W_1 tile, W_2 slice alternating] [tail: mlp_head 16 |
    // next stage's Q / K' / V' 8 per matrix]
    constexpr int N_OUT = OUTPROJ ? NCH : 0, N_FFN = FFN ? 2 * NH : 0;
    const half_t* wo = OUTPROJ ? p.w_o + (size_t)ty * NCH * X16_CHUNK : nullptr;
    const half_t* wf = FFN ? p.w_ffn + (size_t)ty * N_FFN * X16_CHUNK : nullptr;
    const half_t* wh = HEAD ? p.w_head + (size_t)ty * N_FFN * X16_CHUNK : nullptr;
    const int n_tail = QKV ? Qj->n_mat * NCH : (HEAD ? N_FFN : 0);
    const int n_total = N_OUT + N_FFN + n_tail;
    auto chunk_ptr = [&](int i) -> const half_t* {
        if (i < N_OUT) return wo + (size_t)i * X16_CHUNK;
        i -= N_OUT;
        if (i < N_FFN) return wf + (size_t)i * X16_CHUNK;
        i -= N_FFN;
        if constexpr (QKV) return Qj->w[i / NCH] + (size_t)(i % NCH) * X16_CHUNK;
        else if constexpr (HEAD) return wh + (size_t)i * X16_CHUNK;
        else return wf;                                  // TAIL 0: no chunk past the FFN (never reached)
    };
    int gs3 = 0;                                         // (chunks of the tiles done so far) mod 3: the ring does not restart
    auto slot = [&](int i) -> half_t* { return ring + ((gs3 + i) % X16_RING) * X16_CHUNK; };
    int cc = 0;                                          // chunk of the current step, within the tile
    bool has_next = false;                               // another live tile follows the current one
    bool full_wave = true;
    // A step = [products of chunk cc] barrier [everything else: accumulate / GELU / stage + store, request a chunk] barrier.
    // The two wavefronts of a SIMD (w, w + 4) would run the same half at the same time and leave the matrix pipe idle during
    // every second half, so group B (waves 4-7) runs ONE BARRIER BEHIND group A (an extra barrier before its first step, one
    // for A after its last): B's products overlap A's "everything else" and vice versa, with the same code on both sides.
    // Ring protocol under that skew (half h = barrier interval; A: products of chunk k in half 2k, rest in 2k + 1; B: 2k + 1,
    // 2k + 2): chunk k is read in halves 2k and 2k + 1.  A requests its pieces of chunk k + 2 in the rest-half of step k (slot of
    // chunk k - 1, whose last reader finished in half 2k - 1) and confirms chunk k + 1 at the end of that half; B requests its
    // pieces of chunk k + 3 in its rest-half of step k (half 2k + 2: slot of chunk k, read until half 2k + 1) and confirms
    // chunk k + 1 at the end of its products of step k (half 2k + 1) - so every piece of chunk k + 1 is confirmed before the
    // barrier that opens half 2k + 2.  Chunk numbers run on across tiles (k + 2 / k + 3 past the tile's last chunk are the next
    // tile's first ones).  vmcnt(n) = "all but the newest n operations": n counts the request made since (4) and, where a tile
    // flush (4 stores) was issued since, those too; anything else a wave has in flight (the next tile's rows) is older than the
    // request and only makes the wait stricter; partial waves (a store instruction may be skipped) wait for all.
    const bool grp_b = X16_DEPHASE && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= 4;
    // `extra`: row loads this wave issued since the request it confirms (the next tile's attention rows / this tile's residual
    // row): they stay in flight too, or every tile boundary would cost two memory latencies
    int extra = 0;                                       // pending for this wave's next counted wait
    auto ring_wait = [&](bool flush_since) {
        if ((cc + 2 < n_total || has_next) && full_wave) {
            // `extra` is 0, 16 (the 16 loads of a token row) or 64 (the 64 scalar loads of a row of an NCHW map: 4 + 64 is past the
            // counter's 6 bits, 63 is merely stricter); the counts below are exact for those and nothing else, so any other value
            // waits for everything (the walk's waits were only right for extra == 16, unchecked)
            if (extra == 0) { if (flush_since) ring_wait_newest8(); else ring_wait_newest4(); }
            else if (extra == 16) { if (flush_since) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
            else if (extra == 64) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else dma_wait();
        } else dma_wait();
        extra = 0;
    };
    auto step_begin = [&]() { X16_STAMP(cc, 0); };
    // end of the products of step cc; flush_prev: the rest-half of step cc - 1 stored a tile
    auto products_end = [&](bool flush_prev) {
        X16_STAMP(cc, 1);
        if constexpr (X16_DEPHASE) {
            if (grp_b) ring_wait(flush_prev);
            X16_STAMP(cc, 2);
            wg_barrier();
        }
        X16_STAMP(cc, 3);
    };
    bool b_owes = grp_b;                                 // group B owes a barrier: its extra first one, later a deferred one
    // end of step cc; flush_now: this rest-half stored a tile.  defer_b (round 6): the step is followed by work outside the ring
    // (a LayerNorm + operand conversion, the next tile's rows): group B does that work BEFORE this barrier (b_barrier() below) and
    // group A after it, so the two groups' ~5 k cycles of it run side by side instead of one after the other
    auto step_end = [&](bool flush_now, bool defer_b = false) {
        const int ahead = cc + 2 + (grp_b ? 1 : 0);
        if (ahead < n_total) stage_chunk16(chunk_ptr(ahead), slot(ahead));
        else if (has_next) stage_chunk16(chunk_ptr(ahead - n_total), slot(ahead));
        X16_STAMP(cc, 4);
        if (dyn && cc == 0 && drawer) {                  // the ticket of the tile after this one
            const int t = tail16_pull(p, n_tiles);
            if (lane == 0) *nx = t;
        }
        if (!grp_b) ring_wait(flush_now);
        if (HMVIT_X16_DEFER && defer_b && grp_b) b_owes = true;
        else wg_barrier();
        X16_STAMP(cc, 5);
        ++cc;
    };
    auto b_barrier = [&]() {                             // group B's deferred barrier (and its extra first one)
        if (b_owes) wg_barrier();
        b_owes = false;
    };
    stage_chunk16(chunk_ptr(0), slot(0));
    stage_chunk16(chunk_ptr(1), slot(1));
    if (grp_b && n_total > 2) stage_chunk16(chunk_ptr(2), slot(2));

    float4v xacc[16];
    half8 ah[8], al[8];
    bool have_o = false;                                 // xacc holds this tile's attention rows (requested during the tile before)
    bool have_x = false;                                 // TAIL 3: xacc holds this tile's input rows (requested during the tile before)
    // range normalisation (HmvitStageScales): uniform powers of two for the stage's own chain; the stand-alone mlp_head launch
    // (!OUTPROJ && !LN) multiplies the un-normalised row itself and takes per-token factors (head_token_scales)
    const float c_1s = p.c_1[ty], s_gs = p.s_g[ty], k_2s = p.k_2[ty], c_o = p.c_o[ty];

    dma_wait();
    __syncthreads();
    // (group B's extra first barrier: after the first tile's rows, where the deferred barrier of a tile's last step falls later on)
#pragma unroll 1
    for (;;) {
        if (!dyn) {
            tile = blockIdx.x + __builtin_ctz(live_mask) * gridDim.x;
            live_mask &= live_mask - 1;
            has_next = live_mask != 0;
            next_tile = has_next ? blockIdx.x + __builtin_ctz(live_mask) * gridDim.x : tile;
            next_job = job;
        } else {
            has_next = false;                            // known after the out-projection (read_ticket); not consulted before
        }
        const FfnJob& J = p.job[job];
        const QkvJob* Qt = QKV ? &qp->job[job] : nullptr;      // this tile's planes and scales
        const int tok_w = tile * X16_TOKENS + wave * 16, tok = tok_w + tk;
        const bool valid = tok < P;
        full_wave = tok_w + 16 <= P;
        const int tok_c = min(tok, P - 1);
        cc = 0;
        float c_1 = c_1s, s_g = s_gs, k_2 = k_2s;
        float b1_pre = 1.f;             // mlp_head only: the hidden accumulator starts from b_1 b1_pre = b_1 / c_1
        // mlp_head on a raw row: operand scale from the row's own maximum, the hidden bound from |h| <= l1 max|x| + max|b_1|
        auto head_token_scales = [&](float& s_tok) {
            s_tok = 1.f; c_1 = 1.f; s_g = 1.f; k_2 = 1.f; b1_pre = 1.f;
            if (p.dyn_head) {
                const float r = row_absmax16(xacc);
                s_tok = pow2_scale(r);
                b1_pre = p.head.w1[ty] * s_tok;
                c_1 = pow2_inv(b1_pre);
                s_g = pow2_scale(fmaf(p.head.l1[ty], r, p.head.b1max[ty]));
                k_2 = p.head.w2[ty] * s_g;
            }
        };
        // ---- operands of the out-projection: the tile's attention rows ----
        if constexpr (OUTPROJ && A16) {
            // attention output (f16, exact operand: no lo half) of this token
            const half_t* op = reinterpret_cast<const half_t*>(J.o) + (size_t)tok_c * C + 4 * g;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const half4 a = *reinterpret_cast<const half4*>(op + 32 * s);
                const half4 b = *reinterpret_cast<const half4*>(op + 32 * s + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) { ah[s][j] = a[j]; ah[s][4 + j] = b[j]; }
            }
        } else if constexpr (OUTPROJ) {
            // attention output (f32) of this token, in the residual row's layout: slot j of k-step s <- channel
            // 32 s + 16 (j >> 2) + 4 g + (j & 3) = row tile 2 s + (j >> 2), element j & 3
            if (!have_o) {
                const float* op = reinterpret_cast<const float*>(J.o) + (size_t)tok_c * C + 4 * g;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float4 f = *reinterpret_cast<const float4*>(op + 16 * t);
                    xacc[t][0] = f.x; xacc[t][1] = f.y; xacc[t][2] = f.z; xacc[t][3] = f.w;
                }
            }
            rows_to_operands16(xacc, ah, al, 1.f);
            have_o = false;
        }
        // ---- the residual row (TAIL 3: the input row, already requested during the tile before when have_x) ----
        if (!FFN && have_x) {
            have_x = false;
            extra = 0;
        } else {
        if constexpr (XN) {
            const unsigned long long a = (unsigned long long)J.x;
            int4v rs;
            rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
            rs.z = C * P * 4;
            rs.w = 0x00020000;
            const int voff = (4 * g * P + tok_c) * 4;
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) xacc[t][r] = llvm_raw_buffer_load_f32(rs, voff, (16 * t + r) * P * 4, 0);
        } else {
            const float* xp = J.x + (size_t)tok_c * C + 4 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 f = *reinterpret_cast<const float4*>(xp + 16 * t);
                xacc[t][0] = f.x; xacc[t][1] = f.y; xacc[t][2] = f.z; xacc[t][3] = f.w;
            }
        }
        extra = XN ? 64 : 16;
        }
        if constexpr (!OUTPROJ && !LN && FFN) {   // mlp_head: x itself is the operand
            float s_tok;
            head_token_scales(s_tok);
            rows_to_operands16(xacc, ah, al, s_tok);
        }
        b_barrier();

        // ---- phase 1: x' = x + b_o + W_o . O, two 16-channel tiles per chunk ----
        if constexpr (OUTPROJ) {
            // The two row tiles of chunk c are a run-time index into xacc in a rolled loop: one select per register of xacc and
            // iteration (64 selects per step, ~1300 of a step's 4500 cycles in the round-5 trace, against 3150 for an FFN step).
            // Unrolled HMVIT_X16_OUT_UNROLL chunks per iteration the selects are paid once per iteration; fully unrolled (8) the
            // kernel spills 20 registers (measured slower).
#ifndef HMVIT_X16_OUT_UNROLL
#define HMVIT_X16_OUT_UNROLL 2
#endif
            constexpr int OU = HMVIT_X16_OUT_UNROLL;
            static_assert(N_OUT % OU == 0, "out-projection unroll");
#pragma unroll 1
            for (int c0 = 0; c0 < N_OUT; c0 += OU) {
                float4v upd[OU][2];
#pragma unroll
                for (int u = 0; u < OU; ++u) {
                    const int c = c0 + u;
                    step_begin();
                    float4v acc[2];                                          // starts from b_o / c_o (pre-divided on the host)
#pragma unroll
                    for (int T = 0; T < 2; ++T) acc[T] = *reinterpret_cast<const float4v*>(&vec[0][32 * c + 16 * T + 4 * g]);
                    mma_proj16<!A16>(acc, slot(cc), ah, al, lane);          // f16 attention output: exact operand, no lo half
                    products_end(false);
                    upd[u][0] = acc[0] * c_o;
                    upd[u][1] = acc[1] * c_o;
                    if (u == OU - 1) {
#pragma unroll
                        for (int t0 = 0; t0 < 8; t0 += OU)
                            if (t0 == c0) {
#pragma unroll
                                for (int v = 0; v < OU; ++v) { xacc[2 * (t0 + v)] += upd[v][0]; xacc[2 * (t0 + v) + 1] += upd[v][1]; }
                            }
                    }
                    step_end(false, c == N_OUT - 1);
                }
            }
        }
        auto read_ticket = [&]() {
            const int t = __builtin_amdgcn_readfirstlane(*nx);
            pending = false;
            if (t >= 0) {
                next_job = t / n_tiles;
                next_tile = t - next_job * n_tiles;
                has_next = p.job[next_job].cls == cls;
                pending = !has_next;
            }
        };
        if (OUTPROJ && dyn) read_ticket();               // drawn in step 0, eight barriers ago
        if constexpr (LN) ln_to_operands16(xacc, vec[1], vec[2], g, ah, al);
        if constexpr (OUTPROJ) b_barrier();      // (behind the b_2 loop below hipcc spills 59 registers)

        // the row is carried as x k_2 while W_2's (scaled) products accumulate into it.  Stage chain: b_2 arrives as b_2 k_2;
        // mlp_head (no residual, per-token k_2): b_2 at its true scale
        if constexpr (FFN) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 b2 = *reinterpret_cast<const float4*>(&vec[4][16 * t + 4 * g]);
            if constexpr (RESID) {
                xacc[t][0] = fmaf(xacc[t][0], k_2, b2.x); xacc[t][1] = fmaf(xacc[t][1], k_2, b2.y);
                xacc[t][2] = fmaf(xacc[t][2], k_2, b2.z); xacc[t][3] = fmaf(xacc[t][3], k_2, b2.w);
            } else {
                xacc[t][0] = b2.x * k_2; xacc[t][1] = b2.y * k_2; xacc[t][2] = b2.z * k_2; xacc[t][3] = b2.w * k_2;
            }
        }
        }


        // ---- phase 2: per hidden tile hc: h = GELU(W_1[hc] . xn + b_1[hc]);  x'' += W_2[:, hc] . h ----
        // DYN (mlp_head on a raw row): per-token factors, b_1 (row `b1row` of vec) at its true scale; otherwise it arrives as b_1 / c_1
        auto ffn_pass = [&](auto dyn_c, const float* b1row) {
            constexpr bool DYN = decltype(dyn_c)::value;
#pragma unroll 1
            for (int hc = 0; hc < NH; ++hc) {
                step_begin();
                float4v hacc[2];                                 // starts from b_1 / c_1
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const float4 b1 = *reinterpret_cast<const float4*>(&b1row[32 * hc + 16 * T + 4 * g]);
                    if constexpr (DYN) {
                        hacc[T][0] = b1.x * b1_pre; hacc[T][1] = b1.y * b1_pre; hacc[T][2] = b1.z * b1_pre; hacc[T][3] = b1.w * b1_pre;
                    } else {
                        hacc[T][0] = b1.x; hacc[T][1] = b1.y; hacc[T][2] = b1.z; hacc[T][3] = b1.w;
                    }
                }
                mma_proj16(hacc, slot(cc), ah, al, lane);
                products_end(false);
                half8 hh, hl;                                    // hidden channel 32 hc + 16 (j >> 2) + 4 g + (j & 3) = hacc[j >> 2][j & 3]
                {
                    float gv[8];
                    const float2v c1v = {c_1, c_1}, sgv = {s_g, s_g};
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float2v hv = {hacc[j >> 2][j & 3], hacc[j >> 2][(j & 3) + 1]};
                        const float2v gl = gelu_f2(hv * c1v) * sgv;
                        gv[j] = gl.x; gv[j + 1] = gl.y;
                    }
                    split_pk8(gv, hh, hl);
                }
                step_end(false);
                step_begin();
                mma_slice16(xacc, slot(cc), hh, hl, lane);
                products_end(false);
                step_end(false, hc == NH - 1);           // followed by a conversion / the tile's end
            }
        };
        if constexpr (FFN) {
            ffn_pass(std::integral_constant<bool, !OUTPROJ && !LN>{}, vec[3]);
            const float k_inv = pow2_inv(k_2);
#pragma unroll
            for (int t = 0; t < 16; ++t) xacc[t] *= k_inv;
        }

        auto store_x = [&]() {
            if (valid && !(QKV && J.pad)) {
                if constexpr (OUT_NCHW) {
                    float* op = J.out + (size_t)(4 * g) * P + tok;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) op[(size_t)(16 * t + r) * P] = xacc[t][r];
                } else {
                    float* op = J.out + (size_t)tok * C + 4 * g;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
#ifdef HMVIT_EXP_X16_NOSTORE
                        if (xacc[t][0] == 1.2345e-30f)
#endif
                        *reinterpret_cast<float4*>(op + 16 * t) = make_float4(xacc[t][0], xacc[t][1], xacc[t][2], xacc[t][3]);
                }
            }
        };
        if constexpr (TAIL == 0) store_x();

        if constexpr (HEAD) {
            float s_tok;
            head_token_scales(s_tok);
            rows_to_operands16(xacc, ah, al, s_tok);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 b2 = *reinterpret_cast<const float4*>(&vec[6][16 * t + 4 * g]);
                xacc[t][0] = b2.x * k_2; xacc[t][1] = b2.y * k_2; xacc[t][2] = b2.z * k_2; xacc[t][3] = b2.w * k_2;
            }
            b_barrier();
            ffn_pass(std::integral_constant<bool, true>{}, vec[5]);
            {
                const float k_inv = pow2_inv(k_2);
#pragma unroll
                for (int t = 0; t < 16; ++t) xacc[t] *= k_inv;
            }
            if (valid) {
                float* op = J.out + (size_t)(4 * g) * P + tok;      // (C, P) map
#pragma unroll
                for (int t = 0; t < 16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) op[(size_t)(16 * t + r) * P] = xacc[t][r];
            }
        }

        if constexpr (QKV) {
            if (n_tail > 0) {
                ln_to_operands16(xacc, vec[5], vec[6], g, ah, al);
                store_x();                                   // x'' leaves while the first tiles are computed
                b_barrier();
                const float cq0 = Qt->c[0], cq1 = Qt->c[1], cq2 = Qt->c[2], cq3 = Qt->c[3], cq4 = Qt->c[4];
                for (int c = 0; c < n_tail; ++c) {
                    const int mat = c / NCH, t = c - mat * NCH;
                    const float cm = mat == 0 ? cq0 : mat == 1 ? cq1 : mat == 2 ? cq2 : mat == 3 ? cq3 : cq4;
                    // (the 16 stores of x'' are older than the first request made after them: that wait covers them too)
                    auto flushes = [](int k) { return k >= 0 && (A16 ? (k & 3) == 3 : (k & 1) != 0); };
                    step_begin();
                    if constexpr (OUTPROJ && !A16) {
                        // the next tile's attention rows into the residual registers (idle since store_x): requested at the head
                        // of a step, i.e. right behind the counted wait of the step before, and landed a step later
                        if (c == 1 && has_next) {
                            const int ntok = min(next_tile * X16_TOKENS + wave * 16 + tk, P - 1);
                            const float* op = reinterpret_cast<const float*>(p.job[next_job].o) + (size_t)ntok * C + 4 * g;
#pragma unroll
                            for (int t2 = 0; t2 < 16; ++t2) {
                                const float4 f = *reinterpret_cast<const float4*>(op + 16 * t2);
                                xacc[t2][0] = f.x; xacc[t2][1] = f.y; xacc[t2][2] = f.z; xacc[t2][3] = f.w;
                            }
                            have_o = true;
                            extra = 16;
                        }
                    }
                    if constexpr (!FFN) {
                        // TAIL 3: the ticket drawn in step 0 (one barrier ago), then the next tile's input rows into xacc (idle since
                        // the LayerNorm and store_x)
                        if (c == 1) {
                            if (dyn) read_ticket();
                            if (has_next) {
                                const int ntok = min(next_tile * X16_TOKENS + wave * 16 + tk, P - 1);
                                const float* nxp = p.job[next_job].x;
                                if constexpr (XN) {
                                    const unsigned long long a = (unsigned long long)nxp;
                                    int4v rs;
                                    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
                                    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
                                    rs.z = C * P * 4;
                                    rs.w = 0x00020000;
                                    const int voff = (4 * g * P + ntok) * 4;
#pragma unroll
                                    for (int t2 = 0; t2 < 16; ++t2)
#pragma unroll
                                        for (int r = 0; r < 4; ++r) xacc[t2][r] = llvm_raw_buffer_load_f32(rs, voff, (16 * t2 + r) * P * 4, 0);
                                } else {
                                    const float* xp2 = nxp + (size_t)ntok * C + 4 * g;
#pragma unroll
                                    for (int t2 = 0; t2 < 16; ++t2) {
                                        const float4 f = *reinterpret_cast<const float4*>(xp2 + 16 * t2);
                                        xacc[t2][0] = f.x; xacc[t2][1] = f.y; xacc[t2][2] = f.z; xacc[t2][3] = f.w;
                                    }
                                }
                                have_x = true;
                                extra = XN ? 64 : 16;
                            }
                        }
                    }
                    float4v acc[2] = {(float4v)(0.f), (float4v)(0.f)};
                    mma_proj16(acc, slot(cc), ah, al, lane);
                    products_end(flushes(c - 1));
                    if constexpr (A16) store_proj16_h(stg, reinterpret_cast<half_t*>(Qt->y[mat]), t, acc, cm, tk, g, lane, tok_w, P);
                    else store_proj16(stg, reinterpret_cast<float*>(Qt->y[mat]), t, acc, cm, tk, g, lane, tok_w, P);
                    step_end(flushes(c), c == n_tail - 1);
                }
            } else {
                store_x();
            }
        }
        gs3 = (gs3 + n_total) % X16_RING;
        if (dyn) {
            if (!has_next) break;
            job = next_job;
            tile = next_tile;
        } else if (!live_mask) break;
    }
    b_barrier();                                         // the deferred barrier of the last tile's last step
    if (X16_DEPHASE && !grp_b) wg_barrier();             // group A's share of group B's extra first barrier
    if (!(dyn && pending)) break;
    job = next_job;                                      // next segment: rows, chunk sequence and ring start over
    tile = next_tile;
    __syncthreads();
    }
}

template <bool OUTPROJ, bool LN, bool RESID, bool OUT_NCHW, bool A16>
__global__ __launch_bounds__(X16_THREADS, 2) void k_out_ffn16(FfnParams p) {
    tail16_body<OUTPROJ, LN, RESID, OUT_NCHW, 0, false, A16>(p, nullptr);
}
template <bool XN, bool A16>
__global__ __launch_bounds__(X16_THREADS, 2) void k_out_ffn_qkv16(FfnParams p, QkvParams q) {
    tail16_body<true, true, true, false, 1, XN, A16>(p, &q);
}
// k_ln_qkv16 as pulled tiles: the projection phase of tail16_body alone (TAIL 3); FfnParams carries the jobs' input rows (x, XN =
// (C, P) maps), the token-major copy (out; pad = 1: none) and the pull counter
template <bool XN, bool A16>
__global__ __launch_bounds__(X16_THREADS, 2) void k_ln_qkv16p(FfnParams p, QkvParams q) {
    tail16_body<false, false, false, false, 3, XN, A16>(p, &q);
}
template <bool A16>
__global__ __launch_bounds__(X16_THREADS, 2) void k_out_ffn_head16(FfnParams p) {
    tail16_body<true, true, true, false, 2, false, A16>(p, nullptr);
}

template <int C, bool OUTPROJ, bool LN, bool RESID, bool OUT_NCHW, bool SP>
__global__ __launch_bounds__(CHAIN_THREADS, SP ? 1 : 2) void k_out_ffn(FfnParams p) {
    out_ffn_body<C, OUTPROJ, LN, RESID, OUT_NCHW, 0, false, SP>(p, nullptr);
}

template <int C, bool XN, bool SP>
__global__ __launch_bounds__(CHAIN_THREADS, SP ? 1 : 2) void k_out_ffn_qkv(FfnParams p, QkvParams q) {
    out_ffn_body<C, true, true, true, false, 1, XN, SP>(p, &q);
}

// last stage of HeteroFusion: the ego's tail with mlp_head appended (FfnJob::out = the (C, P) output map)
template <int C, bool SP>
__global__ __launch_bounds__(CHAIN_THREADS, SP ? 1 : 2) void k_out_ffn_head(FfnParams p) {
    out_ffn_body<C, true, true, true, false, 2, false, SP>(p, nullptr);
}

// ------------------------------------------------------------------------------------------
// launchers (split = the "split" precision mode: hi / lo weight images, f32 Q / K' / V' / O planes)
// ------------------------------------------------------------------------------------------
// pulled tiles: job classes (a workgroup keeps its vector rows, chunk sequence and ring across tiles of one class) and the job count
constexpr int kX16Persistent = 256;    // one workgroup per CU (140 KB of LDS each)
static void x16_job_classes(FfnParams& p, const QkvParams* q, int n_jobs) {
    p.n_jobs = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        int c = j;
        for (int k = 0; k < j && c == j; ++k) {
            bool same = p.job[k].type == p.job[j].type && p.job[k].x_nchw == p.job[j].x_nchw;
            if (same && q) {
                const QkvJob &a = q->job[k], &b = q->job[j];
                same = a.n_mat == b.n_mat;
                for (int m = 0; same && m < a.n_mat; ++m) same = a.w[m] == b.w[m] && a.c[m] == b.c[m];
            }
            if (same) c = p.job[k].cls;
        }
        p.job[j].cls = c;
    }
}

template <bool SP>
static int launch_ln_qkv_t(const QkvParams& p, int n_jobs, int C, hipStream_t st) {
    dim3 grid(cdiv(p.P, CHAIN_TOKENS), n_jobs), block(CHAIN_THREADS);
    switch (C) {
        case 64: hipLaunchKernelGGL((k_ln_qkv<64, SP>), grid, block, 0, st, p); break;
        case 128: hipLaunchKernelGGL((k_ln_qkv<128, SP>), grid, block, 0, st, p); break;
        case 256: hipLaunchKernelGGL((k_ln_qkv<256, SP>), grid, block, 0, st, p); break;
        default: set_error("ln_qkv: C=%d unsupported", C); return HMVIT_EINVAL;
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
int launch_ln_qkv(const QkvParams& p, int n_jobs, int C, int split, hipStream_t st) {
    if (n_jobs == 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(split != 2 || C == 256, "mixed precision planes need C = 256 (got %d)", C);
    if (split && C == 256 && p.pull && !HMVIT_ENV("HMVIT_X16_STATIC")) {   // pulled tiles (tail16_pull), one workgroup per CU
        FfnParams f;
        std::memset(&f, 0, sizeof(f));
        f.P = p.P; f.pull = p.pull;
        set_ffn_scales(f, nullptr, nullptr);
        for (int j = 0; j < n_jobs; ++j) {
            f.job[j].x = p.job[j].x; f.job[j].out = p.job[j].xs_out; f.job[j].pad = p.job[j].xs_out ? 0 : 1;
            f.job[j].type = p.job[j].type; f.job[j].x_nchw = p.in_nchw;
        }
        x16_job_classes(f, &p, n_jobs);
        const dim3 grid(std::min(kX16Persistent, n_jobs * cdiv(p.P, X16_TOKENS)), 1), block(X16_THREADS);
        if (p.in_nchw) {
            if (split == 2) hipLaunchKernelGGL((k_ln_qkv16p<true, true>), grid, block, 0, st, f, p);
            else hipLaunchKernelGGL((k_ln_qkv16p<true, false>), grid, block, 0, st, f, p);
        } else {
            if (split == 2) hipLaunchKernelGGL((k_ln_qkv16p<false, true>), grid, block, 0, st, f, p);
            else hipLaunchKernelGGL((k_ln_qkv16p<false, false>), grid, block, 0, st, f, p);
        }
        HMVIT_CHECK_LAUNCH();
        return HMVIT_OK;
    }
    if (split && C == 256) {      // 16 tokens per wavefront, images in the x16 layout (weights.py weight_image16)
        const dim3 grid(cdiv(p.P, X16_TOKENS), n_jobs), block(X16_THREADS);
        if (split == 2) hipLaunchKernelGGL(k_ln_qkv16<true>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(k_ln_qkv16<false>, grid, block, 0, st, p);
        HMVIT_CHECK_LAUNCH();
        return HMVIT_OK;
    }
    return split ? launch_ln_qkv_t<true>(p, n_jobs, C, st) : launch_ln_qkv_t<false>(p, n_jobs, C, st);
}

template <int C, bool SP>
static int launch_out_ffn_c(const FfnParams& p, int n_jobs, int variant, hipStream_t st) {
    dim3 grid(cdiv(p.P, CHAIN_TOKENS), n_jobs), block(CHAIN_THREADS);
    switch (variant) {
        case FFN_FULL: hipLaunchKernelGGL((k_out_ffn<C, true, true, true, false, SP>), grid, block, 0, st, p); break;
        case FFN_NO_ATTN: hipLaunchKernelGGL((k_out_ffn<C, false, true, true, false, SP>), grid, block, 0, st, p); break;
        case FFN_HEAD_NCHW: hipLaunchKernelGGL((k_out_ffn<C, false, false, false, true, SP>), grid, block, 0, st, p); break;
        default: set_error("out_ffn: bad variant %d", variant); return HMVIT_EINVAL;
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// Workgroups per job of the x16 tails.  The kernels can walk several tiles per workgroup (tail16_body), but the shipped grid is
// ONE TILE PER WORKGROUP: measured at cfg2 on one box (tools/probe/run_libs.sh, four tails of a forward): 5.55 ms with 1100
// workgroups per job, 5.70 with 204 (5.4 tiles each), 5.96 with 102, 6.05-6.10 with 51 (one workgroup per CU, 22 tiles each) -
// with and without the row loads of a tile boundary kept in flight.  What the walk saves (inputs and first chunks arriving with
// nothing to overlap them, once per tile) is less than what a static partition loses to the spread of the CUs' speeds: the
// hardware dispatcher hands the next tile to whichever CU is free.  -DHMVIT_X16_GX=n / HMVIT_X16_GRID (probe builds) set the
// walk's width for measurements.
static int x16_grid_x(int P, int n_jobs) {
    const int n_tiles = cdiv(P, X16_TOKENS);
    int gx = n_tiles;
#ifdef HMVIT_X16_GX
    gx = HMVIT_X16_GX;
#endif
    if (const char* e = HMVIT_ENV("HMVIT_X16_GRID")) gx = atoi(e);
    if (gx < 1) gx = 1;
    gx = gx > cdiv(n_tiles, X16_MAX_TILES) ? gx : cdiv(n_tiles, X16_MAX_TILES);
    return gx < n_tiles ? gx : n_tiles;
}

int launch_out_ffn_qkv(const FfnParams& p, const QkvParams& q, int n_jobs, int C, int split, hipStream_t st) {
    if (n_jobs == 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(C == 256, "out_ffn_qkv: C=%d unsupported (256)", C);
    dim3 grid(cdiv(p.P, CHAIN_TOKENS), n_jobs), block(CHAIN_THREADS);
    // FfnJob::x_nchw (all jobs of a launch alike): the residual is read from (C, P) maps
    if (split && p.pull && !HMVIT_ENV("HMVIT_X16_STATIC")) {   // pulled tiles: one workgroup per CU (tail16_pull)
        FfnParams pp = p;
        x16_job_classes(pp, &q, n_jobs);
        const int n_wg = std::min(kX16Persistent, n_jobs * cdiv(p.P, X16_TOKENS));
        const dim3 grid16(n_wg, 1), block16(X16_THREADS);
        if (split == 2) {
            if (p.job[0].x_nchw) hipLaunchKernelGGL((k_out_ffn_qkv16<true, true>), grid16, block16, 0, st, pp, q);
            else hipLaunchKernelGGL((k_out_ffn_qkv16<false, true>), grid16, block16, 0, st, pp, q);
        } else {
            if (p.job[0].x_nchw) hipLaunchKernelGGL((k_out_ffn_qkv16<true, false>), grid16, block16, 0, st, pp, q);
            else hipLaunchKernelGGL((k_out_ffn_qkv16<false, false>), grid16, block16, 0, st, pp, q);
        }
        HMVIT_CHECK_LAUNCH();
        return HMVIT_OK;
    }
    if (split) {
        const dim3 grid16(x16_grid_x(p.P, n_jobs), n_jobs), block16(X16_THREADS);
        if (split == 2) {
            if (p.job[0].x_nchw) hipLaunchKernelGGL((k_out_ffn_qkv16<true, true>), grid16, block16, 0, st, p, q);
            else hipLaunchKernelGGL((k_out_ffn_qkv16<false, true>), grid16, block16, 0, st, p, q);
        } else {
            if (p.job[0].x_nchw) hipLaunchKernelGGL((k_out_ffn_qkv16<true, false>), grid16, block16, 0, st, p, q);
            else hipLaunchKernelGGL((k_out_ffn_qkv16<false, false>), grid16, block16, 0, st, p, q);
        }
    } else {
        if (p.job[0].x_nchw) hipLaunchKernelGGL((k_out_ffn_qkv<256, true, false>), grid, block, 0, st, p, q);
        else hipLaunchKernelGGL((k_out_ffn_qkv<256, false, false>), grid, block, 0, st, p, q);
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_out_ffn_head(const FfnParams& p, int n_jobs, int C, int split, hipStream_t st) {
    if (n_jobs == 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(C == 256 && p.w_head && p.hb_1 && p.hb_2, "out_ffn_head: C=%d (256) / head weights missing", C);
    dim3 grid(cdiv(p.P, CHAIN_TOKENS), n_jobs), block(CHAIN_THREADS);
    if (split && p.pull && !HMVIT_ENV("HMVIT_X16_STATIC")) {   // pulled tiles: one workgroup per CU (tail16_pull)
        FfnParams pp = p;
        x16_job_classes(pp, nullptr, n_jobs);
        const dim3 grid16(std::min(kX16Persistent, n_jobs * cdiv(p.P, X16_TOKENS)), 1);
        if (split == 2) hipLaunchKernelGGL(k_out_ffn_head16<true>, grid16, dim3(X16_THREADS), 0, st, pp);
        else hipLaunchKernelGGL(k_out_ffn_head16<false>, grid16, dim3(X16_THREADS), 0, st, pp);
        HMVIT_CHECK_LAUNCH();
        return HMVIT_OK;
    }
    if (split == 2) hipLaunchKernelGGL(k_out_ffn_head16<true>, dim3(x16_grid_x(p.P, n_jobs), n_jobs), dim3(X16_THREADS), 0, st, p);
    else if (split) hipLaunchKernelGGL(k_out_ffn_head16<false>, dim3(x16_grid_x(p.P, n_jobs), n_jobs), dim3(X16_THREADS), 0, st, p);
    else hipLaunchKernelGGL((k_out_ffn_head<256, false>), grid, block, 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_out_ffn(const FfnParams& p, int n_jobs, int C, int variant, int split, hipStream_t st) {
    if (n_jobs == 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(split != 2 || C == 256, "mixed precision planes need C = 256 (got %d)", C);
    if (split && C == 256) {
        const dim3 grid(x16_grid_x(p.P, n_jobs), n_jobs), block(X16_THREADS);
        switch (variant) {
            case FFN_FULL:
                if (split == 2) hipLaunchKernelGGL((k_out_ffn16<true, true, true, false, true>), grid, block, 0, st, p);
                else hipLaunchKernelGGL((k_out_ffn16<true, true, true, false, false>), grid, block, 0, st, p);
                break;
            case FFN_NO_ATTN: hipLaunchKernelGGL((k_out_ffn16<false, true, true, false, false>), grid, block, 0, st, p); break;
            case FFN_HEAD_NCHW: hipLaunchKernelGGL((k_out_ffn16<false, false, false, true, false>), grid, block, 0, st, p); break;
            default: set_error("out_ffn: bad variant %d", variant); return HMVIT_EINVAL;
        }
        HMVIT_CHECK_LAUNCH();
        return HMVIT_OK;
    }
    switch (C) {
        case 64: return split ? launch_out_ffn_c<64, true>(p, n_jobs, variant, st) : launch_out_ffn_c<64, false>(p, n_jobs, variant, st);
        case 128: return split ? launch_out_ffn_c<128, true>(p, n_jobs, variant, st) : launch_out_ffn_c<128, false>(p, n_jobs, variant, st);
        case 256: return split ? launch_out_ffn_c<256, true>(p, n_jobs, variant, st) : launch_out_ffn_c<256, false>(p, n_jobs, variant, st);
        default: set_error("out_ffn: C=%d unsupported", C); return HMVIT_EINVAL;
    }
}

int launch_linear16(const LinJobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return HMVIT_OK;
    int max_m = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const LinJob& j = jobs.j[i];
        HMVIT_CHECK_ARG(j.a && j.M > 0 && j.n_mat >= 1 && j.n_mat <= kMaxLinMats && j.ldy >= 256 && j.ldy % 4 == 0, "linear16: bad job %d", i);
        HMVIT_CHECK_ARG(!j.residual || j.n_mat == 1 || j.sum_inputs, "linear16: a residual goes with a single output (job %d)", i);
        HMVIT_CHECK_ARG(!j.sum_inputs || (!j.ln_gamma && j.drop.p == 0.f), "linear16: the input sum takes no LayerNorm / dropout (job %d)", i);
        HMVIT_CHECK_ARG(j.drop.p == 0.f || (j.n_mat == 1 && j.residual && j.drop.p > 0.f && j.drop.p < 1.f),
                        "linear16: dropout (p=%f) goes with one matrix and a residual (job %d)", (double)j.drop.p, i);
        HMVIT_CHECK_ARG((j.ln_gamma == nullptr) == (j.ln_beta == nullptr), "linear16: LayerNorm needs gamma and beta (job %d)", i);
        for (int m = 0; m < j.n_mat; ++m)
            HMVIT_CHECK_ARG(j.wimg[m] && j.w_inv[m] && (j.sum_inputs ? (m == 0 ? j.y[0] != nullptr : j.a_more[m - 1] != nullptr) : j.y[m] != nullptr),
                            "linear16: null pointer (job %d, matrix %d)", i, m);
        max_m = j.M > max_m ? j.M : max_m;
    }
    hipLaunchKernelGGL(k_linear16, dim3(cdiv(max_m, X16_TOKENS), jobs.n), dim3(X16_THREADS), 0, st, jobs);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_weight_images16(const float* w, half_t* img, float* w_inv, int n_mat, hipStream_t st) {
    if (n_mat <= 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(w && img && w_inv, "weight_images16: null pointer%s", "");
    hipLaunchKernelGGL(k_weight_image16, dim3(n_mat), dim3(1024), 0, st, w, img, w_inv);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

#ifdef HMVIT_PROBE
int debug_x16_trace(unsigned long long* host, int n) {
    HMVIT_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x16_trace), sizeof(unsigned long long) * (n < 1024 ? n : 1024)));
    return HMVIT_OK;
}
#endif

}  // namespace hmvit
