// Fused H3GAT window / grid attention for one (sample, ego, window, head group) per workgroup.
//
// Replaces, for every ego i of HeteroFusionBlock.{local,global}_spatial_multi_agent_attention
// (hetero_fusion.py:363-444): warp_features (:338-361, L^2 bilinear BEV warps + nearest ROI
// masks), the window / dilated-grid rearranges (:384-394, :427-434) and HeteroAttention.forward
// (:187-277) up to, not including, the output projection.  Nothing is materialised: the K / V
// tiles of a window are bilinear-gathered straight from the per-source projected maps
// (Linear(warp(x)) == warp(x W^T) + b, SURVEY.md 8a identity (i)) into LDS.
//
// Work decomposition: one wavefront per head (dim_head = 32), HG heads per workgroup.
// Keys are consumed in chunks of 64 (one source agent for window 8, four for window 4) with an
// online softmax across chunks.  Per chunk and 16-query tile the wave computes the transposed
// logits S^T = K Q^T (rows = keys, cols = queries) so that every lane owns ONE query column:
// the softmax reductions are in-register plus two cross-lane shuffles, and exp(S^T) is already
// in the B-operand layout of the second product O^T = V^T P^T.
//   f16 mode: v_mfma_f32_16x16x32_f16; V^T fragments come from ds_read_b64_tr_b16.
//   f32 mode: v_mfma_f32_16x16x4_f32 (exact f32), element-granular operands, no transposes.
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

template <typename T, int HG>
struct AttnCfg;
template <int HG>
struct AttnCfg<half_t, HG> {
    static constexpr int CH = HG * 32;       // channels of this head group
    static constexpr int QS = CH + 8;        // LDS row strides in elements
    static constexpr int KS = CH + 8;
    static constexpr int VS = CH + 16;
};
template <int HG>
struct AttnCfg<float, HG> {
    static constexpr int CH = HG * 32;
    static constexpr int QS = CH + 2;
    static constexpr int KS = CH + 2;
    static constexpr int VS = CH + 4;
};

// 8 consecutive channels of one token
template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ p, float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        const half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    } else {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}

template <typename T>
__device__ __forceinline__ void store8_lds(T* p, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
        *reinterpret_cast<half8*>(p) = h;
    } else {
        // f32 tiles use odd-ish strides (2 / 4 mod 32): 8-byte aligned only
#pragma unroll
        for (int e = 0; e < 8; e += 2) *reinterpret_cast<float2*>(p + e) = make_float2(v[e], v[e + 1]);
    }
}

// 8 channels of one token, kept in the storage type until all loads of a batch are issued
template <typename T>
struct Raw8;
template <>
struct Raw8<half_t> {
    half8 h;
    __device__ __forceinline__ void load(const half_t* __restrict__ p) { h = *reinterpret_cast<const half8*>(p); }
    __device__ __forceinline__ float get(int e) const { return (float)h[e]; }
};
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* __restrict__ p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ float get(int e) const {
        switch (e) {
            case 0: return a.x; case 1: return a.y; case 2: return a.z; case 3: return a.w;
            case 4: return b.x; case 5: return b.y; case 6: return b.z; default: return b.w;
        }
    }
};

// Bilinear sample (or direct read) of 8 channels of NP projected maps (planes `pstride` elements
// apart) + bias.  All tap loads are issued before the first use (no per-tap branches): taps with
// zero weight read a clamped, valid address.
template <typename T, int NP>
__device__ __forceinline__ void sample8(const T* __restrict__ plane, size_t pstride, int C, int ch, const Taps& t,
                                        bool ident, int self_idx, const float* __restrict__ bias, int bstride,
                                        float (&out)[NP][8]) {
    if (ident) {
        Raw8<T> raw[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) raw[pl].load(plane + pl * pstride + (size_t)self_idx * C + ch);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int e = 0; e < 8; ++e) out[pl][e] = raw[pl].get(e) + bias[pl * bstride + ch + e];
    } else {
        Raw8<T> raw[NP][4];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[pl][k].load(plane + pl * pstride + (size_t)t.idx[k] * C + ch);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = t.w[0] * raw[pl][0].get(e);
#pragma unroll
                for (int k = 1; k < 4; ++k) acc = fmaf(t.w[k], raw[pl][k].get(e), acc);
                out[pl][e] = acc + bias[pl * bstride + ch + e];
            }
    }
}

template <typename T, int WIN, int HG>
__global__ __launch_bounds__(HG * 64, (sizeof(T) == 2 ? 2 : 1)) void k_attention(AttnParams p) {
    using Cfg = AttnCfg<T, HG>;
    constexpr int N = WIN * WIN;          // tokens per window
    constexpr int NQT = N / 16;           // 16-query tiles
    constexpr int SPC = 64 / N;           // source agents per 64-key chunk
    constexpr int NB = (WIN == 8) ? 7 : 1;
    constexpr int CH = Cfg::CH, QS = Cfg::QS, KS = Cfg::KS, VS = Cfg::VS;
    constexpr int TPK = CH / 8;           // lanes cooperating on one key row
    constexpr int KPP = HG * 64 / TPK;    // key rows gathered per pass (= 16)
    constexpr bool F16 = sizeof(T) == 2;

    __shared__ __attribute__((aligned(16))) T Qs[N * QS];
    __shared__ __attribute__((aligned(16))) T Ks[64 * KS];
    __shared__ __attribute__((aligned(16))) T Vs[64 * VS];
    __shared__ __attribute__((aligned(16))) float maskadd[64];

    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / WIN, Y = W / WIN;
    const int NG = C / CH;
    const int win = blockIdx.x / NG, hg = blockIdx.x - win * NG;
    const int ego = blockIdx.y, b = blockIdx.z;
    const int wx = win / Y, wy = win - wx * Y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lq = lane & 15, g = lane >> 4;
    const int head = hg * HG + wave;
    const int te = p.mode[b * L + ego];
    const int ev = p.ego_e[b * L + ego];
    const int ch0 = hg * CH;               // first channel of this head group
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;   // f32 planes carried at a power of two (HmvitStageScales::k_logit)

    const T* qplanes = reinterpret_cast<const T*>(p.q);
    const T* kvplanes = reinterpret_cast<const T*>(p.kv);

    // ---- gather the query tile (ego's own map through T[i,i], normally the identity) ----
    {
        const float* a = p.ainv + ((size_t)(b * L + ego) * L + ego) * 8;
        const bool ident = a[6] != 0.f;
        const T* plane = qplanes + (size_t)(b * L + ego) * P * C;
        const float* bq = p.b_q + te * C + ch0;
        const int cl = (tid % TPK) * 8;
        for (int n = tid / TPK; n < N; n += KPP) {
            int row, col;
            token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
            Taps t;
            if (!ident) t = make_taps(a, col, row, H, W);
            float v[1][8];
            sample8<T, 1>(plane + ch0, 0, C, cl, t, ident, row * W + col, bq, 0, v);
            store8_lds<T>(Qs + n * QS + cl, v[0]);
        }
    }

    // relative-position bias fragments of this head (accumulator layout)
    float4v biasf[NB];
#pragma unroll
    for (int v = 0; v < NB; ++v)
        biasf[v] = *reinterpret_cast<const float4v*>(p.bias_frag + ((size_t)(head * NB + v) * 64 + lane) * 4);

    __syncthreads();

    // query fragments (B operand: k = channel, col = query)
    half8 qh[NQT];
    float qf[NQT][8];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        if constexpr (F16) {
            qh[qt] = *reinterpret_cast<const half8*>(Qs + (qt * 16 + lq) * QS + wave * 32 + g * 8);
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                qf[qt][ks] = reinterpret_cast<const float*>(Qs)[(qt * 16 + lq) * QS + wave * 32 + ks * 4 + g];
        }
    }

    float m_run[NQT], l_run[NQT];
    float4v o_acc[NQT][2];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o_acc[qt][dt] = (float4v)(0.f);
    }

    const int n_chunks = (p.n_src + SPC - 1) / SPC;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        // ---------------- gather 64 keys x CH channels of K and V ----------------
        int any_visible = 0;
        {
            const int cl = (tid % TPK) * 8;
#pragma unroll
            for (int pass = 0; pass < 64 / KPP; ++pass) {
                const int kk = pass * KPP + tid / TPK;
                const int src = chunk * SPC + kk / N;
                const int n = kk % N;
                float kvv[2][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) kvv[0][e] = kvv[1][e] = 0.f;
                bool visible = false;
                if (src < p.n_src) {
                    int row, col;
                    token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
                    const float* a = p.ainv + ((size_t)(b * L + src) * L + ego) * 8;
                    const bool ident = a[6] != 0.f;
                    Taps t;
                    if (ident) {                       // the key's own pixel with weight 1: the same four-tap arithmetic gives it back exactly
                        t.roi = 1.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { t.idx[k] = row * W + col; t.w[k] = k == 0 ? 1.f : 0.f; }
                    } else {
                        t = make_taps(a, col, row, H, W);
                    }
                    visible = (t.roi != 0.f) && (p.cav[b * L + src] != 0);
                    // No divergent region around the tap loads (train.hip k_attention_bwd: a gather behind `if (visible)` was not
                    // reproducible at two workgroups per CU): a wave without a visible key skips, which is wave-uniform; otherwise every
                    // lane loads - pixel 0 where its key is invisible - and the result is selected afterwards.
                    if (__any(visible)) {
                        const int ts = p.mode[b * L + src];
                        const T* kpl = kvplanes + ((size_t)((b * L + src) * p.E + ev) * 2) * P * C + ch0;
                        const float* bk = p.b_kv + (size_t)(te * HMVIT_NUM_TYPES + ts) * 2 * C + ch0;
                        if (!visible) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) t.idx[k] = 0;
                        }
                        // identity source for every lane of the wave (wave-uniform): one direct read per plane instead of four
                        // loads of the same pixel; mixed waves (window 4: several sources per chunk) take the tap path
                        const bool wave_ident = __all(ident);
                        sample8<T, 2>(kpl, (size_t)P * C, C, cl, t, wave_ident, t.idx[0], bk, C, kvv);
                        if (!visible) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) kvv[0][e] = kvv[1][e] = 0.f;
                        }
                    }
                }
                store8_lds<T>(Ks + kk * KS + cl, kvv[0]);
                store8_lds<T>(Vs + kk * VS + cl, kvv[1]);
                if (cl == 0) maskadd[kk] = visible ? 0.f : -INFINITY;
                any_visible |= visible ? 1 : 0;
            }
        }
        any_visible = __syncthreads_or(any_visible);

        if (any_visible || !p.skip_masked) {
            float4v madd[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) madd[kt] = *reinterpret_cast<const float4v*>(maskadd + kt * 16 + 4 * g);

            // operand fragments shared by all query tiles of this chunk
            half8 kh[4], vh[2][2];
            float kf[4][8], vf[4][4][2];
            if constexpr (F16) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    kh[kt] = *reinterpret_cast<const half8*>(Ks + (kt * 16 + lq) * KS + wave * 32 + g * 8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // 4 keys x 16 channels block per 16-lane group, transposed on read
                        const T* base = Vs + (ks * 32 + 4 * g + (lq >> 2)) * VS + wave * 32 + dt * 16 + (lq & 3) * 4;
                        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_t*)(base));
                        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                        half8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (half_t)lo[e];
                            v[4 + e] = (half_t)hi[e];
                        }
                        vh[dt][ks] = v;
                    }
            } else {
                const float* Kf = reinterpret_cast<const float*>(Ks);
                const float* Vf = reinterpret_cast<const float*>(Vs);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) kf[kt][ks] = Kf[(kt * 16 + lq) * KS + wave * 32 + ks * 4 + g];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            vf[kt][r][dt] = Vf[(kt * 16 + 4 * g + r) * VS + wave * 32 + dt * 16 + lq];
            }

#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                // S^T tiles: rows = keys kt*16 + 4g + r, col = query qt*16 + lq
                float4v s[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int bv = (WIN == 8) ? (qt - kt + 3) : 0;
                    float4v acc = biasf[bv];
                    if constexpr (F16) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[kt], qh[qt], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][ks], qf[qt][ks], acc, 0, 0, 0);
                    }
                    s[kt] = acc + madd[kt];
                }
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qt], mx);
                const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                float alpha, rs = 0.f;
                if constexpr (F16) alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_safe); else alpha = expf((m_run[qt] - m_safe) * kl);   // f16 mode: logits are in log2 units
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e;
                        if constexpr (F16) e = __builtin_amdgcn_exp2f(s[kt][r] - m_safe); else e = expf((s[kt][r] - m_safe) * kl);
                        s[kt][r] = e;
                        rs += e;
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run[qt] = l_run[qt] * alpha + rs;
                m_run[qt] = m_new;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o_acc[qt][dt] *= alpha;

                // O^T += V^T P^T
                if constexpr (F16) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 ph;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ph[e] = (half_t)s[2 * ks][e];
                            ph[4 + e] = (half_t)s[2 * ks + 1][e];
                        }
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt][ks], ph, o_acc[qt][dt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int dt = 0; dt < 2; ++dt)
                                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][r][dt], s[kt][r], o_acc[qt][dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- normalise and store: lane holds channels dt*16 + 4g + (0..3) of query qt*16 + lq ----
    T* outp = reinterpret_cast<T*>(p.out) + (size_t)(b * L + ego) * P * C;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        int row, col;
        token_pixel(p.partition, WIN, X, Y, wx, wy, qt * 16 + lq, row, col);
        const float inv = 1.f / l_run[qt];
        if constexpr (!F16) {
            // training: the row's log-sum-exp lets the backward pass rebuild the probabilities (train.hip)
            if (p.lse && g == 0)
                p.lse[((size_t)(b * L + ego) * P + row * W + col) * (C / 32) + head] = m_run[qt] * kl + logf(l_run[qt]);
        }
        T* o = outp + (size_t)(row * W + col) * C + head * 32 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if constexpr (F16) {
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (half_t)(o_acc[qt][dt][r] * inv);
                *reinterpret_cast<half4*>(o + dt * 16) = h;
            } else {
                *reinterpret_cast<float4*>(o + dt * 16) =
                    make_float4(o_acc[qt][dt][0] * inv, o_acc[qt][dt][1] * inv, o_acc[qt][dt][2] * inv,
                                o_acc[qt][dt][3] * inv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent producer / consumer variant (f16, window 8, 4 heads = 128 channels per workgroup): k_attention_pc.
//
// 8 wavefronts: waves 0-3 compute (one head each, exactly as k_attention), waves 4-7 only gather.  While the compute
// waves work on key chunk g out of LDS buffer g & 1, the loader waves blend chunk g + 1 (or the next item's query tile
// and identity chunk) into the other buffer; one barrier per chunk.  256 workgroups walk a tiled schedule of
// (sample, ego, window, head group) items (pc_fetch) so the pipeline never drains.  The loader is the critical role
// and is organised for memory-level parallelism and instruction count: see pc_loader_loop_general (pass-granular
// software pipeline, structured buffer loads, unconditional body), pc_loader_loop_fast (identity chunk as a copy,
// visible chunks only), pc_taps (taps of all chunks of an item in one shot), blend_h (packed-f16 4-tap sum).
// ------------------------------------------------------------------------------------------
struct PcItem {
    int b, ego, wx, wy, hg;
};

// Work schedule of the persistent kernel.  Workgroup b runs on XCD b % 8 (the usual round-robin placement; only
// speed depends on it) and each XCD has its own L2, so the gridDim / 8 workgroups of an XCD take, per step, the
// windows of ONE tile of TH x 8 adjacent windows (times the head groups): neighbouring windows gather
// overlapping rows of the source maps - in the dilated grid partition the keys of window (wx, wy + 1) are the
// right-hand neighbours of the keys of (wx, wy), i.e. three of its four bilinear taps - and with a
// window-index-strided assignment those re-reads always landed in another XCD's L2 (measured: 7.5 GB fetched
// per grid launch for 0.7 GB of K'/V').  Step k of XCD x is tile k * 8 + x of the (sample, ego, tile) list.
// Returns false when the list is exhausted; windows of a border tile that fall outside the map are skipped.
// The walk over the list is incremental: the list position s = k * 8 + x is kept decomposed into (sample, ego, tile row,
// tile column), and a step of 8 is a few scalar adds and compares.  (The first version divided s out on every call: three
// integer divisions, i.e. about 100 dependent VALU instructions - roughly 1000 cycles per item on the critical role.)
struct PcCursor {
    int b, ego, trow, tcol;   // position in the (sample, ego, tile) list
    int started;
    int wpx;                  // workgroups per XCD
};
// gridDim.x comes out of the dispatch packet with a VECTOR load: read inside pc_fetch it put an s_waitcnt vmcnt(0) - a drain of the
// loader's taps in flight - into every item (round-4 ISA reading); read once, here
__device__ __forceinline__ PcCursor pc_cursor() {
    PcCursor c = {0, 0, 0, 0, 0, __builtin_amdgcn_readfirstlane((int)(gridDim.x >> 3))};
    return c;
}
constexpr int kPcs2Grid = 256;     // k_attention_pcs2: one workgroup per CU, a compile-time constant for the item walk
__device__ __forceinline__ PcCursor pcs2_cursor() {
    PcCursor c = {0, 0, 0, 0, 0, kPcs2Grid >> 3};
    return c;
}

// Head group of a workgroup.  Default: the two head groups of a window alternate inside an XCD (an XCD step = wpx / NG windows x NG
// head groups).  variant bit 0x800 ("head group per XCD"): XCD x works on head group x % NG only, so its workgroups cover twice
// as many windows per step - in the dilated grid stages a 4 x 8 instead of a 2 x 8 window tile per lattice node, i.e. 45 instead
// of 54 gathered pixels per 32 keys - and its L2 only ever holds that half of the K' / V' rows.
__device__ __forceinline__ int pc_head_group(const AttnParams& p, int NG) {
    return (p.variant & 0x800) ? (int)(blockIdx.x & 7) % NG : (int)(blockIdx.x >> 3) % NG;
}

// World-ordered list (p.sched, launch_attn_schedule): the list is cut into segments of sched_sub x (items per XCD step); XCD x
// takes segments x, x + 8, ... and its workgroups walk a segment in sched_sub steps of consecutive items, so that what an
// XCD's workgroups gather at any moment - for ALL egos - lies under the same few hundred pixels of ground.
__device__ __forceinline__ bool pc_fetch_sched(const AttnParams& p, int X, int Y, int NG, PcCursor& cur, PcItem& it) {
    const int wpx = cur.wpx, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const bool hx = (p.variant & 0x800) != 0;
    const int tps = hx ? wpx : wpx / NG, t = hx ? j : j / NG, sub = p.sched_sub;
    const int lanes = hx ? 8 / NG : 8, lane0 = hx ? x / NG : x;          // XCDs sharing a head group, position among them
    while (true) {
        if (!cur.started) { cur.trow = lane0; cur.tcol = 0; cur.started = 1; }
        else if (++cur.tcol == sub) { cur.tcol = 0; cur.trow += lanes; }
        if (cur.trow * sub * tps >= p.n_sched) return false;
        const int pos = (cur.trow * sub + cur.tcol) * tps + t;
        if (pos >= p.n_sched) continue;
        const int* a = p.sched + __builtin_amdgcn_readfirstlane(pos);
        unsigned w;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(a) : "memory");
        const int wy = w & 1023, wx = (w >> 10) & 1023, ego = (w >> 20) & 15, s = w >> 24;
        if (p.prune) {
            const unsigned* v_ = p.vis_mask + __builtin_amdgcn_readfirstlane(((s * p.n_ego + ego) * X + wx) * Y + wy);
            unsigned v;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(v_) : "memory");
            if (v >> 31) continue;
        }
        it.b = __builtin_amdgcn_readfirstlane(s);
        it.ego = __builtin_amdgcn_readfirstlane(ego);
        it.wx = __builtin_amdgcn_readfirstlane(wx);
        it.wy = __builtin_amdgcn_readfirstlane(wy);
        it.hg = __builtin_amdgcn_readfirstlane(pc_head_group(p, NG));
        return true;
    }
}

__device__ __forceinline__ bool pc_fetch(const AttnParams& p, int X, int Y, int NG, bool ego_fastest, PcCursor& cur, PcItem& it) {
    if (p.sched) return pc_fetch_sched(p, X, Y, NG, cur, it);
    const int wpx = cur.wpx, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const bool hx = (p.variant & 0x800) != 0;          // head group per XCD (pc_head_group)
    const int TH = (hx ? wpx : wpx / NG) / 8;          // tile = TH x 8 windows
    const int ntx = (X + TH - 1) / TH, nty = (Y + 7) / 8;
    const int t = hx ? j : j / NG;
    while (true) {
        // advance by 8 list positions (by x for the first call): the fastest index first, carries into the others
        int step = cur.started ? (hx ? 8 / NG : 8) : (hx ? x / NG : x);
        cur.started = 1;
        if (ego_fastest) {
            cur.ego += step;
            while (cur.ego >= p.n_ego) { cur.ego -= p.n_ego; ++cur.tcol; }
            while (cur.tcol >= nty) { cur.tcol -= nty; ++cur.trow; }
            while (cur.trow >= ntx) { cur.trow -= ntx; ++cur.b; }
        } else {
            cur.tcol += step;
            while (cur.tcol >= nty) { cur.tcol -= nty; ++cur.trow; }
            while (cur.trow >= ntx) { cur.trow -= ntx; ++cur.ego; }
            while (cur.ego >= p.n_ego) { cur.ego -= p.n_ego; ++cur.b; }
        }
        if (cur.b >= p.B) return false;
        const int s = cur.b, ego = cur.ego;
        const int wx = cur.trow * TH + (t >> 3), wy = cur.tcol * 8 + (t & 7);
        if (wx < X && wy < Y) {
            if (p.prune) {   // items the pruned last stage cannot reach (k_window_need -> bit 31 of the visibility word)
                const unsigned* a = p.vis_mask + __builtin_amdgcn_readfirstlane(((s * p.n_ego + ego) * X + wx) * Y + wy);
                unsigned v;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
                if (v >> 31) continue;
            }
            // wave-uniform: keep the fields in SGPRs so that the per-chunk metadata (affine record, agent
            // types, plane bases) is fetched with scalar loads
            it.b = __builtin_amdgcn_readfirstlane(s);
            it.ego = __builtin_amdgcn_readfirstlane(ego);
            it.wx = __builtin_amdgcn_readfirstlane(wx);
            it.wy = __builtin_amdgcn_readfirstlane(wy);
            it.hg = __builtin_amdgcn_readfirstlane(pc_head_group(p, NG));
            return true;
        }
    }
}

// bit c = chunk c of the item has at least one visible key (k_tile_vis); all ones when the table is not in use.
// A scalar load on purpose: a vector load of this wave-uniform word would be followed by s_waitcnt vmcnt(0), i.e. it
// would drain the loader's in-flight taps once per item.
__device__ __forceinline__ unsigned pc_item_vis(const AttnParams& p, const PcItem& it, int X, int Y, bool enabled) {
    if (!enabled || !p.vis_mask) return 0xffffffffu;
    const int pos = __builtin_amdgcn_readfirstlane(((it.b * p.n_ego + it.ego) * X + it.wx) * Y + it.wy);
    const unsigned* a = p.vis_mask + pos;
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
    return v | 1u;
}

template <int HG, int CW, int LWX>
struct PcShared {
    static constexpr int CH = HG * 32;             // channels of the head group
    static constexpr int QS = CH + 8, KS = CH + 8, VS = CH + 16;
    static constexpr int TPK = HG * 4;             // loader lanes per key row (8 channels each)
    static constexpr int KPW = 64 / TPK;           // keys per loader wave and pass
    static constexpr int LWG = HG * LWX;           // loader waves
    static constexpr int CWG = HG * CW;            // compute waves (CW per head)
    static constexpr int KPP = LWG * KPW;          // keys gathered per pass by all loader waves
    static constexpr int NP = 64 / KPP;            // passes per chunk
    static constexpr int NK = NP * KPW;            // keys owned by one loader wave
    static constexpr int NQW = 4 / CW;             // 16-query tiles per compute wave
    static constexpr bool BIAS_LDS = CW > 1;       // relative-position bias fragments in LDS
    static constexpr int MAX_PAIRS = 128;      // B * L * L affine records kept in LDS
    half_t Qs[2][64 * QS];
    half_t Ks[2][64 * KS];
    half_t Vs[2][64 * VS];
    float maskadd[2][64];
    int vis[2][LWG];
    float ainv[MAX_PAIRS * 8];                 // sampling maps of every (source, ego) pair
    float bkv[HMVIT_NUM_TYPES * HMVIT_NUM_TYPES][2][CH];   // folded k / v biases of this head group
    float bq[HMVIT_NUM_TYPES][CH];
    int mode[kMaxSlots], cav[kMaxSlots], ego_e[kMaxSlots];   // copies of the kernel-argument byte arrays
    // per loader wave: bilinear taps of its NK keys for TG consecutive chunks, computed in one
    // shot (lane = (chunk, key)); a tap that needs no load (zero weight, masked key) has index -1
    static constexpr int TG = 4;
    int tidx[LWG][TG][NK][4];
    unsigned tw[LWG][TG][NK][4];                  // tap weights as packed f16 pairs (w, w)
    int tvis[LWG][TG][NK];
    float biasf[BIAS_LDS ? HG * 7 * 256 : 4];  // [head][variant][lane][4], accumulator order
};

// source agent of chunk c for ego e: the ego itself first, then the others in order
__device__ __forceinline__ int pc_src(int c, int ego) { return c == 0 ? ego : (c <= ego ? c - 1 : c); }

// debug trace: workgroup 0, one lane of one wave per role, stamps[(iter * 8 + slot)]
#ifdef HMVIT_PROBE
#define PC_TRACE(cond, iter, slot)                                                              \
    do {                                                                                        \
        if (p.trace && blockIdx.x == 0 && (cond) && (iter) < 64)                                \
            p.trace[(iter) * 8 + (slot)] = __builtin_readcyclecounter();                        \
    } while (0)
// k_attention_pcs2: workgroup 0, compute wave 0 (region 0) and loader wave 0 (region 1), 16 slots per step (tests/tools/pcs2_trace.py)
#define PC2_TRACE(region, cond, iter, slot)                                                              \
    do {                                                                                                 \
        if (p.trace && blockIdx.x == 0 && (cond) && (iter) < 64)                                         \
            p.trace[4096 + (region) * 1024 + (iter) * 16 + (slot)] = __builtin_readcyclecounter();       \
    } while (0)
#else
#define PC_TRACE(cond, iter, slot) do {} while (0)
#define PC2_TRACE(region, cond, iter, slot) do {} while (0)
#endif

// 16 bytes of a projected map through a STRUCTURED buffer descriptor (record = the C channels of
// one token): the address base + soffset + index * stride + offset is formed by the texture
// addresser, so a tap costs the loader no VALU instruction, and an index outside [0, tokens) - the
// tables use -1 - returns zeros (range check probed on gfx950 in tools/probe/sbuf_probe.hip).
// hipcc has no builtin for the idxen form; the LLVM intrinsic is bound by name.
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__device__ uint4v llvm_struct_buffer_load_b128(int4v rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.v4i32");

__device__ __forceinline__ int4v token_rsrc(const void* base, int stride_bytes, int n_tokens) {
    const unsigned long long a = (unsigned long long)base;
    int4v rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(((unsigned)(a >> 32) & 0xffffu) | ((unsigned)stride_bytes << 16)));
    rs.z = n_tokens;
    rs.w = 0x00020000;
    return rs;
}
__device__ __forceinline__ half8 tok_load8(int4v rs, int token, int off_bytes, int soff) {
    return __builtin_bit_cast(half8, llvm_struct_buffer_load_b128(rs, token, off_bytes, soff, 0));
}

// wave-uniform description of one gather = (work item, key chunk)
struct PcGather {
    PcItem it;
    int4v rs_kv, rs_q;
    int chunk, slot, kvbuf, qbuf, te, tsel;   // slot: row of the tap tables holding this chunk
    bool valid, q_ident, self_vis;
};

template <int HG, int CW, int LWX>
__device__ __forceinline__ PcGather pc_describe(const AttnParams& p, const PcShared<HG, CW, LWX>& sm, const PcItem& it, int chunk,
                                                int g, int qi, bool valid) {
    constexpr int CH = PcShared<HG, CW, LWX>::CH;
    const int L = p.L, P = p.H * p.W, C = p.C;
    PcGather G;
    G.it = it; G.chunk = chunk; G.kvbuf = g & 1; G.qbuf = qi; G.valid = valid;
    const int src = pc_src(chunk, it.ego);
    const int te = __builtin_amdgcn_readfirstlane(sm.mode[it.b * L + it.ego]);
    const int ts = __builtin_amdgcn_readfirstlane(sm.mode[it.b * L + src]);
    const int ev = __builtin_amdgcn_readfirstlane(sm.ego_e[it.b * L + it.ego]);
    G.te = te; G.tsel = te * HMVIT_NUM_TYPES + ts;
    const half_t* kpl = reinterpret_cast<const half_t*>(p.kv) + ((size_t)((it.b * L + src) * p.E + ev) * 2) * P * C + it.hg * CH;
    const half_t* qpl = reinterpret_cast<const half_t*>(p.q) + (size_t)(it.b * L + it.ego) * P * C + it.hg * CH;
    G.rs_kv = token_rsrc(kpl, C * 2, P);
    G.rs_q = token_rsrc(qpl, C * 2, P);
    G.q_ident = __builtin_amdgcn_readfirstlane(__float_as_int(sm.ainv[((it.b * L + it.ego) * L + it.ego) * 8 + 6])) != 0;
    G.self_vis = (__builtin_amdgcn_readfirstlane(sm.cav[it.b * L + it.ego]) != 0) && !(p.variant & 0x20);
    G.slot = chunk % PcShared<HG, CW, LWX>::TG;
    return G;
}

// bits of the f16 pair (w, w)
__device__ __forceinline__ unsigned pack_ww(float w) {
    const half2v h = half2v{(half_t)w, (half_t)w};
    return __builtin_bit_cast(unsigned, h);
}

// Bilinear taps of this loader wave's keys for chunks [chunk0, chunk0 + TG) of one item.
template <int HG, int CW, int LWX>
__device__ __forceinline__ void pc_taps(const AttnParams& p, PcShared<HG, CW, LWX>& sm, const PcItem& it, int chunk0, int lw, int lane) {
    using SM = PcShared<HG, CW, LWX>;
    constexpr int KPW = SM::KPW, NK = SM::NK, KPP = SM::KPP, TG = SM::TG;
    const int H = p.H, W = p.W, L = p.L, X = H / 8, Y = W / 8;
    const int dbg = p.variant;
#pragma unroll
    for (int base = 0; base < TG * NK; base += 64) {
        const int e = base + lane;
        const int c = e / NK, j = e % NK;
        const int chunk = chunk0 + c;
        if (e < TG * NK && chunk < p.n_src) {
            const int src = pc_src(chunk, it.ego);
            const float* a = sm.ainv + ((it.b * L + src) * L + it.ego) * 8;
            const int n = (j / KPW) * KPP + KPW * lw + (j % KPW);
            int row, col;
            token_pixel(p.partition, 8, X, Y, it.wx, it.wy, n, row, col);
            const bool cav = (sm.cav[it.b * L + src] != 0) && !(dbg & 0x20);
            int ix[4];
            float w[4];
            bool vis;
            if ((a[6] != 0.f) || (dbg & 0x10)) {   // identity map: the key's own pixel
                ix[0] = row * W + col; ix[1] = ix[2] = ix[3] = -1;
                w[0] = 1.f; w[1] = w[2] = w[3] = 0.f;
                vis = cav;
            } else {
                const Taps t = make_taps(a, col, row, H, W);
                vis = cav && t.roi != 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ix[k] = (t.w[k] != 0.f) ? t.idx[k] : -1;
                    w[k] = t.w[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!vis) ix[k] = -1;
                else if ((dbg & 0x400) && ix[k] >= 0) ix[k] &= 4095;   // probe: L2-resident footprint
            }
            *reinterpret_cast<int4*>(sm.tidx[lw][c][j]) = make_int4(ix[0], ix[1], ix[2], ix[3]);
            *reinterpret_cast<uint4v*>(sm.tw[lw][c][j]) = uint4v{pack_ww(w[0]), pack_ww(w[1]), pack_ww(w[2]), pack_ww(w[3])};
            sm.tvis[lw][c][j] = vis ? 1 : 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS write -> read ordering
}

// bias + sum_k w[k] * tap_k over NT taps for the 8 channels of a lane, in packed f16 (v_pk_fma_f16: two
// channels per instruction, half the VALU time of f32 FMAs on converted inputs).  Each of the NT roundings
// is half an f16 ulp of a value that is stored as f16 anyway; against the oracle the end-to-end error of the
// f16 mode moved from 4.5e-4 to at most 4.8e-4 (tools/err_report.py).  NT = 1 with w = 1 is the exact x + b.
template <int NT>
__device__ __forceinline__ half8 blend_h(const half8 bias, const half8 (&taps)[4], const uint4v w2) {
    half8 o;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        half2v acc = half2v{bias[2 * e2], bias[2 * e2 + 1]};
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const half2v x = half2v{taps[k][2 * e2], taps[k][2 * e2 + 1]};
            const unsigned wk = w2[k];   // (bit_cast straight from the vector element is miscompiled by hipcc 7.2: it always reads element 0)
            acc = __builtin_elementwise_fma(__builtin_bit_cast(half2v, wk), x, acc);
        }
        o[2 * e2] = acc[0]; o[2 * e2 + 1] = acc[1];
    }
    return o;
}

// query row + folded bias, summed in f32 (|b_q| can be an order of magnitude above |q|: rounding it to f16
// first would shift every logit of the head)
__device__ __forceinline__ half8 add_bias_f32(const half8 x, const float* bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias);
    const float4 b1 = *reinterpret_cast<const float4*>(bias + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] + bb[e]);
    return o;
}

__device__ __forceinline__ half8 to_half8(const float* v) {
    const float4 b0 = *reinterpret_cast<const float4*>(v);
    const float4 b1 = *reinterpret_cast<const float4*>(v + 4);
    return half8{(half_t)b0.x, (half_t)b0.y, (half_t)b0.z, (half_t)b0.w, (half_t)b1.x, (half_t)b1.y, (half_t)b1.z, (half_t)b1.w};
}

__device__ __forceinline__ void pc_wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// per-workgroup tables: affine records, and the biases of the head group this workgroup serves
// (the item stride gridDim.x is a multiple of the number of head groups, so hg never changes)
template <int HG, int CW, int LWX>
__device__ __forceinline__ void pc_load_tables(const AttnParams& p, PcShared<HG, CW, LWX>& sm, int hg) {
    constexpr int CH = PcShared<HG, CW, LWX>::CH;
    const int n_rec = p.B * p.L * p.L * 8;
    for (int i = threadIdx.x; i < n_rec; i += blockDim.x) sm.ainv[i] = p.ainv[i];
    for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * HMVIT_NUM_TYPES * 2 * CH; i += blockDim.x) {
        const int e = i / (2 * CH), pl = (i / CH) & 1, c = i % CH;
        sm.bkv[e][pl][c] = p.b_kv[(size_t)e * 2 * p.C + pl * p.C + hg * CH + c];
    }
    for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * CH; i += blockDim.x)
        sm.bq[i / CH][i % CH] = p.b_q[(i / CH) * p.C + hg * CH + (i % CH)];
    if (threadIdx.x < kMaxSlots) {
        sm.mode[threadIdx.x] = p.mode[threadIdx.x];
        sm.cav[threadIdx.x] = p.cav[threadIdx.x];
        sm.ego_e[threadIdx.x] = p.ego_e[threadIdx.x];
    }
    if constexpr (PcShared<HG, CW, LWX>::BIAS_LDS) {
        const float* src = p.bias_frag + (size_t)hg * HG * 7 * 256;   // heads hg*HG .. hg*HG+HG-1 are contiguous
        for (int i = threadIdx.x; i < HG * 7 * 256; i += blockDim.x) sm.biasf[i] = src[i];
    }
}

// Loader role.  Both role loops execute 1 + (number of (item, chunk) gathers) barriers; gather g
// fills K/V buffer g & 1 and is consumed by the compute waves in the barrier interval after the one
// it was stored in.
//
// The loop is software-pipelined at the granularity of a PASS (KPW keys per wave = 8 tap loads of
// 16 B per lane): the loads of a pass of gather g + 1 are issued right after the same pass of gather g
// has been blended out of its registers, so a full chunk of tap loads (NP passes) is always in flight
// across the blend, the barrier and the tap arithmetic - the loader never sits in an empty-queue wait.
// Everything inside the loop body is unconditional (masked keys / zero-weight taps / "no query tile
// for this chunk" are index -1 = zero-returning out-of-range records), so the compiler's vmcnt
// bookkeeping stays exact across the back edge.  The loader's barrier is the raw s_barrier preceded by
// an LDS-only wait: __syncthreads() would drain the in-flight taps (vmcnt(0)) every chunk.
template <int HG, int CW, int LWX>
__device__ __forceinline__ void pc_loader_loop_general(const AttnParams& p, PcShared<HG, CW, LWX>& sm, int lw, int ltid) {
    using SM = PcShared<HG, CW, LWX>;
    constexpr int QS = SM::QS, KS = SM::KS, VS = SM::VS, CH = SM::CH, TPK = SM::TPK, KPW = SM::KPW, NP = SM::NP, KPP = SM::KPP, TG = SM::TG;
    const int X = p.H / 8, Y = p.W / 8, NG = p.C / (HG * 32);
    const int n_src = p.n_src;
    const int plane_bytes = p.H * p.W * p.C * 2;
    const int lane = ltid & 63;
    const int cl = (ltid % TPK) * 8, cl_bytes = cl * 2;
    const int kin = ltid / TPK;                  // key row of this lane inside a KPP-key pass
    const int kj = kin % KPW;                    // ... inside this wave's share of the pass
    const bool ego_fastest = (p.variant & 0x200) == 0;

    half8 R[NP][2][4], RQ[NP];
    uint4v Wt[NP];
    int vflag[NP];

    auto issue = [&](int pass, const PcGather& G) {
        const int c = G.slot, j = pass * KPW + kj;
        int4 ix = *reinterpret_cast<const int4*>(sm.tidx[lw][c][j]);
        Wt[pass] = *reinterpret_cast<const uint4v*>(sm.tw[lw][c][j]);
        vflag[pass] = sm.tvis[lw][c][j];
        int qtok = -1;
        if (G.chunk == 0 && G.q_ident) {
            int row, col;
            token_pixel(p.partition, 8, X, Y, G.it.wx, G.it.wy, pass * KPP + kin, row, col);
            qtok = row * p.W + col;
        }
        if (!G.valid) { ix = make_int4(-1, -1, -1, -1); qtok = -1; }
        const int ixa[4] = {ix.x, ix.y, ix.z, ix.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R[pass][0][k] = tok_load8(G.rs_kv, ixa[k], cl_bytes, 0);
            R[pass][1][k] = tok_load8(G.rs_kv, ixa[k], cl_bytes, plane_bytes);
        }
        RQ[pass] = tok_load8(G.rs_q, qtok, cl_bytes, 0);
    };

    bool any = false, allv = true;
    auto blend = [&](int pass, const PcGather& G) {
        const int kk = pass * KPP + kin;
        *reinterpret_cast<half8*>(sm.Ks[G.kvbuf] + kk * KS + cl) =
            blend_h<4>(to_half8(&sm.bkv[G.tsel][0][cl]), R[pass][0], Wt[pass]);
        *reinterpret_cast<half8*>(sm.Vs[G.kvbuf] + kk * VS + cl) =
            blend_h<4>(to_half8(&sm.bkv[G.tsel][1][cl]), R[pass][1], Wt[pass]);
        if (G.chunk == 0) {
            if (G.q_ident) {
                *reinterpret_cast<half8*>(sm.Qs[G.qbuf] + kk * QS + cl) = add_bias_f32(RQ[pass], &sm.bq[G.te][cl]);
            } else {
                // T[i,i] is not the identity (never produced by the reference's dataset): slow path
                const float* aq = sm.ainv + ((G.it.b * p.L + G.it.ego) * p.L + G.it.ego) * 8;
                const half_t* qpl = reinterpret_cast<const half_t*>(p.q) + (size_t)(G.it.b * p.L + G.it.ego) * p.H * p.W * p.C + G.it.hg * CH;
                int row, col;
                token_pixel(p.partition, 8, X, Y, G.it.wx, G.it.wy, kk, row, col);
                const Taps tt = make_taps(aq, col, row, p.H, p.W);
                float v[1][8];
                sample8<half_t, 1>(qpl, 0, p.C, cl, tt, false, row * p.W + col, sm.bq[G.te], 0, v);
                store8_lds<half_t>(sm.Qs[G.qbuf] + kk * QS + cl, v[0]);
            }
        }
        const bool vis = vflag[pass] != 0;
        if ((ltid % TPK) == 0) sm.maskadd[G.kvbuf][kk] = vis ? 0.f : -INFINITY;
        any |= vis;
        allv &= vis;
    };

    PcCursor item = pc_cursor();
    int chunk = 0, g = 0, qi = 0;
    PcItem it;
    if (!pc_fetch(p, X, Y, NG, ego_fastest, item, it)) { pc_wg_barrier(); return; }
    PcGather G = pc_describe<HG, CW, LWX>(p, sm, it, 0, 0, 0, true);
    pc_taps<HG, CW, LWX>(p, sm, it, 0, lw, lane);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) issue(pass, G);

#pragma unroll 1
    while (true) {
        // the gather after this one
        bool nvalid = true;
        if (++chunk == n_src) {
            chunk = 0;
            nvalid = pc_fetch(p, X, Y, NG, ego_fastest, item, it);
            qi ^= 1;
        }
        const PcGather N = pc_describe<HG, CW, LWX>(p, sm, it, chunk, g + 1, qi, nvalid);
        if (nvalid && (chunk % TG) == 0) pc_taps<HG, CW, LWX>(p, sm, it, chunk, lw, lane);
        any = false; allv = true;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            __builtin_amdgcn_sched_barrier(0);
            blend(pass, G);
            __builtin_amdgcn_sched_barrier(0);
            issue(pass, N);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool wave_any = __any(any), wave_all = __all(allv);
        if (lane == 0) sm.vis[G.kvbuf][lw] = (wave_any ? 1 : 0) | (wave_all ? 2 : 0);   // bit 0: some key visible, bit 1: all
        pc_wg_barrier();
        if (!nvalid) break;
        G = N;
        ++g;
    }
    pc_wg_barrier();   // the interval in which the compute waves consume the last chunk
}

// Loader role, fast variant: every agent's self transform T[i,i] is the identity (always, for the
// reference's datasets).  Chunk 0 of an item - the ego's own map - is then a plain copy: one load each
// for the K', V' and query rows of a key instead of eight tap loads plus a query load, and the query
// tile never needs a load slot in the other chunks.  The texture addresser, not HBM, bounds the loader
// (about 22 cycles per 1 KB wave load, hits and out-of-range records included: tools/probe/
// sbuf_probe.hip), so 140 instead of 180 loads per lane and item is a direct win.  The loop is the same
// pass-granular software pipeline as the general variant, unrolled over the chunk sequence
// I (identity) -> G (general) ... G -> I of the next item so that each body stays straight-line code.
template <int HG, int CW, int LWX>
__device__ __forceinline__ void pc_loader_loop_fast(const AttnParams& p, PcShared<HG, CW, LWX>& sm, int lw, int ltid) {
    using SM = PcShared<HG, CW, LWX>;
    constexpr int QS = SM::QS, KS = SM::KS, VS = SM::VS, CH = SM::CH, TPK = SM::TPK, KPW = SM::KPW, NP = SM::NP, KPP = SM::KPP, TG = SM::TG;
    const int X = p.H / 8, Y = p.W / 8, NG = p.C / (HG * 32);
    const int n_src = p.n_src;
    const int plane_bytes = p.H * p.W * p.C * 2;
    const int lane = ltid & 63;
    const int cl = (ltid % TPK) * 8, cl_bytes = cl * 2;
    const int kin = ltid / TPK;                  // key row of this lane inside a KPP-key pass
    const int kj = kin % KPW;                    // ... inside this wave's share of the pass
    const bool ego_fastest = (p.variant & 0x200) == 0;

    half8 R[NP][2][4];
    uint4v Wt[NP];
    int vflag[NP];
    bool any = false, allv = true;

    auto issueG = [&](int pass, const PcGather& G) {
        const int j = pass * KPW + kj;
        const int4 ix = *reinterpret_cast<const int4*>(sm.tidx[lw][G.slot][j]);
        Wt[pass] = *reinterpret_cast<const uint4v*>(sm.tw[lw][G.slot][j]);
        vflag[pass] = sm.tvis[lw][G.slot][j];
        const int ixa[4] = {ix.x, ix.y, ix.z, ix.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R[pass][0][k] = tok_load8(G.rs_kv, ixa[k], cl_bytes, 0);
            R[pass][1][k] = tok_load8(G.rs_kv, ixa[k], cl_bytes, plane_bytes);
        }
    };
    auto issueI = [&](int pass, const PcGather& G) {
        int row, col;
        token_pixel(p.partition, 8, X, Y, G.it.wx, G.it.wy, pass * KPP + kin, row, col);
        const int tok = row * p.W + col;
        const int tq = G.valid ? tok : -1;
        const int tk = (G.valid && G.self_vis) ? tok : -1;
        R[pass][0][0] = tok_load8(G.rs_kv, tk, cl_bytes, 0);
        R[pass][1][0] = tok_load8(G.rs_kv, tk, cl_bytes, plane_bytes);
        R[pass][0][1] = tok_load8(G.rs_q, tq, cl_bytes, 0);
    };
    half8 bias_k, bias_v;   // folded biases of the chunk being blended
    auto load_bias = [&](const PcGather& G) {
        bias_k = to_half8(&sm.bkv[G.tsel][0][cl]);
        bias_v = to_half8(&sm.bkv[G.tsel][1][cl]);
    };
    auto blendG = [&](int pass, const PcGather& G) {
        const int kk = pass * KPP + kin;
        *reinterpret_cast<half8*>(sm.Ks[G.kvbuf] + kk * KS + cl) = blend_h<4>(bias_k, R[pass][0], Wt[pass]);
        *reinterpret_cast<half8*>(sm.Vs[G.kvbuf] + kk * VS + cl) = blend_h<4>(bias_v, R[pass][1], Wt[pass]);
        const bool vis = vflag[pass] != 0;
        if ((ltid % TPK) == 0) sm.maskadd[G.kvbuf][kk] = vis ? 0.f : -INFINITY;
        any |= vis;
        allv &= vis;
    };
    auto blendI = [&](int pass, const PcGather& G) {
        const int kk = pass * KPP + kin;
        *reinterpret_cast<half8*>(sm.Ks[G.kvbuf] + kk * KS + cl) = add_bias_f32(R[pass][0][0], &sm.bkv[G.tsel][0][cl]);
        *reinterpret_cast<half8*>(sm.Vs[G.kvbuf] + kk * VS + cl) = add_bias_f32(R[pass][1][0], &sm.bkv[G.tsel][1][cl]);
        *reinterpret_cast<half8*>(sm.Qs[G.qbuf] + kk * QS + cl) = add_bias_f32(R[pass][0][1], &sm.bq[G.te][cl]);
        if ((ltid % TPK) == 0) sm.maskadd[G.kvbuf][kk] = G.self_vis ? 0.f : -INFINITY;
    };
    auto publish = [&](const PcGather& G, bool some, bool every) {
        __builtin_amdgcn_sched_barrier(0);
        const bool wave_any = __any(some), wave_all = __all(every);
        if (lane == 0) sm.vis[G.kvbuf][lw] = (wave_any ? 1 : 0) | (wave_all ? 2 : 0);   // bit 0: some key visible, bit 1: all
        pc_wg_barrier();
    };

    PcCursor item = pc_cursor();
    int g = 0, qi = 0;
    PcItem it;
    if (!pc_fetch(p, X, Y, NG, ego_fastest, item, it)) { pc_wg_barrier(); return; }
    unsigned vis = pc_item_vis(p, it, X, Y, true);   // chunks of the item with at least one visible key (bit 0 = the ego)
    PcGather G = pc_describe<HG, CW, LWX>(p, sm, it, 0, 0, 0, true);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) issueI(pass, G);

    // Per item: identity chunk, then the VISIBLE general chunks (chunks without a visible key are skipped by both
    // roles - no loads, no blend, no barrier), then on to the next item.  The control flow keeps the shape
    // I -> G, loop G -> G, G -> I (or I -> I) with straight-line bodies, which is what lets hipcc keep the 128 tap
    // registers in place and its vmcnt bookkeeping exact.
    const unsigned chunk_bits = (1u << n_src) - 1u;
#pragma unroll 1
    while (true) {
        bool nvalid;
        PcGather N;
        unsigned rest = vis & ~1u & chunk_bits;
        const bool has_general = rest != 0;
        if (has_general) {
            // identity chunk while the first visible general chunk is requested
            int c = __builtin_ctz(rest);
            rest &= rest - 1;
            int tap_group = (c - 1) / TG;       // group of TG chunks whose taps are in the LDS tables
            pc_taps<HG, CW, LWX>(p, sm, it, 1 + tap_group * TG, lw, lane);
            N = pc_describe<HG, CW, LWX>(p, sm, it, c, g + 1, qi, true);
            N.slot = (c - 1) % TG;
            load_bias(G);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                __builtin_amdgcn_sched_barrier(0);
                blendI(pass, G);
                __builtin_amdgcn_sched_barrier(0);
                issueG(pass, N);
            }
            publish(G, G.self_vis, G.self_vis);
            G = N; ++g;
            // general chunks while the following visible one is requested
#pragma unroll 1
            while (rest) {
                c = __builtin_ctz(rest);
                rest &= rest - 1;
                if ((c - 1) / TG != tap_group) {
                    tap_group = (c - 1) / TG;
                    pc_taps<HG, CW, LWX>(p, sm, it, 1 + tap_group * TG, lw, lane);
                }
                N = pc_describe<HG, CW, LWX>(p, sm, it, c, g + 1, qi, true);
                N.slot = (c - 1) % TG;
                any = false; allv = true;
                load_bias(G);
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    __builtin_amdgcn_sched_barrier(0);
                    blendG(pass, G);
                    __builtin_amdgcn_sched_barrier(0);
                    issueG(pass, N);
                }
                publish(G, any, allv);
                G = N; ++g;
            }
        }
        // last chunk of the item while chunk 0 of the next item is requested
        nvalid = pc_fetch(p, X, Y, NG, ego_fastest, item, it);
        if (nvalid) vis = pc_item_vis(p, it, X, Y, true);
        qi ^= 1;
        N = pc_describe<HG, CW, LWX>(p, sm, it, 0, g + 1, qi, nvalid);
        load_bias(G);
        if (has_general) {
            any = false; allv = true;
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                __builtin_amdgcn_sched_barrier(0);
                blendG(pass, G);
                __builtin_amdgcn_sched_barrier(0);
                issueI(pass, N);
            }
            publish(G, any, allv);
        } else {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                __builtin_amdgcn_sched_barrier(0);
                blendI(pass, G);
                __builtin_amdgcn_sched_barrier(0);
                issueI(pass, N);
            }
            publish(G, G.self_vis, G.self_vis);
        }
        if (!nvalid) break;
        G = N; ++g;
    }
    pc_wg_barrier();   // the interval in which the compute waves consume the last chunk
}

template <int HG, int CW, int LWX>
__device__ __forceinline__ void pc_compute_loop(const AttnParams& p, PcShared<HG, CW, LWX>& sm, int wave, int lane, bool use_vis) {
    using SM = PcShared<HG, CW, LWX>;
    constexpr int QS = SM::QS, KS = SM::KS, VS = SM::VS, NQW = SM::NQW, LWG = SM::LWG;
    const int hl = wave / CW, qbase = (wave % CW) * NQW;   // head inside the group, first query tile
    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / 8, Y = W / 8, NG = C / (HG * 32);
    const int n_src = p.n_src;
    const int lq = lane & 15, g = lane >> 4;
    PcCursor item = pc_cursor();
    PcItem it;
    __syncthreads();
    if (!pc_fetch(p, X, Y, NG, (p.variant & 0x200) == 0, item, it)) return;

    float4v biasf[7];
    int bias_head = -1;
    half8 qh[NQW];
    float m_run[NQW];
    float4v o_acc[NQW][2], l_acc[NQW];
    const half8 ones = (half8)(half_t)1.0f;
    int gstep = 0, qi = 0;
    while (true) {
        const int head = it.hg * HG + hl;
        // chunks without a visible key are skipped by the loader and here alike (pc_item_vis)
        const unsigned vis = pc_item_vis(p, it, X, Y, use_vis) & ((1u << n_src) - 1u);
        const int c_last = 31 - __builtin_clz(vis);
        for (int c = 0; c < n_src; ++c) {
            if (!((vis >> c) & 1u)) continue;
            const int buf = gstep & 1;
            if (c == 0) {
                if constexpr (!SM::BIAS_LDS) {
                    if (head != bias_head) {
#pragma unroll
                        for (int v = 0; v < 7; ++v)
                            biasf[v] = *reinterpret_cast<const float4v*>(p.bias_frag + ((size_t)(head * 7 + v) * 64 + lane) * 4);
                        bias_head = head;
                    }
                }
#pragma unroll
                for (int qt = 0; qt < NQW; ++qt) {
                    qh[qt] = *reinterpret_cast<const half8*>(sm.Qs[qi] + ((qbase + qt) * 16 + lq) * QS + hl * 32 + g * 8);
                    m_run[qt] = -INFINITY;
                    l_acc[qt] = (float4v)(0.f);
                    o_acc[qt][0] = (float4v)(0.f);
                    o_acc[qt][1] = (float4v)(0.f);
                }
            }
            int vis_or = 0, vis_and = 3;
#pragma unroll
            for (int w = 0; w < LWG; ++w) {
                vis_or |= sm.vis[buf][w];
                vis_and &= sm.vis[buf][w];
            }
            const bool any_visible = (vis_or & 1) != 0;
            const bool all_visible = (vis_and & 2) != 0;     // no key of the chunk is masked: skip the mask add
            if ((any_visible || !p.skip_masked) && !(p.variant & 0x40)) {
                const half_t* Kb = sm.Ks[buf];
                const half_t* Vb = sm.Vs[buf];
                float4v madd[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) madd[kt] = (float4v)(0.f);
                if (!all_visible) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) madd[kt] = *reinterpret_cast<const float4v*>(sm.maskadd[buf] + kt * 16 + 4 * g);
                }
                half8 kh[4], vh[2][2];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    kh[kt] = *reinterpret_cast<const half8*>(Kb + (kt * 16 + lq) * KS + hl * 32 + g * 8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        // V^T tile dt takes the head's channels 8 q4 + 4 dt + r (q4 = lq & 3) as its rows 4 q4 + r, so
                        // that an accumulator lane ends up with 8 consecutive channels over the two tiles
                        const half_t* base = Vb + (ks * 32 + 4 * g + (lq >> 2)) * VS + hl * 32 + (lq & 3) * 8 + dt * 4;
                        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_t*)(base));
                        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                        half8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (half_t)lo[e];
                            v[4 + e] = (half_t)hi[e];
                        }
                        vh[dt][ks] = v;
                    }
#pragma unroll
                for (int qt = 0; qt < NQW; ++qt) {
                    float4v s[4];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
                        float4v bias_c;
                        if constexpr (SM::BIAS_LDS)
                            bias_c = *reinterpret_cast<const float4v*>(sm.biasf + ((hl * 7 + (qbase + qt - kt + 3)) * 64 + lane) * 4);
                        else
                            bias_c = biasf[qt - kt + 3];
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[kt], qh[qt], bias_c, 0, 0, 0);
                    }
                    if (!all_visible) {
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt) s[kt] += madd[kt];
                    }
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
                    mx = max_over_lane_groups(mx);
                    const float m_new = max_raw(m_run[qt], mx);     // both operands come out of v_max: no canonicalising pair
                    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                    // logits are in log2 units (log2 e folded into W_q and the bias fragments on the host)
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_safe);
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m_safe);
                    m_run[qt] = m_new;
                    o_acc[qt][0] *= alpha;
                    o_acc[qt][1] *= alpha;
                    l_acc[qt] *= alpha;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 ph;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ph[e] = (half_t)s[2 * ks][e];
                            ph[4 + e] = (half_t)s[2 * ks + 1][e];
                        }
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt][ks], ph, o_acc[qt][dt], 0, 0, 0);
                        // softmax denominator on the matrix core: an all-ones "V^T" tile makes every
                        // accumulator row the sum over keys of (the f16-rounded) P for this lane's query
                        l_acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, l_acc[qt], 0, 0, 0);
                    }
                }
            }
            PC_TRACE(wave == 0 && lane == 0, gstep, 5);
            if (c == c_last) {
                half_t* outp = reinterpret_cast<half_t*>(p.out) + (size_t)(it.b * L + it.ego) * P * C;
#pragma unroll
                for (int qt = 0; qt < NQW; ++qt) {
                    int row, col;
                    token_pixel(p.partition, 8, X, Y, it.wx, it.wy, (qbase + qt) * 16 + lq, row, col);
                    const float inv = __builtin_amdgcn_rcpf(l_acc[qt][0]);     // 1 ulp; the quotient is rounded to f16 next
                    half_t* o = outp + (size_t)(row * W + col) * C + head * 32 + 8 * g;
                    half8 h;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[4 * dt + r] = (half_t)(o_acc[qt][dt][r] * inv);
                    *reinterpret_cast<half8*>(o) = h;   // one 16-byte store per query row and lane
                }
            }
            PC_TRACE(wave == 0 && lane == 0, gstep, 6);
            __syncthreads();
            PC_TRACE(wave == 0 && lane == 0, gstep, 7);
            ++gstep;
        }
        if (!pc_fetch(p, X, Y, NG, (p.variant & 0x200) == 0, item, it)) break;
        qi ^= 1;
    }
}

template <int HG, int CW, int LWX>
__global__ __launch_bounds__((HG * CW + HG * LWX) * 64) void k_attention_pc(AttnParams p) {
    using SM = PcShared<HG, CW, LWX>;
    __shared__ __attribute__((aligned(16))) SM sm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    pc_load_tables<HG, CW, LWX>(p, sm, pc_head_group(p, p.C / (HG * 32)));   // head group of this workgroup (pc_fetch)
    __syncthreads();
    // wave-uniform role split at the outermost level: the two loops have disjoint live ranges, so
    // the kernel's register count is the maximum of the two roles, not their sum
    // (workgroup-uniform) are all self transforms the identity?  Selects the loader loop; the loop that handles
    // general self transforms walks every chunk, so the visibility table is only honoured with the fast one
    bool self_ident = true;
    for (int s = threadIdx.x & 63; s < p.B * p.L; s += 64) self_ident &= sm.ainv[(s * p.L + s % p.L) * 8 + 6] != 0.f;
    const bool fast = __all(self_ident) && !(p.variant & 0x1000);
    if (wave >= SM::CWG) {
        // loader waves outrank the compute waves on their SIMD: the sooner the gather's loads are
        // issued, the more of the memory round trip overlaps with the compute waves' MFMA / softmax
        if (!(p.variant & 0x800)) __builtin_amdgcn_s_setprio(3);
        if (fast)
            pc_loader_loop_fast<HG, CW, LWX>(p, sm, wave - SM::CWG, threadIdx.x - SM::CWG * 64);
        else
            pc_loader_loop_general<HG, CW, LWX>(p, sm, wave - SM::CWG, threadIdx.x - SM::CWG * 64);
    } else {
        // probe 0x8000 (with 0x800): the compute waves outrank the loader instead.  Faster on the dense probe scene
        // (tests/tools/attn_probe.py: -6 % local, -11 % grid), 9 % slower in the fused forward, where every chunk that is
        // walked has visible keys and the loader's issue slots are the scarcer ones.
        if (p.variant & 0x8000) __builtin_amdgcn_s_setprio(3);
        pc_compute_loop<HG, CW, LWX>(p, sm, wave, threadIdx.x & 63, fast);
    }
}

template <int HG, int CW, int LWX>
static int launch_attn_pc(const AttnParams& p, hipStream_t st, int wg_per_cu) {
    const int grid = 256 * wg_per_cu;     // persistent workgroups: 8 XCDs x (a tile of windows x head groups), see pc_fetch
    hipLaunchKernelGGL((k_attention_pc<HG, CW, LWX>), dim3(grid), dim3((HG * CW + HG * LWX) * 64), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// Split-precision persistent kernel (HMVIT_PREC_SPLIT).  The round-2/3 form (k_attention_pcs: Q through the loader and LDS, tap
// tables per step) was superseded by k_attention_pcs2 below and is no longer in the source (git history: round 4).  f32 planes,
// fp32-class products on the f16 matrix pipe: every MFMA operand is a (hi, lo) pair of f16 halves (x = hi + lo, |lo| <= 2^-11 |hi|)
// and every product is three MFMAs (lo x hi, hi x lo, hi x hi; f32 accumulate); a gather is a HALF chunk of 32 keys, the visibility
// word carries a bit per half chunk (k_tile_vis); logits in natural units (exp(x) = exp2(x log2 e)); O is stored as f32.
// Requires identity self transforms (pairwise_t[b, i, i] = I, HmvitFusionDesc::self_identity).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4v tok_load4f(int4v rs, int token, int off_bytes, int soff) {
    return llvm_struct_buffer_load_b128(rs, token, off_bytes, soff, 0);
}
__device__ unsigned llvm_struct_buffer_load_b32(int4v rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.i32");
__device__ __forceinline__ unsigned tok_load1(int4v rs, int token, int off_bytes, int soff) {
    return llvm_struct_buffer_load_b32(rs, token, off_bytes, soff, 0);
}


// 4 f32 values -> hi / lo halves, stored to two LDS rows
__device__ __forceinline__ void store_split4(half_t* dh, half_t* dl, const float (&v)[4]) {
    half4 h, l;
    split_pk4(v[0], v[1], v[2], v[3], h, l);          // 6 instructions for the four values (common.hpp)
    *reinterpret_cast<half4*>(dh) = h;
    *reinterpret_cast<half4*>(dl) = l;
}


// ------------------------------------------------------------------------------------------
// k_attention_pcs2 (round 4): the same persistent producer / consumer kernel with the LOADER role rebuilt around what the
// round-4 role ablations measured (tools/probe/run_libs.sh; four attention launches of a cfg2 forward: 6.11 ms as shipped,
// 5.09 ms with the loader's instruction stream alone - no tap memory traffic, no compute role -, 3.48 ms with the compute
// role alone, 1.32 ms with both roles reduced to their bookkeeping): the kernel was bound by the loader wave's INSTRUCTION
// stream, not by memory - per 32-key step ~20 exposed LDS round trips (tap weights, biases, visibility, indices, each read
// right where it was used behind a scheduling fence), 32 v_mov_b64 + a near-full vmcnt wait at the loop head (the identity /
// general / last-gather variants of the loop body kept the 128 tap registers from being updated in place), a divergent
// mask store per pass, and per-step descriptor look-ups in LDS.  Here:
//   * ONE loop body for every gather.  The ego's own half chunks are gathers like any other whose table entries say
//     "tap 0 = the key's own pixel, weight 1, taps 1-3 out of range" (index -1 returns zeros without a memory access, and
//     1 x + 0 0 + 0 0 + 0 0 + b is exact); the end of the list is a gather through an empty descriptor.  The 128 tap
//     registers are loop-carried in place: vmcnt(24) before each pass, nothing else;
//   * Q does not pass through the loader or LDS any more: a compute wave reads its head's 64 x 32 f32 query block straight
//     from the plane (8 loads per lane, requested one item ahead), adds the bias and splits it in registers.  70 KB of LDS
//     and a fifth of the loader's tile stores go away;
//   * per ITEM, not per step: the tap tables of all of the item's half chunks (indices, weights, -inf masks, any / all flags),
//     built one item ahead by the 64 lanes of each loader wave for the wave's own 8 keys per half chunk (two table sets by
//     item parity; only the owning wave ever touches its rows, so program order is the only synchronisation); agent types,
//     K' / V' variant and validity of the item's sources are packed into scalars once;
//   * per STEP: one batch of LDS reads (4 x weights of the gather being blended, 4 x indices of the gather being requested,
//     2 x bias, 1 x mask row), ONE wait, then 4 x [blend + split + store pass p, request pass p of the next gather]; the
//     mask row and the flags reach the consumer's buffer through one 9-lane store.
// The compute role is k_attention_pcs's, minus the Q tile.
// ------------------------------------------------------------------------------------------
struct PcShared2 {
    static constexpr int HG = 4, CH = 128, KEYS = 32;
    static constexpr int KS = CH + 8, VS = CH + 16;   // halves per LDS row
    static constexpr int TPK = CH / 4;             // loader lanes per key row (4 f32 channels each)
    static constexpr int KPW = 64 / TPK;           // keys per loader wave and pass (2)
    static constexpr int LWG = 4, CWG = 4;
    static constexpr int KPP = LWG * KPW;          // keys per pass (8)
    static constexpr int NP = KEYS / KPP;          // passes per half chunk (4)
    static constexpr int NK = NP * KPW;            // keys owned by one loader wave (8)
    static constexpr int NS = 16;                  // half-chunk slots per item: 2 x n_src, n_src <= 8
    static constexpr int MAX_PAIRS = 128;
    half_t Kh[2][KEYS * KS], Kl[2][KEYS * KS];
    half_t Vh[2][KEYS * VS], Vl[2][KEYS * VS];
    float maskadd[2][KEYS];
    int vis[2][LWG];
    float ainv[MAX_PAIRS * 8];
    float bkv[HMVIT_NUM_TYPES * HMVIT_NUM_TYPES][2][CH];
    float bq[HMVIT_NUM_TYPES][CH];
    int mode[kMaxSlots], cav[kMaxSlots], ego_e[kMaxSlots];
    // item tables, [item parity][loader wave][slot = 2 chunk + half][key of the wave]
    int tidx[2][LWG][NS][NK][4];
    float tw[2][LWG][NS][NK][4];
    float tmask[2][LWG][NS][16];                   // [0..7]: 0 / -inf per key, [8]: bit 0 some key visible, bit 1 all (int bits)
    float qstage[CWG][8 * 64 * 4];                 // per compute wave: the next item's raw query block, landed by LDS-DMA
    int iconst[kMaxSlots][2];                      // per (sample, ego): PcItemC, filled once per workgroup
    unsigned items[4];                             // dynamic assignment: item words handed from the loader role to the compute role
};

// wave-uniform constants of an item: te | ev << 4 | self_vis << 8, and the (te, ts) pair of every chunk's source (4 bits each)
struct PcItemC {
    int tev;
    unsigned tsel;
};
__device__ __forceinline__ void pcs2_fill_consts(const AttnParams& p, PcShared2& sm) {    // after mode / cav / ego_e are in LDS
    const int L = p.L, i = threadIdx.x;
    if (i < p.B * L && i < kMaxSlots) {
        const int b = i / L, ego = i - b * L;
        const int te = sm.mode[i], ev = sm.ego_e[i], sv = sm.cav[i];
        unsigned ts = 0;
        for (int c = 0; c < p.n_src && c < 8; ++c) ts |= (unsigned)(te * HMVIT_NUM_TYPES + sm.mode[b * L + pc_src(c, ego)]) << (4 * c);
        sm.iconst[i][0] = te | (ev << 4) | ((sv != 0 ? 1 : 0) << 8);
        sm.iconst[i][1] = (int)ts;
    }
}
__device__ __forceinline__ PcItemC pcs2_item_consts(const AttnParams& p, const PcShared2& sm, const PcItem& it) {
    const int2 v = *reinterpret_cast<const int2*>(sm.iconst[it.b * p.L + it.ego]);
    PcItemC r;
    r.tev = __builtin_amdgcn_readfirstlane(v.x);
    r.tsel = __builtin_amdgcn_readfirstlane(v.y);
    return r;
}
// half chunks to walk: both halves of the ego's own chunk, then the visible general ones: bit s = slot 2 chunk + half
__device__ __forceinline__ unsigned pcs2_bits(const AttnParams& p, const PcItem& it, int X, int Y) {
    const unsigned half_bits = ((1u << (2 * p.n_src)) - 1u) << 8;
    return ((pc_item_vis(p, it, X, Y, true) & half_bits) >> 8) | 3u;
}

// ---- dynamic item assignment (AttnParams::queue) ----
// The static walk gives workgroup j of XCD x the j-th window of every tile the XCD takes; items differ in their number of visible
// half chunks and CUs in their speed, so the workgroups run out of items 7.5 % of the launch apart on average (round-4 stamps,
// tools/probe/r04_attn_balance.py: grid launch 2538 us, mean workgroup busy 2346 us).  With a queue the workgroups of an (XCD, head
// group) pull the SAME item sequence in the same order from one counter: whoever is free takes the next window, the windows in
// flight on an XCD stay neighbours (the L2 argument of the tile order is unchanged).  Compute wave 0 pulls (one scalar atomic per
// item, two items ahead) and hands the item word to both roles through a 4-entry LDS ring.
// The decode of a ticket runs once per item on a wave that also does a quarter of the workgroup's MFMA work, so it is kept to a
// few dozen scalar instructions: shifts for the powers of two, multiply-high by a precomputed reciprocal for the rest (a plain
// `/` is ~40 dependent VALU instructions on this target; with five of them per ticket the first version cost 2 us per item).
struct PcsSeq {
    int tl;                    // log2(windows per XCD step)
    int lanes, xs;             // the tiles / list segments of a head group are dealt over `lanes` sequences; XCD x holds number x >> xs
    int seg, n_tiles, TH, ntx, nty;
    unsigned m_seg, m_ego, m_nty, m_ntx;
};
// ceil(2^32 / d): __umulhi(n, m) == n / d for n d < 2^32 (the caller's size guard, hmvit_fusion_forward); d == 1 is handled apart
__device__ __forceinline__ unsigned pcs2_magic(int d) { return d > 1 ? (unsigned)((0x100000000ull + d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int pcs2_div(int n, int d, unsigned m) { return d > 1 ? (int)__umulhi((unsigned)n, m) : n; }
__device__ __forceinline__ PcsSeq pcs2_seq(const AttnParams& p, int X, int Y, int NG, int wpx) {
    const bool hx = (p.variant & 0x800) != 0;
    PcsSeq q;
    const int tps = hx ? wpx : wpx / NG;                 // windows per XCD step: 32 or 16
    q.tl = 31 - __builtin_clz(tps);
    q.lanes = hx ? 8 / NG : 8;
    q.xs = hx ? 31 - __builtin_clz(NG) : 0;
    q.seg = p.sched_sub * tps;                           // list mode: a segment = sched_sub steps of consecutive list entries
    q.TH = tps / 8;
    q.ntx = (X + q.TH - 1) / q.TH;
    q.nty = (Y + 7) / 8;
    q.n_tiles = p.B * p.n_ego * q.ntx * q.nty;
    q.m_seg = pcs2_magic(q.seg); q.m_ego = pcs2_magic(p.n_ego); q.m_nty = pcs2_magic(q.nty); q.m_ntx = pcs2_magic(q.ntx);
    return q;
}
// n-th item of XCD x's sequence -> packed word wy | wx << 10 | ego << 20 | b << 24, 0xffffffff = past the end,
// 0xfffffffe = a hole (outside the map / past the list / pruned): pull again
__device__ __forceinline__ unsigned pcs2_seq_item(const AttnParams& p, const PcsSeq& q, int X, int Y, bool ego_fastest, int x, int n) {
    const int lane0 = x >> q.xs;
    unsigned w;
    if (p.sched) {
        const int ks = pcs2_div(n, q.seg, q.m_seg), r = n - ks * q.seg;
        const int s0 = (lane0 + q.lanes * ks) * q.seg;
        if (s0 >= p.n_sched) return 0xffffffffu;
        const int pos = s0 + r;
        if (pos >= p.n_sched) return 0xfffffffeu;
        const int* a = p.sched + __builtin_amdgcn_readfirstlane(pos);
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(a) : "memory");
    } else {
        const int k = n >> q.tl, t = n - (k << q.tl);    // step of this sequence, window inside the step
        const int T = lane0 + q.lanes * k;               // tile ordinal in the (sample, ego, tile) list
        if (T >= q.n_tiles) return 0xffffffffu;
        int b, ego, trow, tcol;
        if (ego_fastest) {
            const int r1 = pcs2_div(T, p.n_ego, q.m_ego), r2 = pcs2_div(r1, q.nty, q.m_nty);
            ego = T - r1 * p.n_ego; tcol = r1 - r2 * q.nty;
            b = pcs2_div(r2, q.ntx, q.m_ntx); trow = r2 - b * q.ntx;
        } else {
            const int r1 = pcs2_div(T, q.nty, q.m_nty), r2 = pcs2_div(r1, q.ntx, q.m_ntx);
            tcol = T - r1 * q.nty; trow = r1 - r2 * q.ntx;
            b = pcs2_div(r2, p.n_ego, q.m_ego); ego = r2 - b * p.n_ego;
        }
        const int wx = trow * q.TH + (t >> 3), wy = tcol * 8 + (t & 7);
        if (wx >= X || wy >= Y) return 0xfffffffeu;
        w = (unsigned)wy | ((unsigned)wx << 10) | ((unsigned)ego << 20) | ((unsigned)b << 24);
    }
    if (p.prune) {
        const int wy = w & 1023, wx = (w >> 10) & 1023, ego = (w >> 20) & 15, b = w >> 24;
        const unsigned* v_ = p.vis_mask + __builtin_amdgcn_readfirstlane(((b * p.n_ego + ego) * X + wx) * Y + wy);
        unsigned v;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(v_) : "memory");
        if (v >> 31) return 0xfffffffeu;
    }
    return w;
}
// Compute wave 0: turn tickets into the next item word of the ring slot (holes are skipped with further pulls).  The pull is a
// SCALAR atomic (gfx950 executes s_atomic_add: tools/probe/satomic_probe.hip): its result comes back on lgkmcnt.
// When the XCD's own sequence is used up the workgroup goes on with the sequences of the other XCDs that serve its head group
// (`hops` = how many it has left behind): the XCDs differ by a few per cent in speed too, and those items are only the last
// handful of a launch, so that their rows come out of another XCD's L2 hardly matters.
__device__ __forceinline__ int pcs2_pull(int* ctr) {
    int v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    return v;
}
__device__ __forceinline__ void pcs2_publish(const AttnParams& p, const PcsSeq& q, unsigned* ring, int X, int Y, int NG, bool ego_fastest,
                                             int& hops, int slot, int lane) {
    const bool hx = (p.variant & 0x800) != 0;
    const int stride = hx ? NG : 1;                                // XCDs with this workgroup's head group: every stride-th
    unsigned w = 0xffffffffu;
    while (hops < q.lanes) {
        const int x = ((int)(blockIdx.x & 7) + hops * stride) & 7;
        int* ctr = p.queue + (hx ? x : x * NG + pc_head_group(p, NG));
        w = pcs2_seq_item(p, q, X, Y, ego_fastest, x, pcs2_pull(ctr));
        if (w == 0xffffffffu) ++hops;
        else if (w != 0xfffffffeu) break;
    }
    if (lane == 0) ring[slot & 3] = w;
}
__device__ __forceinline__ void pcs2_unpack(const AttnParams& p, unsigned w, int NG, PcItem& it) {
    it.wy = __builtin_amdgcn_readfirstlane((int)(w & 1023));
    it.wx = __builtin_amdgcn_readfirstlane((int)((w >> 10) & 1023));
    it.ego = __builtin_amdgcn_readfirstlane((int)((w >> 20) & 15));
    it.b = __builtin_amdgcn_readfirstlane((int)(w >> 24));
    it.hg = __builtin_amdgcn_readfirstlane(pc_head_group(p, NG));
}

// tables of one item for this loader wave: lane = (slot in the round, key j of the wave)
__device__ __forceinline__ void pcs2_tables(const AttnParams& p, PcShared2& sm, const PcItem& it, int par, int lw, int lane) {
    using SM = PcShared2;
#ifdef HMVIT_EXP_PCS_NOTABLES
    return;
#endif
    const int H = p.H, W = p.W, L = p.L, X = H / 8, Y = W / 8;
    const int j = lane & 7, n_slots = 2 * p.n_src;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && n_slots <= 8) break;
        const int s = (lane >> 3) + 8 * r;
        if (s < n_slots) {
            const int chunk = s >> 1, half = s & 1;
            const int src = pc_src(chunk, it.ego);
            const float* a = sm.ainv + ((it.b * L + src) * L + it.ego) * 8;
            const int n = half * 32 + (j / SM::KPW) * SM::KPP + SM::KPW * lw + (j % SM::KPW);
            int row, col;
            token_pixel(p.partition, 8, X, Y, it.wx, it.wy, n, row, col);
            const bool cav = sm.cav[it.b * L + src] != 0;
            int ix[4];
            float w[4];
            bool vis;
            if (a[6] != 0.f) {   // identity map (the ego itself, or a source at the same pose): the key's own pixel
                ix[0] = row * W + col; ix[1] = ix[2] = ix[3] = -1;
                w[0] = 1.f; w[1] = w[2] = w[3] = 0.f;
                vis = cav;
            } else {
                const Taps t = make_taps(a, col, row, H, W);
                vis = cav && t.roi != 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ix[k] = (t.w[k] != 0.f) ? t.idx[k] : -1;
                    w[k] = t.w[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!vis) ix[k] = -1;
            *reinterpret_cast<int4*>(sm.tidx[par][lw][s][j]) = make_int4(ix[0], ix[1], ix[2], ix[3]);
            *reinterpret_cast<float4*>(sm.tw[par][lw][s][j]) = make_float4(w[0], w[1], w[2], w[3]);
            sm.tmask[par][lw][s][j] = vis ? 0.f : -INFINITY;
            const unsigned bits8 = (unsigned)(__ballot(vis) >> (8 * (lane >> 3))) & 0xffu;
            if (j == 0) sm.tmask[par][lw][s][8] = __int_as_float((bits8 != 0 ? 1 : 0) | (bits8 == 0xffu ? 2 : 0));
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS write -> read ordering
}

// wave-uniform description of one gather (half chunk `slot` of an item)
struct PcGath2 {
    int4v rs;                 // K' plane of the source, head group's channels (V' = + plane_bytes); rs.z = 0: nothing to load
    int par, slot, buf, tsel;
};
__device__ __forceinline__ PcGath2 pcs2_gather(const AttnParams& p, const PcItem& it, const PcItemC& ic, int par, int slot, int g) {
    const int L = p.L, P = p.H * p.W, C = p.C;
    PcGath2 G;
    G.par = par; G.slot = slot; G.buf = g & 1;
    const int chunk = slot >> 1;
    const int src = pc_src(chunk, it.ego), ev = (ic.tev >> 4) & 15;
    G.tsel = (ic.tsel >> (4 * chunk)) & 15;
    const float* kpl = reinterpret_cast<const float*>(p.kv) + ((size_t)((it.b * L + src) * p.E + ev) * 2) * P * C + it.hg * PcShared2::CH;
    G.rs = token_rsrc(kpl, C * 4, P);
    return G;
}

// The 64 x 32 f32 query block of head `head` (absolute) of item t -> 8 KB of LDS at `qlds` by LDS-DMA: piece i = 2 qt + half,
// 64 lanes x 16 bytes; lane (lq, g) of piece (qt, half) takes channels 16 half + 4 g .. + 3 of query qt 16 + lq, so that one
// request reads 64 contiguous bytes per token (as 16-byte pieces 32 bytes apart the 8 requests of an item took 3-6 k cycles to
// issue, round-5 trace).  No registers while it is in flight.
__device__ __forceinline__ void pcs2_request_q(const AttnParams& p, const PcItem& t, int head, unsigned qlds, int lane, int X, int Y) {
    const int lq = lane & 15, g = lane >> 4, L = p.L, W = p.W, C = p.C, P = p.H * p.W;
    const float* qpl = reinterpret_cast<const float*>(p.q) + (size_t)(t.b * L + t.ego) * P * C + head * 32 + g * 4;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        int row, col;
        token_pixel(p.partition, 8, X, Y, t.wx, t.wy, qt * 16 + lq, row, col);
        const float* a = qpl + (size_t)(row * W + col) * C;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float* src = a + 16 * hf;
            const unsigned dst = __builtin_amdgcn_readfirstlane(qlds + (2 * qt + hf) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    }
}

template <bool DYN>
__device__ __forceinline__ void pcs2_loader_loop(const AttnParams& p, PcShared2& sm, int lw, int lane) {
    using SM = PcShared2;
    constexpr int KS = SM::KS, VS = SM::VS, TPK = SM::TPK, KPW = SM::KPW, NP = SM::NP, KPP = SM::KPP;
    const int X = p.H / 8, Y = p.W / 8, NG = p.C / SM::CH;
    const int plane_bytes = p.H * p.W * p.C * 4;
    const int ltid = lw * 64 + lane;
    const int cl = (ltid % TPK) * 4, cl_bytes = cl * 4;
    const int kin = ltid / TPK;                  // key row of this lane inside a KPP-key pass
    const int kj = kin % KPW;
    const bool ego_fastest = (p.variant & 0x200) == 0;
    // destination of this lane's share of the 9-lane mask / flag store: key j = lane of the wave sits in tile row
    // (j / KPW) KPP + KPW lw + j % KPW; lane 8 carries the flags
    const int mrow = (lane / KPW) * KPP + KPW * lw + (lane % KPW);

    uint4v R[NP][2][4];      // the 32 tap loads of the gather in flight - the only state carried from a request to its blend

    auto request = [&](int pass, const PcGath2& N, const int4 ix) {
        const int ixa[4] = {ix.x, ix.y, ix.z, ix.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#ifdef HMVIT_EXP_PCS_NOLOAD
            R[pass][0][k] = tok_load4f(N.rs, -1 - (ixa[k] & 1), cl_bytes, 0);
            R[pass][1][k] = tok_load4f(N.rs, -1 - (ixa[k] & 1), cl_bytes, plane_bytes);
#else
            R[pass][0][k] = tok_load4f(N.rs, ixa[k], cl_bytes, 0);
            R[pass][1][k] = tok_load4f(N.rs, ixa[k], cl_bytes, plane_bytes);
#endif
        }
    };
    // four taps x four channels as packed f32 multiply-adds (v_pk_fma_f32: two channels per instruction, the tap weight
    // broadcast by op_sel) - written on 2-vectors: the packed-split asm below hides the pairs from hipcc's own vectoriser
    auto blend4 = [&](const uint4v (&t)[4], const float4 w, const float4 b, bool ident, float (&o)[4]) {
        const float4v t0 = __builtin_bit_cast(float4v, t[0]), t1 = __builtin_bit_cast(float4v, t[1]);
        const float4v t2 = __builtin_bit_cast(float4v, t[2]), t3 = __builtin_bit_cast(float4v, t[3]);
        const float2v b01 = {b.x, b.y}, b23 = {b.z, b.w};
        float2v a01, a23;
        if (ident) {
            a01 = t0.xy + b01;
            a23 = t0.zw + b23;
        } else {
            a01 = __builtin_elementwise_fma((float2v)(w.x), t0.xy, b01);
            a23 = __builtin_elementwise_fma((float2v)(w.x), t0.zw, b23);
            a01 = __builtin_elementwise_fma((float2v)(w.y), t1.xy, a01);
            a23 = __builtin_elementwise_fma((float2v)(w.y), t1.zw, a23);
            a01 = __builtin_elementwise_fma((float2v)(w.z), t2.xy, a01);
            a23 = __builtin_elementwise_fma((float2v)(w.z), t2.zw, a23);
            a01 = __builtin_elementwise_fma((float2v)(w.w), t3.xy, a01);
            a23 = __builtin_elementwise_fma((float2v)(w.w), t3.zw, a23);
        }
        o[0] = a01.x; o[1] = a01.y; o[2] = a23.x; o[3] = a23.y;
    };

    PcCursor cur = pcs2_cursor();
    PcItem it, itn;
    // item source: the static walk (pc_fetch), or the ring of item words that compute wave 0 fills from the (XCD, head group)
    // counter (pcs2_compute_loop): item i + 1 is read in the step that blends item i's first half chunk
    constexpr bool dyn = DYN;
    int n_items = 0;
    auto fetch = [&](PcItem& out) -> bool {
        if constexpr (!dyn) return pc_fetch(p, X, Y, NG, ego_fastest, cur, out);
        const unsigned w = __builtin_amdgcn_readfirstlane((int)sm.items[n_items & 3]);
        ++n_items;
        if (w == 0xffffffffu) return false;
        pcs2_unpack(p, w, NG, out);
        return true;
    };
    if (!fetch(it)) { pc_wg_barrier(); return; }
    unsigned rest = pcs2_bits(p, it, X, Y), restn = 0;
    PcItemC ic = pcs2_item_consts(p, sm, it), icn = ic;
    bool itn_valid = false;
    int par = 0, g = 0;
    // The query block of the NEXT item is requested by this role (loader wave lw for compute wave lw's head), two steps after the
    // blend of an item's first gather: the compute wave has its own block in registers by then (it converts it in the step after
    // that blend), and the earliest reader of the new block comes behind this step's closing barrier (an item has at least two
    // steps), which a counted wait in front of the barrier covers.  Issued by the compute waves themselves (round 4) the 8 requests
    // of an item took 4-6 k cycles - at priority 0, behind the loaders' tap requests in the vector-memory queue, on the one role
    // whose lateness stalls the whole workgroup at the item boundary (tests/tools/pcs2_trace.py, round 5).
    PcItem qit = it;
    int q_delay = 0;
    const unsigned qlds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm.qstage[lw];
    pcs2_tables(p, sm, it, par, lw, lane);
    PcGath2 G = pcs2_gather(p, it, ic, par, __builtin_ctz(rest), g);
    rest &= rest - 1;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass)
        request(pass, G, *reinterpret_cast<const int4*>(sm.tidx[G.par][lw][G.slot][pass * KPW + kj]));

#pragma unroll 1
    while (true) {
        // ---- the next item's query block (before this step's own fetch can replace qit) ----
        bool q_fired = false;
#ifndef HMVIT_EXP_PCS_NOQ
        if (q_delay > 0 && --q_delay == 0) {
            pcs2_request_q(p, qit, qit.hg * SM::HG + lw, qlds, lane, X, Y);
            q_fired = true;
        }
#endif
        // ---- the gather after G ----
        bool nvalid = true;
        if (rest == 0) {
            nvalid = itn_valid;
            if (itn_valid) { it = itn; ic = icn; rest = restn; par ^= 1; }
        }
        PcGath2 N = G;
        if (nvalid) {
            const int s = __builtin_ctz(rest);
            rest &= rest - 1;
            N = pcs2_gather(p, it, ic, par, s, g + 1);
            if (s == 1) {
                // G is the item's first half chunk: the last reader of the other table set (the blend of the previous item's
                // last gather) is behind us - fetch the next item and build its tables now
                itn_valid = fetch(itn);
                if (itn_valid) {
                    restn = pcs2_bits(p, itn, X, Y);
                    icn = pcs2_item_consts(p, sm, itn);
                    pcs2_tables(p, sm, itn, par ^ 1, lw, lane);
                    qit = itn;
                    q_delay = 2;          // this step blends the item's first gather: fire at the top of the step after the next
                }
            }
        } else {
            N.rs.z = 0;      // empty descriptor: the requests of the closing step return zeros without touching memory
        }
        // ---- one step: blend G pass by pass while N is requested ----
        // (the fields are wave-uniform by construction; saying so keeps the descriptor in SGPRs - without it hipcc wraps every
        // one of the 32 requests in a waterfall loop)
        N.rs.x = __builtin_amdgcn_readfirstlane(N.rs.x); N.rs.y = __builtin_amdgcn_readfirstlane(N.rs.y);
        N.rs.z = __builtin_amdgcn_readfirstlane(N.rs.z); N.rs.w = __builtin_amdgcn_readfirstlane(N.rs.w);
        N.par = __builtin_amdgcn_readfirstlane(N.par); N.slot = __builtin_amdgcn_readfirstlane(N.slot);
        N.buf = __builtin_amdgcn_readfirstlane(N.buf); N.tsel = __builtin_amdgcn_readfirstlane(N.tsel);
        PC2_TRACE(1, lw == 0 && lane == 0, g, 0);
        float4 wt[NP];
        int4 ixn[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            wt[pass] = *reinterpret_cast<const float4*>(sm.tw[G.par][lw][G.slot][pass * KPW + kj]);
            ixn[pass] = *reinterpret_cast<const int4*>(sm.tidx[N.par][lw][N.slot][pass * KPW + kj]);
        }
        const float4 bK = *reinterpret_cast<const float4*>(&sm.bkv[G.tsel][0][cl]);
        const float4 bV = *reinterpret_cast<const float4*>(&sm.bkv[G.tsel][1][cl]);
        const float mk = sm.tmask[G.par][lw][G.slot][lane & 15];
        auto blend_body = [&](int pass) {
            const int kk = pass * KPP + kin;
#ifndef HMVIT_EXP_PCS_NOLOADER
            float k4[4];
            blend4(R[pass][0], wt[pass], bK, false, k4);
#ifndef HMVIT_EXP_PCS_NOSTORE
            store_split4(sm.Kh[G.buf] + kk * KS + cl, sm.Kl[G.buf] + kk * KS + cl, k4);
#else
            if (k4[0] == 1.2345f) store_split4(sm.Kh[G.buf] + kk * KS + cl, sm.Kl[G.buf] + kk * KS + cl, k4);
#endif
            blend4(R[pass][1], wt[pass], bV, false, k4);
#ifndef HMVIT_EXP_PCS_NOSTORE
            store_split4(sm.Vh[G.buf] + kk * VS + cl, sm.Vl[G.buf] + kk * VS + cl, k4);
#else
            if (k4[0] == 1.2345f) store_split4(sm.Vh[G.buf] + kk * VS + cl, sm.Vl[G.buf] + kk * VS + cl, k4);
#endif
#endif
        };
        auto request_body = [&](int pass) {
#ifndef HMVIT_EXP_PCS_NOLOADER
            request(pass, N, ixn[pass]);
#endif
        };
        // Measured and dropped (round 5, same-box A/B): the 8 tap requests of pass p - 1 issued one by one between the vector
        // instructions of the blend of pass p (scheduling groups of 1 request + 5 VALU) instead of back to back behind their own
        // blend - 5.32 against 5.35 ms for the four launches: the vector-memory queue is not what a pass waits for.
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            __builtin_amdgcn_sched_barrier(0);
            blend_body(pass);
            __builtin_amdgcn_sched_barrier(0);
            PC2_TRACE(1, lw == 0 && lane == 0, g, 1 + 2 * pass);
            request_body(pass);
            PC2_TRACE(1, lw == 0 && lane == 0, g, 2 + 2 * pass);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (lane < 9) {
            float* dst = lane < 8 ? &sm.maskadd[G.buf][mrow] : reinterpret_cast<float*>(&sm.vis[G.buf][lw]);
            *dst = mk;
        }
        PC2_TRACE(1, lw == 0 && lane == 0, g, 9);
        // the query block requested at the top of this step is older than the step's 32 tap requests
        if (q_fired) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        pc_wg_barrier();
        PC2_TRACE(1, lw == 0 && lane == 0, g, 10);
        if (!nvalid) break;
        G = N; ++g;
    }
    pc_wg_barrier();   // the interval in which the compute waves consume the last gather
}

template <bool DYN>
__device__ __forceinline__ void pcs2_compute_loop(const AttnParams& p, PcShared2& sm, int wave, int lane, int hops, const PcsSeq& seq) {
    using SM = PcShared2;
    constexpr int KS = SM::KS, VS = SM::VS, LWG = SM::LWG;
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;
    const float LOG2E = 1.4426950408889634f * kl;
    const int hl = wave;
    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / 8, Y = W / 8, NG = C / SM::CH;
    const int lq = lane & 15, g = lane >> 4;
    const bool ego_fastest = (p.variant & 0x200) == 0;
    PcCursor cur = pcs2_cursor();
    PcItem it, itn;
    constexpr bool dyn = DYN;
    int n_items = 0;
    // item i: the static walk, or sm.items[i & 3].  Compute wave 0 draws the tickets: items 0 and 1 before the kernel's last
    // __syncthreads, item i + 2 at the end of its FIRST step of item i - in the time it would otherwise wait at that step's barrier
    // for the loader role (the role that bounds the kernel; there the pull cost 0.1 ms of the 5.6).  Readers: this role at the
    // start of item i + 1, the loader in the step that blends the first half chunk of item i + 1; an item has at least two
    // steps, so both come behind the barrier that follows the write.
    auto fetch = [&](PcItem& out) -> bool {
        if constexpr (!dyn) return pc_fetch(p, X, Y, NG, ego_fastest, cur, out);
        const unsigned w = __builtin_amdgcn_readfirstlane((int)sm.items[n_items & 3]);
        ++n_items;
        if (w == 0xffffffffu) return false;
        pcs2_unpack(p, w, NG, out);
        return true;
    };
    pc_wg_barrier();
    if (!fetch(it)) return;

    // this wave's query block of an item: 4 tiles of 16 queries x the head's 32 channels, lane (lq, g) takes channels 8 g .. 8 g + 7,
    // in the wave's own 8 KB of LDS (piece i = 2 qt + half: 64 lanes x 16 bytes).  The first item's block is requested here; every
    // later one by loader wave `hl` while this wave works on the item before (pcs2_loader_loop)
    const unsigned qlds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm.qstage[hl];
#ifndef HMVIT_EXP_PCS_NOQ
    pcs2_request_q(p, it, it.hg * SM::HG + hl, qlds, lane, X, Y);
#endif

    float4v biasf[7];
    int bias_head = -1;
    half8 qhh[4], qhl[4];
    float m_run[4];          // running row maximum in exponent units: max(logit) * LOG2E (one rounding, shared by every element of the row)
    float4v o_acc[4][2], l_acc[4];
    const half8 ones = (half8)(half_t)1.0f;
    int gstep = 0;
    while (true) {
        PC2_TRACE(0, hl == 0 && lane == 0, gstep, 5);
        const bool nvalid = fetch(itn);
        bool drawn = false;
        const int head = it.hg * SM::HG + hl;
        unsigned todo = pcs2_bits(p, it, X, Y);
        PC2_TRACE(0, hl == 0 && lane == 0, gstep, 6);
        const int h_last = 31 - __builtin_clz(todo);
        if (head != bias_head) {
#pragma unroll
            for (int v = 0; v < 7; ++v)
                biasf[v] = *reinterpret_cast<const float4v*>(p.bias_frag + ((size_t)(head * 7 + v) * 64 + lane) * 4);
            bias_head = head;
        }
        {   // Q + b_q -> (hi, lo) operand halves
#ifndef HMVIT_EXP_PCS_NOQ
            if (gstep == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (later blocks: landed before the loader's barrier)
            PC2_TRACE(0, hl == 0 && lane == 0, gstep, 7);
            const int te = sm.mode[it.b * L + it.ego];
            const float4 b0 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8]);
            const float4 b1 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8 + 4]);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    // this lane's channel quad 2 g + hf sits in piece (qt, g >> 1) at lane slot (2 (g & 1) + hf, lq)
                    const float4 q4 = *reinterpret_cast<const float4*>(&sm.qstage[hl][((2 * qt + (g >> 1)) * 64 + (2 * (g & 1) + hf) * 16 + lq) * 4]);
                    half4 h4, l4;
                    split_pk4(q4.x + bb[4 * hf], q4.y + bb[4 * hf + 1], q4.z + bb[4 * hf + 2], q4.w + bb[4 * hf + 3], h4, l4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        qhh[qt][4 * hf + e] = h4[e];
                        qhl[qt][4 * hf + e] = l4[e];
                    }
                }
                m_run[qt] = -INFINITY;
                l_acc[qt] = (float4v)(0.f);
                o_acc[qt][0] = (float4v)(0.f);
                o_acc[qt][1] = (float4v)(0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the block is in registers before its LDS rows are requested again
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);        // (raised for the epilogue of the item before: the whole item boundary runs at the loaders' level)
            PC2_TRACE(0, hl == 0 && lane == 0, gstep, 8);
#else
            for (int qt = 0; qt < 4; ++qt) { m_run[qt] = -INFINITY; l_acc[qt] = (float4v)(0.f); o_acc[qt][0] = (float4v)(0.f); o_acc[qt][1] = (float4v)(0.f); qhh[qt] = ones; qhl[qt] = ones; }
#endif
        }
        while (todo) {
            const int h = __builtin_ctz(todo);
            todo &= todo - 1;
            const int buf = gstep & 1;
            PC2_TRACE(0, hl == 0 && lane == 0, gstep, 0);
            int vis_or = 0, vis_and = 3;
#pragma unroll
            for (int w = 0; w < LWG; ++w) {
                vis_or |= sm.vis[buf][w];
                vis_and &= sm.vis[buf][w];
            }
            const bool any_visible = (vis_or & 1) != 0;
            const bool all_visible = (vis_and & 2) != 0;
#ifdef HMVIT_EXP_PCS_NOMATH
            if (false) {
#else
            if (any_visible || !p.skip_masked) {
#endif
                float4v madd[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) madd[kt] = (float4v)(0.f);
                if (!all_visible) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) madd[kt] = *reinterpret_cast<const float4v*>(sm.maskadd[buf] + kt * 16 + 4 * g);
                }
                half8 khh[2], khl[2], vhh[2], vhl[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    khh[kt] = *reinterpret_cast<const half8*>(sm.Kh[buf] + (kt * 16 + lq) * KS + hl * 32 + g * 8);
                    khl[kt] = *reinterpret_cast<const half8*>(sm.Kl[buf] + (kt * 16 + lq) * KS + hl * 32 + g * 8);
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // V^T tile dt takes the head's channels 8 q4 + 4 dt + r (q4 = lq & 3) as its rows 4 q4 + r (as k_attention_pc)
                    const int off = (4 * g + (lq >> 2)) * VS + hl * 32 + (lq & 3) * 8 + dt * 4;
#pragma unroll
                    for (int hlx = 0; hlx < 2; ++hlx) {
                        const half_t* base = (hlx ? sm.Vl[buf] : sm.Vh[buf]) + off;
                        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                        half8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (half_t)lo[e];
                            v[4 + e] = (half_t)hi[e];
                        }
                        if (hlx) vhl[dt] = v; else vhh[dt] = v;
                    }
                }
                // One 16-query tile against the 32 keys of the buffer.  HK (which half of the 64 key positions: bias tiles 2 HK + kt)
                // and ALLV (every key visible: no mask) are compile-time so that the bias fragment is a plain MFMA C operand and
                // the mask costs nothing when there is none: with run-time selects hipcc spent 24 of ~90 VALU instructions per
                // tile on v_cndmask (round-4 ISA reading; the role is bound by its instruction stream, 4.2 ms of 6.0).
                // Exponent: p = exp2(s c - M) as ONE fma per element, M = max(s) c rounded once per row (the same M enters
                // alpha and every element, so its rounding cancels in the normalisation).
                auto tile = [&](auto hk_c, auto allv_c, int qt_rt) {
                    constexpr int HK = decltype(hk_c)::value;
                    constexpr bool ALLV = decltype(allv_c)::value;
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt) {
                        float4v s[2];
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            const float4v b0 = biasf[qt - kt + 3], b1 = biasf[qt - kt + 1 >= 0 ? qt - kt + 1 : 0];
                            float4v acc = (h & 1) ? b1 : b0;
                            acc += madd[kt];
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khl[kt], qhh[qt], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh[kt], qhl[qt], acc, 0, 0, 0);
                            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh[kt], qhh[qt], acc, 0, 0, 0);
                        }
                        float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
                        mx = fmaxf(mx, fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
                        mx = max_over_lane_groups(mx);
                        const float m_new = max_raw(m_run[qt], mx * LOG2E);
                        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_safe);
                        half8 ph, pl;
                        {
                            float e8[8];
#pragma unroll
                            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                                for (int r = 0; r < 4; ++r) e8[4 * kt + r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], LOG2E, -m_safe));
                            split_pk8(e8, ph, pl);
                        }
                        m_run[qt] = m_new;
                        o_acc[qt][0] *= alpha;
                        o_acc[qt][1] *= alpha;
                        l_acc[qt] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhl[dt], ph, o_acc[qt][dt], 0, 0, 0);
                            o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhh[dt], pl, o_acc[qt][dt], 0, 0, 0);
                            o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhh[dt], ph, o_acc[qt][dt], 0, 0, 0);
                        }
                        l_acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pl, l_acc[qt], 0, 0, 0);
                        l_acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, l_acc[qt], 0, 0, 0);
#ifndef HMVIT_EXP_PCS_Q4
                        if (qt & 1) __builtin_amdgcn_sched_barrier(0);   // two tiles in flight at a time: four overflow the register file
#endif
                        if (qt & 1) PC2_TRACE(0, hl == 0 && lane == 0, gstep, 1 + (qt >> 1));
                    }
                };
                using F_ = std::false_type;
                tile(std::integral_constant<int, 0>{}, F_{}, 0);
            }
#ifdef HMVIT_EXP_PCS_NOEPI
            if (false) {
#else
            if (h == h_last) {
#endif
                PC2_TRACE(0, hl == 0 && lane == 0, gstep, 10);
                // the loader role runs at priority 3 and owns the issue slots and the vector-memory queue most of the time: at
                // priority 0 the 8 output stores of an item took 4-11 k cycles (tests/tools/pcs2_trace.py, round 5)
                __builtin_amdgcn_s_setprio(3);
                float* outp = reinterpret_cast<float*>(p.out) + (size_t)(it.b * L + it.ego) * P * C;
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    int row, col;
                    token_pixel(p.partition, 8, X, Y, it.wx, it.wy, qt * 16 + lq, row, col);
                    const float inv = 1.f / l_acc[qt][0];
                    // lane (lq, g) holds channel quads 2 g and 2 g + 1 of its token; stored like that, one store instruction writes
                    // 16-byte pieces 32 bytes apart (64 of them) and the item's 32 stores held the vector-memory path for 4-10 k
                    // cycles (round-5 trace).  Two row swaps per register (lanes l, l ^ 16, l ^ 32, l ^ 48 belong to one token) give
                    // lane g quads g and 4 + g: each store writes 64 contiguous bytes per token.
                    float a[4] = {o_acc[qt][0][0] * inv, o_acc[qt][0][1] * inv, o_acc[qt][0][2] * inv, o_acc[qt][0][3] * inv};
                    float b[4] = {o_acc[qt][1][0] * inv, o_acc[qt][1][1] * inv, o_acc[qt][1][2] * inv, o_acc[qt][1][3] * inv};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        swap16_rows(a[e], b[e]);      // a: quads (0, 1, 4, 5) by row, b: (2, 3, 6, 7)
                        swap32_rows(a[e], b[e]);      // a: (0, 1, 2, 3),          b: (4, 5, 6, 7)
                    }
                    float* o = outp + (size_t)(row * W + col) * C + head * 32 + 4 * g;
                    *reinterpret_cast<float4*>(o) = make_float4(a[0], a[1], a[2], a[3]);
                    *reinterpret_cast<float4*>(o + 16) = make_float4(b[0], b[1], b[2], b[3]);
                    if (p.lse && g == 0)
                        p.lse[((size_t)(it.b * L + it.ego) * P + row * W + col) * (C / 32) + head] = m_run[qt] * 0.6931471805599453f + logf(l_acc[qt][0]);
                }
                // (the raised priority is kept across the item boundary: back to 0 behind the next item's query conversion)
            }
            if (dyn && hl == 0 && !drawn) pcs2_publish(p, seq, sm.items, X, Y, NG, ego_fastest, hops, n_items, lane);
            drawn = true;
            PC2_TRACE(0, hl == 0 && lane == 0, gstep, 3);
            pc_wg_barrier();
            PC2_TRACE(0, hl == 0 && lane == 0, gstep, 4);
            ++gstep;
        }
        if (!nvalid) break;
        it = itn;
    }
}

template <bool DYN>
__global__ __launch_bounds__(512) void k_attention_pcs2(AttnParams p) {
    using SM = PcShared2;
    __shared__ __attribute__((aligned(16))) SM sm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hg = pc_head_group(p, p.C / SM::CH);
    {
        const int n_rec = p.B * p.L * p.L * 8;
        for (int i = threadIdx.x; i < n_rec; i += blockDim.x) sm.ainv[i] = p.ainv[i];
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * HMVIT_NUM_TYPES * 2 * SM::CH; i += blockDim.x) {
            const int e = i / (2 * SM::CH), pl = (i / SM::CH) & 1, c = i % SM::CH;
            sm.bkv[e][pl][c] = p.b_kv[(size_t)e * 2 * p.C + pl * p.C + hg * SM::CH + c];
        }
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * SM::CH; i += blockDim.x)
            sm.bq[i / SM::CH][i % SM::CH] = p.b_q[(i / SM::CH) * p.C + hg * SM::CH + (i % SM::CH)];
        if (threadIdx.x < kMaxSlots) {
            sm.mode[threadIdx.x] = p.mode[threadIdx.x];
            sm.cav[threadIdx.x] = p.cav[threadIdx.x];
            sm.ego_e[threadIdx.x] = p.ego_e[threadIdx.x];
        }
    }
    int hops = 0;
    PcsSeq seq = {};
    if (DYN && wave == 0) {                 // dynamic item assignment: compute wave 0 draws the first two items (pcs2_compute_loop)
        const int NG = p.C / SM::CH;
        seq = pcs2_seq(p, p.H / 8, p.W / 8, NG, kPcs2Grid >> 3);
        for (int i = 0; i < 2; ++i) pcs2_publish(p, seq, sm.items, p.H / 8, p.W / 8, NG, (p.variant & 0x200) == 0, hops, i, threadIdx.x & 63);
    }
    __syncthreads();
    pcs2_fill_consts(p, sm);
    __syncthreads();
#ifdef HMVIT_PROBE
    if (p.trace && threadIdx.x == 0) p.trace[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();       // 100 MHz
#endif
    if (wave >= SM::CWG) {
#ifdef HMVIT_EXP_PCS_LPRIO
        __builtin_amdgcn_s_setprio(HMVIT_EXP_PCS_LPRIO);
#else
        __builtin_amdgcn_s_setprio(3);
#endif
        pcs2_loader_loop<DYN>(p, sm, wave - SM::CWG, threadIdx.x & 63);
    } else {
        pcs2_compute_loop<DYN>(p, sm, wave, threadIdx.x & 63, hops, seq);
    }
#ifdef HMVIT_PROBE
    // when this workgroup ran out of items (probe builds: the spread of these over the 256 workgroups is what the static item
    // partition costs, tools/probe/r04_attn_balance.py)
    if (p.trace && threadIdx.x == 0) p.trace[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

static int launch_attn_pcs(const AttnParams& p_in, hipStream_t st) {
    AttnParams p = p_in;
    if (const char* e = HMVIT_ENV("HMVIT_ATTN_TRACE")) p.trace = (unsigned long long*)strtoull(e, nullptr, 0);
    // items: pulled from per-(XCD, head group) counters when the caller provides them (AttnParams::queue), else the static walk
    if (p.queue) hipLaunchKernelGGL(k_attention_pcs2<true>, dim3(kPcs2Grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(k_attention_pcs2<false>, dim3(kPcs2Grid), dim3(512), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

#include "attn_patch.hpp"
#include "attn_patch16.hpp"

template <typename T, int WIN, int HG>
static int launch_attn_t(const AttnParams& p, hipStream_t st) {
    const int NG = p.C / (HG * 32);
    dim3 grid((p.H / WIN) * (p.W / WIN) * NG, p.n_ego, p.B);
    hipLaunchKernelGGL((k_attention<T, WIN, HG>), grid, dim3(HG * 64), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// Generic shapes: any window (N = window^2 <= 256 tokens) and any dim_head <= 64 (hetero_fusion.py:187-277 takes both from the
// yaml; the shipped configs use 8 and 32, which the MFMA kernels above serve).  Plain f32 FMAs, one thread per query, one
// workgroup per (window, head, ego): keys in chunks of 64 gathered into LDS (same taps, biases and visibility rule as k_attention),
// online softmax per thread.  Correct for every shape the reference accepts, two orders of magnitude slower than the MFMA kernels:
// a fallback, not a tuned path.  The relative-position bias arrives dense: (heads, N, N) (weights.py bias_dense).
// ------------------------------------------------------------------------------------------
constexpr int ANY_KC = 64;
template <int DHM>
__global__ __launch_bounds__(256) void k_attention_any(AttnParams p) {
    __shared__ float Ks[ANY_KC][DHM + 1], Vs[ANY_KC][DHM + 1], maskadd[ANY_KC];
    const int WIN = p.window, N = WIN * WIN, DH = p.dim_head, C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int heads = C / DH, X = H / WIN, Y = W / WIN;
    const int win = blockIdx.x / heads, head = blockIdx.x - win * heads;
    const int ego = blockIdx.y, b = blockIdx.z;
    const int wx = win / Y, wy = win - wx * Y;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int te = p.mode[b * L + ego], ev = p.ego_e[b * L + ego];
    const int ch0 = head * DH;
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;
    const float* qplane = reinterpret_cast<const float*>(p.q) + (size_t)(b * L + ego) * P * C;
    const float* kvplanes = reinterpret_cast<const float*>(p.kv);
    const bool active = tid < N;

    float q[DHM], o[DHM];
#pragma unroll
    for (int d = 0; d < DHM; ++d) q[d] = o[d] = 0.f;
    int qrow = 0, qcol = 0;
    if (active) {
        token_pixel(p.partition, WIN, X, Y, wx, wy, tid, qrow, qcol);
        const float* a = p.ainv + ((size_t)(b * L + ego) * L + ego) * 8;
        const float* bq = p.b_q + te * C + ch0;
        if (a[6] != 0.f) {
            const float* src = qplane + (size_t)(qrow * W + qcol) * C + ch0;
#pragma unroll
            for (int d = 0; d < DHM; ++d)
                if (d < DH) q[d] = src[d] + bq[d];
        } else {
            const Taps t = make_taps(a, qcol, qrow, H, W);
#pragma unroll
            for (int d = 0; d < DHM; ++d)
                if (d < DH) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(t.w[k], qplane[(size_t)t.idx[k] * C + ch0 + d], acc);
                    q[d] = acc + bq[d];
                }
        }
    }
    float m_run = -INFINITY, l_run = 0.f;
    const float* bias = p.bias_frag + ((size_t)head * N + (active ? tid : 0)) * N;     // row of this query

    const int n_keys = p.n_src * N;
    for (int k0 = 0; k0 < n_keys; k0 += ANY_KC) {
        // gather: key kk = tid % 64, channels d = tid / 64, + nthr / 64, ...
        int any_visible = 0;
        {
            const int kk = tid % ANY_KC, key = k0 + kk;
            bool visible = false;
            Taps t;
            bool ident = false;
            int self_idx = 0, src = 0;
            if (key < n_keys) {
                src = key / N;
                int row, col;
                token_pixel(p.partition, WIN, X, Y, wx, wy, key - src * N, row, col);
                const float* a = p.ainv + ((size_t)(b * L + src) * L + ego) * 8;
                ident = a[6] != 0.f;
                self_idx = row * W + col;
                t.roi = 1.f;
                if (!ident) t = make_taps(a, col, row, H, W);
                visible = (t.roi != 0.f) && (p.cav[b * L + src] != 0);
            }
            const int ts = p.mode[b * L + src];
            const float* kpl = kvplanes + ((size_t)((b * L + src) * p.E + ev) * 2) * P * C + ch0;
            const float* bk = p.b_kv + (size_t)(te * HMVIT_NUM_TYPES + ts) * 2 * C + ch0;
            for (int d = tid / ANY_KC; d < DHM; d += nthr / ANY_KC) {     // channels DH .. DHM - 1 are zero padding
                // (no divergent region around the tap loads: an invisible key, or a padding channel, loads pixel 0 / channel 0 and the
                // result is selected afterwards - see k_attention_bwd in train.hip for what a divergent gather did at two workgroups per CU)
                float kvv[2] = {0.f, 0.f};
                const bool live = visible && d < DH;
                const int dd = d < DH ? d : 0;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const float* plane = kpl + (size_t)pl * P * C + dd;
                    const int i0 = live ? (ident ? self_idx : t.idx[0]) : 0;
                    float acc = (ident ? 1.f : t.w[0]) * plane[(size_t)i0 * C];
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const int ik = (live && !ident) ? t.idx[k] : 0;
                        acc = fmaf((live && !ident) ? t.w[k] : 0.f, plane[(size_t)ik * C], acc);
                    }
                    kvv[pl] = live ? acc + bk[pl * C + dd] : 0.f;
                }
                Ks[kk][d] = kvv[0];
                Vs[kk][d] = kvv[1];
            }
            if (tid < ANY_KC) maskadd[kk] = visible ? 0.f : -INFINITY;
            any_visible = visible ? 1 : 0;
        }
        any_visible = __syncthreads_or(any_visible);
        if ((any_visible || !p.skip_masked) && active) {
            const int kmax = min(ANY_KC, n_keys - k0);
            for (int kk = 0; kk < kmax; ++kk) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DHM; ++d) s = fmaf(q[d], Ks[kk][d], s);       // channels >= DH hold zeros on the query side
                const int kn = (k0 + kk) % N;
                s = s * kl + bias[kn] + maskadd[kk];
                const float m_new = fmaxf(m_run, s);
                if (m_new == -INFINITY) continue;
                const float alpha = expf(m_run - m_new), e = expf(s - m_new);
                l_run = l_run * alpha + e;
#pragma unroll
                for (int d = 0; d < DHM; ++d) o[d] = fmaf(e, Vs[kk][d], o[d] * alpha);
                m_run = m_new;
            }
        }
        __syncthreads();
    }
    if (active) {
        const float inv = 1.f / l_run;
        float* op = reinterpret_cast<float*>(p.out) + (size_t)(b * L + ego) * P * C + (size_t)(qrow * W + qcol) * C + ch0;
#pragma unroll
        for (int d = 0; d < DHM; ++d)
            if (d < DH) op[d] = o[d] * inv;
        if (p.lse) p.lse[((size_t)(b * L + ego) * P + qrow * W + qcol) * heads + head] = m_run + logf(l_run);
    }
}

static int launch_attn_any(const AttnParams& p, hipStream_t st) {
    const int N = p.window * p.window, DH = p.dim_head;
    HMVIT_CHECK_ARG(N <= 256, "attention (generic): window=%d gives %d tokens per window (at most 256)", p.window, N);
    HMVIT_CHECK_ARG(DH >= 1 && DH <= 64 && p.C % DH == 0, "attention (generic): dim_head=%d unsupported (1 .. 64, a divisor of C=%d)", DH, p.C);
    const int threads = N <= 64 ? 64 : (N + 63) / 64 * 64;
    dim3 grid((p.H / p.window) * (p.W / p.window) * (p.C / DH), p.n_ego, p.B);
    if (DH <= 16) hipLaunchKernelGGL(k_attention_any<16>, grid, dim3(threads), 0, st, p);
    else if (DH <= 32) hipLaunchKernelGGL(k_attention_any<32>, grid, dim3(threads), 0, st, p);
    else hipLaunchKernelGGL(k_attention_any<64>, grid, dim3(threads), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// Which (ego, window, source) tiles have any visible key?  One wave per window, lane = key: the same tap
// arithmetic as the loader (make_taps' nearest-pixel ROI test x the agent-validity mask).  In the local stage of
// cfg2 a third of the (ego, source != ego, window) tiles lie entirely outside the source's field of view; the
// persistent kernel skips them outright (the reference computes them densely and masks them to -inf: same result,
// tests/test_hip_fusion.py::test_skip_masked_tiles_is_exact_f16).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_vis(AttnParams p, unsigned* __restrict__ vis_mask, const unsigned char* __restrict__ need) {
    const int X = p.H / 8, Y = p.W / 8, n_pos = p.B * p.n_ego * X * Y;
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p.queue && blockIdx.x == 0 && threadIdx.x < 16) p.queue[threadIdx.x] = 0;   // the pull counters of the attention launch that follows
    if (p.tail_pull && blockIdx.x == 0 && threadIdx.x >= 16 && threadIdx.x < 64) p.tail_pull[threadIdx.x - 16] = 0;   // and of the stage's tails
    if (pos >= n_pos) return;
    int r = pos;
    const int wy = r % Y; r /= Y;
    const int wx = r % X; r /= X;
    const int ego = r % p.n_ego, b = r / p.n_ego;
    int row, col;
    token_pixel(p.partition, 8, X, Y, wx, wy, lane, row, col);
    unsigned mask = 0;
    for (int c = 0; c < p.n_src; ++c) {
        const int src = pc_src(c, ego);
        const float* a = p.ainv + ((size_t)(b * p.L + src) * p.L + ego) * 8;
        bool vis = p.cav[b * p.L + src] != 0;
        if (a[6] == 0.f) vis = vis && make_taps(a, col, row, p.H, p.W).roi != 0.f;
        const unsigned long long bal = __ballot(vis);
        if (bal) mask |= 1u << c;
        // bits 8 + 2c / 9 + 2c: keys 0..31 / 32..63 of the chunk (the split-precision kernel walks 32-key half chunks)
        if (bal & 0xffffffffull) mask |= 1u << (8 + 2 * c);
        if (bal >> 32) mask |= 1u << (9 + 2 * c);
    }
    if (need && !need[pos]) mask |= 0x80000000u;      // local partition only: pos = (b, ego, window) as in k_window_need
    if (lane == 0) vis_mask[pos] = mask;
}

// ------------------------------------------------------------------------------------------
// Reachability of the stages before the pruned last one.  The last stage of HeteroFusion only produces ego 0's row, so
// of the other agents' maps it reads nothing but the K' / V' rows under ego 0's bilinear taps; a window of agent j
// without such a tap is dead code in the (local) stage before: its attention item and its chain tail are skipped (the
// reference computes them and discards them).  One thread per ego pixel and source agent marks the windows of the taps
// it would load (same criterion as pc_taps, without the visibility test: a superset).  from = nullptr: what ego 0
// reads over its whole map; from = such a table: what all egos read from the windows they still compute, plus those
// windows themselves - the tokens the chain tail of the stage before that one has to produce.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_need(AttnParams p, const unsigned char* __restrict__ from,
                                                     unsigned char* __restrict__ to) {
    const int H = p.H, W = p.W, X = H / 8, Y = W / 8, XY = X * Y;
    const int pix = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int n_from = from ? p.n_ego : 1, j = blockIdx.z % n_from, b = blockIdx.z / n_from;
    if (pix >= H * W) return;
    const int row = pix / W, col = pix - row * W, own = (row >> 3) * Y + (col >> 3);
    if (from && !from[(size_t)(b * p.n_ego + j) * XY + own]) return;      // ego j does not compute this pixel's window
    unsigned char* base = to + (size_t)(b * p.n_ego + k) * XY;
    const float* a = p.ainv + ((size_t)(b * p.L + k) * p.L + j) * 8;     // source k sampled on ego j's grid
    if (k == j || a[6] != 0.f) { base[own] = 1; return; }
    if (!p.cav[b * p.L + k]) return;
    const Taps t = make_taps(a, col, row, H, W);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t.w[i] != 0.f) {
            const int r = t.idx[i] / W, c = t.idx[i] - r * W;
            base[(r >> 3) * Y + (c >> 3)] = 1;
        }
}

int launch_window_need(const AttnParams& p, const unsigned char* from, unsigned char* to, hipStream_t st) {
    HMVIT_CHECK_ARG(p.window == 8 && p.H % 8 == 0 && p.W % 8 == 0, "window_need: window=%d (8)", p.window);
    if (p.B <= 0 || p.n_ego <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_window_need, dim3(cdiv(p.H * p.W, 256), p.n_ego, p.B * (from ? p.n_ego : 1)), dim3(256), 0, st, p, from, to);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// World-ordered work list of the local (window) stages.  The rows of source j that a window gathers lie under the window's
// WORLD position, whichever ego owns it: windows of different egos over the same ground read the same rows of every source map.
// The (sample, ego, window) items are therefore sorted by the position of their centre in the frame of agent 0 of the sample
// (through the pair affines the gather itself uses) along a boustrophedon of 16-pixel bands cut into 8-pixel cells, and the
// persistent kernels walk that list (pc_fetch_sched): the items an XCD holds at any time then share their gathers in its L2
// across egos - each source row is fetched about once per XCD instead of once per ego.  Counting sort, three small launches.
// ------------------------------------------------------------------------------------------
constexpr int SCHED_MARGIN = 256, SCHED_BAND = 16, SCHED_CELL = 8;
static inline int sched_cols(int W) { return cdiv(W + 2 * SCHED_MARGIN, SCHED_CELL); }
static inline int sched_keys(int H, int W) { return cdiv(H + 2 * SCHED_MARGIN, SCHED_BAND) * sched_cols(W); }
size_t attn_schedule_bytes(int B, int n_ego, int H, int W) {
    const size_t n = (size_t)B * n_ego * (H / 8) * (W / 8);
    return (2 * n + (size_t)B * sched_keys(H, W) + 1) * sizeof(int) + 256;
}

__device__ __forceinline__ int sched_key(const AttnParams& p, int b, int ego, int wx, int wy) {
    const float* a = p.ainv + ((size_t)(b * p.L + 0) * p.L + ego) * 8;      // ego pixel -> pixel of agent 0 of the sample
    const float u = wy * 8 + 3.5f, v = wx * 8 + 3.5f;
    float sx = u, sy = v;
    if (a[6] == 0.f) {
        sx = fmaf(a[0], u, fmaf(a[1], v, a[2]));
        sy = fmaf(a[3], u, fmaf(a[4], v, a[5]));
    }
    const int He = p.H + 2 * SCHED_MARGIN, We = p.W + 2 * SCHED_MARGIN;
    const int r = (int)fminf(fmaxf(sy + SCHED_MARGIN, 0.f), (float)(He - 1));
    const int c = (int)fminf(fmaxf(sx + SCHED_MARGIN, 0.f), (float)(We - 1));
    const int ncol = (We + SCHED_CELL - 1) / SCHED_CELL, band = r / SCHED_BAND, col = c / SCHED_CELL;
    const int nband = (He + SCHED_BAND - 1) / SCHED_BAND;
    return (b * nband + band) * ncol + ((band & 1) ? ncol - 1 - col : col);
}
__global__ __launch_bounds__(256) void k_sched_count(AttnParams p, int n, int* __restrict__ keys, int* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int X = p.H / 8, Y = p.W / 8;
    const int wy = i % Y, wx = (i / Y) % X, ego = (i / (X * Y)) % p.n_ego, b = i / (X * Y * p.n_ego);
    const int k = sched_key(p, b, ego, wx, wy);
    keys[i] = k;
    atomicAdd(hist + k, 1);
}
// exclusive prefix sum of hist[0, n) in place, one workgroup
__global__ __launch_bounds__(1024) void k_sched_scan(int* __restrict__ hist, int n) {
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) {
        const int c = hist[i];
        hist[i] = run;
        run += c;
    }
}
__global__ __launch_bounds__(256) void k_sched_fill(AttnParams p, int n, const int* __restrict__ keys, int* __restrict__ offs,
                                                    int* __restrict__ sched) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int X = p.H / 8, Y = p.W / 8;
    const int wy = i % Y, wx = (i / Y) % X, ego = (i / (X * Y)) % p.n_ego, b = i / (X * Y * p.n_ego);
    const int pos = atomicAdd(offs + keys[i], 1);   // order inside a cell is immaterial (items are independent)
    sched[pos] = wy | (wx << 10) | (ego << 20) | (b << 24);
}
// ws: attn_schedule_bytes(B, n_ego, H, W) bytes; the list (n = B n_ego H/8 W/8 packed items) starts at ws
int launch_attn_schedule(const AttnParams& p, int* ws, hipStream_t st) {
    const int X = p.H / 8, Y = p.W / 8, n = p.B * p.n_ego * X * Y;
    HMVIT_CHECK_ARG(X <= 1024 && Y <= 1024 && p.n_ego <= 16 && p.B <= 128, "attention schedule: map %dx%d / n_ego=%d / B=%d too large", p.H, p.W, p.n_ego, p.B);
    const int nk = p.B * sched_keys(p.H, p.W);
    int* sched = ws;
    int* keys = ws + n;
    int* hist = ws + 2 * n;
    HMVIT_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)(nk + 1) * sizeof(int), st));
    hipLaunchKernelGGL(k_sched_count, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n, keys, hist);
    hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(1024), 0, st, hist, nk);
    hipLaunchKernelGGL(k_sched_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n, keys, hist, sched);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// number of items the schedule keeps (bit 31 clear): the profile's count of the attention items actually run (bench.py's
// roofline numerator)
__global__ __launch_bounds__(256) void k_count_live(const unsigned* __restrict__ vis_mask, int n, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n && !(vis_mask[i] >> 31);
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, __popcll(b));
}
int launch_count_live(const unsigned* vis_mask, int n, int* out, hipStream_t st) {
    if (n <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_count_live, dim3(cdiv(n, 256)), dim3(256), 0, st, vis_mask, n, out);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_tile_vis(const AttnParams& p, unsigned* vis_mask, const unsigned char* need, hipStream_t st) {
    HMVIT_CHECK_ARG(p.n_src <= 8 && p.window == 8, "tile_vis: n_src=%d (<= 8), window=%d (8)", p.n_src, p.window);
    const int n_pos = p.B * p.n_ego * (p.H / 8) * (p.W / 8);
    if (n_pos <= 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(!need || p.partition == HMVIT_PART_WINDOW, "tile_vis: reachability pruning is for the local partition");
    hipLaunchKernelGGL(k_tile_vis, dim3(cdiv(n_pos, 4)), dim3(256), 0, st, p, vis_mask, need);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_attention(const AttnParams& p, int precision, hipStream_t st) {
    HMVIT_CHECK_ARG(p.window >= 1 && p.H % p.window == 0 && p.W % p.window == 0, "attention: %dx%d not divisible by window %d",
                    p.H, p.W, p.window);
    HMVIT_CHECK_ARG(p.B * p.L <= kMaxSlots, "attention: B*L=%d exceeds %d per launch", p.B * p.L, kMaxSlots);
    if (p.n_ego <= 0 || p.B <= 0) return HMVIT_OK;
    const int dim_head = p.dim_head ? p.dim_head : 32;
    if ((p.window != 4 && p.window != 8) || dim_head != 32) {
        HMVIT_CHECK_ARG(precision == HMVIT_PREC_F32, "attention: window=%d / dim_head=%d run in the exact-f32 mode only (generic kernel)",
                        p.window, dim_head);
        AttnParams q = p;
        q.dim_head = dim_head;
        return launch_attn_any(q, st);
    }
    HMVIT_CHECK_ARG(p.C == 64 || p.C == 128 || p.C == 256, "attention: C=%d unsupported (64, 128, 256)", p.C);
    const bool w8 = p.window == 8;
    if (precision == HMVIT_PREC_SPLIT && w8 && p.C >= 128 && p.self_identity && p.n_src <= 8 &&
        p.B * p.L * p.L <= PcShared2::MAX_PAIRS) {
        AttnParams q = p;
        // local stages over rigid transforms: the de-duplicated patch kernel (attn_patch.hpp)
        if (p.partition == HMVIT_PART_WINDOW && p.C == 256 && p.rigid_patch && p.n_src <= PatchShared::NCH + 1 && !HMVIT_ENV("HMVIT_NO_PATCH"))
            return (p.rigid_patch == 2 && p.patch_tab) ? launch_attn_patch16(q, st) : launch_attn_patch(q, st);
        if (p.partition == HMVIT_PART_GRID) q.variant ^= 0x200;   // item order as for k_attention_pc
        // head group per XCD: bit 0 grid stages, bit 1 local stages.  Measured at cfg2 (tools/probe/r03_attn_ab.sh): local stages
        // 1660 -> 1593 us and 4.56 -> 4.13 GB fetched; the grid stages fetch 18 % less (11.3 -> 9.2 GB) but run 6 % SLOWER
        // (2620 -> 2780 us: their time is not set by bytes), so only the local stages take it
        int hgx = 2;
        if (const char* e = HMVIT_ENV("HMVIT_ATTN_HGX")) hgx = atoi(e);
        if ((8 % (p.C / PcShared2::CH)) == 0 && ((p.partition == HMVIT_PART_GRID) ? (hgx & 1) : (hgx & 2))) q.variant |= 0x800;
        return launch_attn_pcs(q, st);
    }
    if (precision == HMVIT_PREC_F32 || precision == HMVIT_PREC_SPLIT) {   // f32 planes, exact-f32 MFMA
        return w8 ? launch_attn_t<float, 8, 2>(p, st) : launch_attn_t<float, 4, 2>(p, st);
    }
    if (p.C == 64) return w8 ? launch_attn_t<half_t, 8, 2>(p, st) : launch_attn_t<half_t, 4, 2>(p, st);
    int variant = p.variant;
    if (const char* e = HMVIT_ENV("HMVIT_ATTN_DEBUG")) variant ^= atoi(e);   // probe switches (tools/attn_probe.py, tools/attn_diff.py)
    if (const char* e = HMVIT_ENV("HMVIT_ATTN_VARIANT")) variant ^= (int)strtol(e, nullptr, 0);   // same, without switching the visibility table off
    if (w8 && (variant & 1) == 0 && p.B * p.L * p.L <= PcShared<4, 1, 1>::MAX_PAIRS) {
        AttnParams q = p;
        q.variant = variant;
        // item order: egos interleaved per window for the local partition (cross-ego cache reuse of the
        // gathered rows), ego-major for the dilated grid partition
        if (p.partition == HMVIT_PART_GRID) q.variant ^= 0x200;
        if (const char* e = HMVIT_ENV("HMVIT_ATTN_TILE44")) {      // probe: 4 x 4 window tiles (1: grid stages, 2: local, 3: both)
            const int m = atoi(e);
            if ((p.partition == HMVIT_PART_GRID && (m & 1)) || (p.partition != HMVIT_PART_GRID && (m & 2))) q.variant |= 0x2000;
            if ((p.partition == HMVIT_PART_GRID && (m & 4)) || (p.partition != HMVIT_PART_GRID && (m & 8))) q.variant |= 0x4000;
        }
        if (const char* e = HMVIT_ENV("HMVIT_ATTN_TRACE")) q.trace = (unsigned long long*)strtoull(e, nullptr, 0);
        // wave configurations (heads per group, compute waves per head, loader waves per head):
        //   default: 4 heads, 1 + 1 -> 8 waves of <= 256 VGPRs: a loader wave keeps 4 passes = 32 tap loads
        //            per lane in flight
        //   variant bit 2: 4 heads, 2 + 2 -> 16 waves of <= 128 VGPRs;  bit 4: 2 heads, 1 + 1, two workgroups per CU
        if (q.variant & 2) return launch_attn_pc<4, 2, 2>(q, st, 1);
        if (q.variant & 4) return launch_attn_pc<2, 1, 1>(q, st, 2);
        return launch_attn_pc<4, 1, 1>(q, st, 1);
    }
    return w8 ? launch_attn_t<half_t, 8, 4>(p, st) : launch_attn_t<half_t, 4, 4>(p, st);
}

// ------------------------------------------------------------------------------------------
// debug: dump what ds_read_b64_tr_b16 returns when LDS holds lds[i] = i (u16) and lane l
// passes the address of element 4*l.
// ------------------------------------------------------------------------------------------
__global__ void k_debug_tr16(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(lds + threadIdx.x * 4));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (uint16_t)v[e];
}

int launch_debug_tr16(uint16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_tr16, dim3(1), dim3(64), 0, st, out);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
