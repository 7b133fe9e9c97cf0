// Backward pass of the fusion hot path (training, exact-f32 MFMA): the adjoints of the f32-mode forward kernels.
//
// The reference trains HeteroFusion through torch.autograd (train_camera.py:163-199); here every forward kernel of
// the f32 pipeline has a hand-written adjoint:
//
//   k_gemm_tn          weight / bias gradients of the typed Linears (dW = dY^T A, reduction over the tokens of an
//                      agent map), v_mfma_f32_32x32x2_f32
//   k_layernorm_bwd    HeteroLayerNorm backward (base_transformer.py:172-177), per-type dgamma / dbeta
//   k_add_drop, k_gelu_drop, k_gelu_bwd    residual + Dropout, GELU + Dropout and their adjoints
//                      (hetero_fusion.py:65-66, base_transformer.py:186-192); the dropout mask is a pure function of
//                      (seed, salt, element index), so the backward pass regenerates it instead of storing it
//   k_attention_bwd    adjoint of k_attention (attn.hip): the probabilities are rebuilt from the saved row
//                      log-sum-exp, dS = P o (dP - rowsum(dO o O)); dQ, the gradients of the GATHERED keys / values
//                      and the gradient of the relative-position bias fragments (hetero_fusion.py:216-267)
//   k_warp_adjoint     adjoint of the bilinear key gather (warp_features, hetero_fusion.py:338-361): scatter written
//                      as a gather over the few ego pixels whose taps can touch a source pixel -> no atomics
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

// ------------------------------------------------------------------------------------------
// dropout
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// MODE 0: y = x + drop(a) (x may be null); 1: y = drop(gelu(a)); 2: y = drop'(x) * gelu'(a) (x = dh, a = pre); 3: mask
template <int MODE>
__global__ __launch_bounds__(256) void k_elementwise(const float* __restrict__ x, const float* __restrict__ a,
                                                      float* __restrict__ y, size_t n4, DropCfg d) {
    const float inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    const unsigned key = drop_key(d);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != 3) av = reinterpret_cast<const float4*>(a)[i];
        if ((MODE == 0 && x) || MODE == 2) xv = reinterpret_cast<const float4*>(x)[i];
        float s[4] = {1.f, 1.f, 1.f, 1.f};
        if (d.p > 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = drop_scale(d, key, 4 * i + e, inv_keep);
        }
        float4 r;
        if (MODE == 0) {
            r = make_float4(xv.x + s[0] * av.x, xv.y + s[1] * av.y, xv.z + s[2] * av.z, xv.w + s[3] * av.w);
        } else if (MODE == 1) {
            r = make_float4(s[0] * gelu_exact(av.x), s[1] * gelu_exact(av.y), s[2] * gelu_exact(av.z), s[3] * gelu_exact(av.w));
        } else if (MODE == 2) {
            r = make_float4(s[0] * xv.x * gelu_grad(av.x), s[1] * xv.y * gelu_grad(av.y), s[2] * xv.z * gelu_grad(av.z),
                            s[3] * xv.w * gelu_grad(av.w));
        } else {
            r = make_float4(s[0], s[1], s[2], s[3]);
        }
        reinterpret_cast<float4*>(y)[i] = r;
    }
}

template <int MODE>
static int launch_ew(const float* x, const float* a, float* y, size_t n, DropCfg d, hipStream_t st) {
    if (n == 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(n % 4 == 0, "elementwise: n=%zu must be a multiple of 4", n);
    HMVIT_CHECK_ARG(d.p >= 0.f && d.p < 1.f, "dropout p=%f out of [0, 1)", d.p);
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL((k_elementwise<MODE>), dim3(blocks), dim3(256), 0, st, x, a, y, n4, d);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
int launch_add_drop(const float* x, const float* a, float* y, size_t n, DropCfg d, hipStream_t st) { return launch_ew<0>(x, a, y, n, d, st); }
int launch_gelu_drop(const float* pre, float* h, size_t n, DropCfg d, hipStream_t st) { return launch_ew<1>(nullptr, pre, h, n, d, st); }
int launch_gelu_bwd(const float* pre, const float* dh, float* dpre, size_t n, DropCfg d, hipStream_t st) { return launch_ew<2>(dh, pre, dpre, n, d, st); }
int launch_dropout_mask(float* mask, size_t n, DropCfg d, hipStream_t st) { return launch_ew<3>(nullptr, nullptr, mask, n, d, st); }

// ------------------------------------------------------------------------------------------
// dW += dY^T A  (and dbias += column sums of dY)
// ------------------------------------------------------------------------------------------
// Tile: 128 (n) x 128 (k) of dW per workgroup over a slice of TN_ROWS tokens; 2 x 2 wavefronts of 64 x 64.  The slabs
// are staged as they lie in memory (row = token): the MFMA contraction index is the token, so both operands are read
// down LDS columns (consecutive lanes = consecutive columns: conflict-free).
constexpr int TN_BM = 32;         // tokens per slab
constexpr int TN_ROWS = 1024;     // tokens per workgroup
constexpr int TN_LS = 128;

__global__ __launch_bounds__(256) void k_gemm_tn(GemmTnJobs jobs) {
    __shared__ __attribute__((aligned(16))) float Ds[TN_BM * TN_LS];
    __shared__ __attribute__((aligned(16))) float As[TN_BM * TN_LS];
    const GemmTnJob& J = jobs.j[blockIdx.z];
    const int tiles_k = (J.K + 127) / 128, tiles_n = (J.N + 127) / 128;
    if ((int)blockIdx.x >= tiles_k * tiles_n) return;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = blockIdx.y * TN_ROWS;
    if (m_begin >= J.M) return;
    const int m_end = min(J.M, m_begin + TN_ROWS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wn = wave >> 1, wk = wave & 1, r = lane & 31, hi = lane >> 5;

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float colsum = 0.f;
    const bool do_bias = J.dbias != nullptr && tk == 0;

    for (int m0 = m_begin; m0 < m_end; m0 += TN_BM) {
        // 32 rows x 128 columns of each operand: 1024 float4, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 5, col = (c & 31) * 4;
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), av = dv;
            if (m0 + row < m_end) {
                if (n0 + col < J.N) dv = *reinterpret_cast<const float4*>(J.dy + (size_t)(m0 + row) * J.ld_dy + n0 + col);
                if (k0 + col < J.K) av = *reinterpret_cast<const float4*>(J.a + (size_t)(m0 + row) * J.ld_a + k0 + col);
            }
            *reinterpret_cast<float4*>(Ds + row * TN_LS + col) = dv;
            *reinterpret_cast<float4*>(As + row * TN_LS + col) = av;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < TN_BM / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = Ds[(2 * kk + hi) * TN_LS + wn * 64 + i * 32 + r];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = As[(2 * kk + hi) * TN_LS + wk * 64 + j * 32 + r];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int row = 0; row < TN_BM; ++row) colsum += Ds[row * TN_LS + tid];
        }
        __syncthreads();
    }
    // acc[i][j][e]: row n = (e & 3) + 8 (e >> 2) + 4 hi of block i, column k = r of block j
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wk * 64 + j * 32 + r;
            if (k >= J.K) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (n < J.N) unsafeAtomicAdd(J.dw + (size_t)n * J.K + k, acc[i][j][e]);
            }
        }
    if (do_bias && tid < 128 && n0 + tid < J.N) unsafeAtomicAdd(J.dbias + n0 + tid, colsum);
}

// The same product on the f16 pipes (split operands, fp32-class accuracy; -DHMVIT_TRAIN_EXACT_F32 keeps k_gemm_tn): the
// contraction index of v_mfma_f32_32x32x16_f16 must run along an operand's 8-half register, so the slabs are staged
// TRANSPOSED - LDS row = column of dY (or of A), 32 tokens per row as (hi, lo) planes.  A thread fetches two consecutive token
// rows of four columns and writes four 32-bit (token pair) words per plane; 16-byte blocks of a row are XOR-swizzled with
// the row index so that the 64 lanes of a store hit 32 banks twice (free) instead of 4 banks sixteen times.
constexpr int TS_BM = 32, TS_LS = 40;      // tokens per slab, halves per LDS row (32 + padding: 80-byte rows keep 16-byte alignment)

__device__ __forceinline__ int ts_addr(int row, int token) {       // half index of (row, token) in a transposed plane
    const int blk = (token >> 3) ^ ((row >> 4) & 3);
    return row * TS_LS + blk * 8 + (token & 7);
}

// Range (round 5): both operands are gradients / activations at whatever magnitude the pass has reached - behind large FFN weights
// dY grows by 1e3-1e6 on its way down and left f16's range in this kernel (round 4 detected the Inf / NaN afterwards and re-ran the
// whole backward from a lower level).  Now every 32-token slab is split at its OWN powers of two: the staging threads reduce
// max |dY| and max |A| of the slab they hold in registers (during the previous slab's products: no extra barrier), the slab is
// scaled to [2^13, 2^14) on its way into LDS, and the products accumulate at the running pair of scales; when a slab needs
// different scales (more than a few binades away - rare: magnitudes drift slowly along the token axis) the accumulator is
// multiplied by the ratio first (a power of two: exact; a second accumulator would cost the kernel its second wave per SIMD).
// Exact powers of two throughout; no pre-pass, no host read, any magnitude.
__global__ __launch_bounds__(256) void k_gemm_tn_split(GemmTnJobs jobs) {
    __shared__ __attribute__((aligned(16))) half_t Dh[128 * TS_LS], Dl[128 * TS_LS], Ah[128 * TS_LS], Al[128 * TS_LS];
    __shared__ float red[16][128];
    __shared__ float smax[2][4][2];          // [slab parity][wave][dY, A]
    const GemmTnJob& J = jobs.j[blockIdx.z];
    const int tiles_k = (J.K + 127) / 128, tiles_n = (J.N + 127) / 128;
    // XCD-aware order (round 5).  The tiles of a token slice read the same rows of dY and A - each operand column block twice at 2 x 2
    // tiles - and a workgroup's linear index modulo 8 is its XCD, each with an L2 of its own: with (tile, slice) = (blockIdx.x,
    // blockIdx.y) the four tiles of a slice sat on four XCDs and every operand came from HBM twice (61 GB fetched per cfg2 step for
    // 29 GB of operands).  With four tiles, groups of 32 consecutive workgroups now cover 8 slices x 4 tiles so that the tiles of a
    // slice are consecutive workgroups of ONE XCD (the launch pads the slice count to a multiple of 8).
    int tile = blockIdx.x, slice = blockIdx.y;
    if (gridDim.x == 4) {
        const int lid = blockIdx.x + 4 * blockIdx.y;
        tile = (lid & 31) >> 3;
        slice = (lid >> 5) * 8 + (lid & 7);
    }
    if (tile >= tiles_k * tiles_n) return;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = slice * TN_ROWS;
    if (m_begin >= J.M) return;
    const int m_end = min(J.M, m_begin + TN_ROWS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wn = wave >> 1, wk = wave & 1, r = lane & 31, hi = lane >> 5;

    float16v run[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) run[i][j][e] = 0.f;
    const bool do_bias = J.dbias != nullptr && tk == 0;
    float4 bsum[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};

    // item = tid + 256 i: token pair (item >> 5), column group item & 31 (columns 4 cg .. 4 cg + 3).  (Round 6, measured and not kept: token
    // pair fastest over the lanes - item & 15, item >> 4 - makes the 32 lanes of a ds_write_b32 group hit 32 different banks instead of 8
    // banks four times (the 65 % bank-conflict share of profiles/r06_train_pmc.txt), bit-identical results, and the kernel got SLOWER, 757 ->
    // 817 us per launch: a load then reads 16 rows x 64 bytes instead of 2 rows x 512, and the staging stores were not what it waits for.)
    float4 fd[2][2], fa[2][2];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + 256 * i, pr = item >> 5, cg = item & 31;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = m0 + 2 * pr + t;
                fd[i][t] = fa[i][t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < m_end) {
                    if (n0 + 4 * cg < J.N) fd[i][t] = *reinterpret_cast<const float4*>(J.dy + (size_t)m * J.ld_dy + n0 + 4 * cg);
                    if (k0 + 4 * cg < J.K) fa[i][t] = *reinterpret_cast<const float4*>(J.a + (size_t)m * J.ld_a + k0 + 4 * cg);
                }
            }
        }
    };
    // max |.| of the slab in this thread's registers -> the wave's partial in smax[par]
    auto publish_max = [&](int par) {
        unsigned ud = 0u, ua = 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // bit patterns with the sign shifted out order like magnitudes: one shift + half a three-input integer maximum per value
                ud = max(max(ud, __float_as_uint(fd[i][t].x) << 1), max(__float_as_uint(fd[i][t].y) << 1, max(__float_as_uint(fd[i][t].z) << 1, __float_as_uint(fd[i][t].w) << 1)));
                ua = max(max(ua, __float_as_uint(fa[i][t].x) << 1), max(__float_as_uint(fa[i][t].y) << 1, max(__float_as_uint(fa[i][t].z) << 1, __float_as_uint(fa[i][t].w) << 1)));
            }
        const float md = __uint_as_float(wave_umax(ud) >> 1), ma = __uint_as_float(wave_umax(ua) >> 1);
        if (lane == 0) { smax[par][wave][0] = md; smax[par][wave][1] = ma; }
    };
    auto put_plane = [&](half_t* ph, half_t* pl, const float4& x0, const float4& x1, float sc, int pr, int cg) {
        const float v0[4] = {x0.x * sc, x0.y * sc, x0.z * sc, x0.w * sc}, v1[4] = {x1.x * sc, x1.y * sc, x1.z * sc, x1.w * sc};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hh, ll;
            split_pk2(v0[e], v1[e], hh, ll);            // (token 2 pr, token 2 pr + 1) of column 4 cg + e
            const int o = ts_addr(4 * cg + e, 2 * pr);
            *reinterpret_cast<unsigned*>(ph + o) = hh;
            *reinterpret_cast<unsigned*>(pl + o) = ll;
        }
    };
    float sd_run = 1.f, sa_run = 1.f;        // the scales `run` is accumulated at
    bool run_live = false;
    auto rescale = [&](float f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) run[i][j][e] *= f;
    };
    fetch(m_begin);
    publish_max(0);
    __syncthreads();
    int par = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += TS_BM, par ^= 1) {
        // Optimistic staging: the slab goes into LDS at the RUNNING scales at once, while the four partial maxima published during
        // the previous slab's products are read and checked; only when the slab does not fit them (outside [2^8, 2^15): rare) it
        // is staged again at its own scales and the accumulator follows by the ratio.  (Waiting for the maxima first put an LDS
        // round trip and the scale arithmetic between the two barriers of every slab.)
        auto stage = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int item = tid + 256 * i, pr = item >> 5, cg = item & 31;
                put_plane(Dh, Dl, fd[i][0], fd[i][1], sd_run, pr, cg);
                put_plane(Ah, Al, fa[i][0], fa[i][1], sa_run, pr, cg);
            }
        };
        float md = 0.f, ma = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { md = fmaxf(md, smax[par][w][0]); ma = fmaxf(ma, smax[par][w][1]); }
        if (run_live) stage();
        // keep the running scales while the slab stays inside [2^8, 2^15) with them (top of the f16 range, low halves normal)
        const float td = md * sd_run, ta = ma * sa_run;
        const bool keep = run_live && td < 32768.f && ta < 32768.f && (td >= 256.f || md == 0.f) && (ta >= 256.f || ma == 0.f);
        if (!keep) {
            const float sd_need = pow2_scale(md), sa_need = pow2_scale(ma);
            // in two steps: each ratio is a power of two within 2^+-80, their product can leave f32 (2^+-160) and an infinite factor would
            // turn an accumulator that holds 0 into NaN
            if (run_live) { rescale(sd_need * pow2_inv(sd_run)); rescale(sa_need * pow2_inv(sa_run)); }
            sd_run = sd_need; sa_run = sa_need; run_live = true;
            stage();                                         // (same addresses, same threads: plain overwrite)
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bsum[i].x += fd[i][0].x + fd[i][1].x; bsum[i].y += fd[i][0].y + fd[i][1].y;
                bsum[i].z += fd[i][0].z + fd[i][1].z; bsum[i].w += fd[i][0].w + fd[i][1].w;
            }
        }
        __syncthreads();
        const bool more = m0 + TS_BM < m_end;
        if (more) fetch(m0 + TS_BM);        // the next slab travels during the products
#pragma unroll
        for (int kk = 0; kk < TS_BM / 16; ++kk) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = ts_addr(wn * 64 + i * 32 + r, kk * 16 + hi * 8);
                ah[i] = *reinterpret_cast<const half8*>(Dh + o);
                al[i] = *reinterpret_cast<const half8*>(Dl + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = ts_addr(wk * 64 + j * 32 + r, kk * 16 + hi * 8);
                bh[j] = *reinterpret_cast<const half8*>(Ah + o);
                bl[j] = *reinterpret_cast<const half8*>(Al + o);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    run[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], run[i][j], 0, 0, 0);
                    run[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], run[i][j], 0, 0, 0);
                    run[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], run[i][j], 0, 0, 0);
                }
        }
        if (more) publish_max(par ^ 1);     // (the other parity's partials were last read before this iteration's first barrier)
        __syncthreads();
    }
    const float inv_run = pow2_inv(sd_run) * pow2_inv(sa_run);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wk * 64 + j * 32 + r;
            if (k >= J.K) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (n < J.N) unsafeAtomicAdd(J.dw + (size_t)n * J.K + k, run[i][j][e] * inv_run);
            }
        }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + 256 * i, pr = item >> 5, cg = item & 31;
            *reinterpret_cast<float4*>(&red[pr][4 * cg]) = bsum[i];
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < J.N) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += red[q][tid];
            unsafeAtomicAdd(J.dbias + n0 + tid, sum);
        }
    }
}

int launch_gemm_tn(const GemmTnJobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return HMVIT_OK;
    int max_tiles = 0, max_slices = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const GemmTnJob& j = jobs.j[i];
        HMVIT_CHECK_ARG(j.N % 4 == 0 && j.K % 4 == 0 && j.ld_dy % 4 == 0 && j.ld_a % 4 == 0, "gemm_tn: N=%d K=%d ld=%d/%d must be multiples of 4",
                        j.N, j.K, j.ld_dy, j.ld_a);
        max_tiles = max(max_tiles, cdiv(j.N, 128) * cdiv(j.K, 128));
        max_slices = max(max_slices, cdiv(j.M, TN_ROWS));
    }
    if (max_tiles == 0 || max_slices == 0) return HMVIT_OK;
    if (max_tiles == 4) max_slices = (max_slices + 7) / 8 * 8;      // k_gemm_tn_split's XCD-aware order covers slices in eights
#ifdef HMVIT_TRAIN_EXACT_F32
    hipLaunchKernelGGL(k_gemm_tn, dim3(max_tiles, max_slices, jobs.n), dim3(256), 0, st, jobs);
#else
    hipLaunchKernelGGL(k_gemm_tn_split, dim3(max_tiles, max_slices, jobs.n), dim3(256), 0, st, jobs);
#endif
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// BatchNorm2d in training mode (batch statistics) fused with ReLU, on NHWC maps viewed as (M = N H W, C): the normalisation of
// the detection tail's convolution blocks when the model trains (naive_decoder.py:45-54 under nn.Module.train()).
//   forward   sums[c] += sum_m x, sums[C + c] += sum_m x^2          (k_bn_reduce<0>)
//             y = relu(gamma (x - mean) rstd + beta)                 (k_bn_apply<0>)
//   backward  g = dy [y > 0];  sums[c] += sum g, sums[C + c] += sum g xhat          (k_bn_reduce<1>)
//             dx = gamma rstd (g - sums[c] / M - xhat sums[C + c] / M)              (k_bn_apply<1>)
// Blocking: 64 rows x 256 channels per pass (thread (q = tid & 63, rg = tid >> 6) owns columns 4 q .. 4 q + 3 of rows rg, rg + 4, ...),
// rows per workgroup scaled to the launch; the four row groups meet in LDS, one atomic add per column and workgroup.
// ------------------------------------------------------------------------------------------

template <int BWD>
__global__ __launch_bounds__(256) void k_bn_reduce(BnArgs a, int rows_per_wg) {
    __shared__ float4 red[2][4][64];
    const int q = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + 4 * q;
    const int m_begin = blockIdx.y * rows_per_wg, m_end = min(a.M, m_begin + rows_per_wg);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < a.C) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
        if (BWD) {
            const float4 m4 = *reinterpret_cast<const float4*>(a.mean + c), r4 = *reinterpret_cast<const float4*>(a.rstd + c);
            mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w; rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
        } else if (a.mean) {   // forward, second pass: sums of (x - pivot) and (x - pivot)^2 (no cancellation in the variance)
            const float4 m4 = *reinterpret_cast<const float4*>(a.mean + c);
            mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
        }
        for (int m = m_begin + rg; m < m_end; m += 4) {
            const size_t o = (size_t)m * a.C + c;
            const float4 xv = *reinterpret_cast<const float4*>(a.x + o);
            const float x[4] = {xv.x, xv.y, xv.z, xv.w};
            if (BWD) {
                const float4 dv = *reinterpret_cast<const float4*>(a.dy + o);
                float g[4] = {dv.x, dv.y, dv.z, dv.w};
                if (a.relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(a.y + o);
                    const float y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += g[e]; s1[e] = fmaf(g[e], (x[e] - mu[e]) * rs[e], s1[e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = x[e] - mu[e]; s0[e] += d; s1[e] = fmaf(d, d, s1[e]); }
            }
        }
    }
    red[0][rg][q] = make_float4(s0[0], s0[1], s0[2], s0[3]);
    red[1][rg][q] = make_float4(s1[0], s1[1], s1[2], s1[3]);
    __syncthreads();
    if (rg < 2 && c < a.C) {
        const float4 p0 = red[rg][0][q], p1 = red[rg][1][q], p2 = red[rg][2][q], p3 = red[rg][3][q];
        float* o = a.out + rg * a.C + c;
        unsafeAtomicAdd(o + 0, (p0.x + p1.x) + (p2.x + p3.x));
        unsafeAtomicAdd(o + 1, (p0.y + p1.y) + (p2.y + p3.y));
        unsafeAtomicAdd(o + 2, (p0.z + p1.z) + (p2.z + p3.z));
        unsafeAtomicAdd(o + 3, (p0.w + p1.w) + (p2.w + p3.w));
    }
}

template <int BWD>
__global__ __launch_bounds__(256) void k_bn_apply(BnArgs a) {
    const size_t n4 = (size_t)a.M * a.C / 4;
    const float inv_m = 1.f / (float)a.M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % (size_t)a.C);
        const float4 xv = reinterpret_cast<const float4*>(a.x)[i];
        const float4 m4 = *reinterpret_cast<const float4*>(a.mean + c), r4 = *reinterpret_cast<const float4*>(a.rstd + c);
        const float4 g4 = *reinterpret_cast<const float4*>(a.gamma + c);
        const float x[4] = {xv.x, xv.y, xv.z, xv.w}, mu[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
        const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
        float r[4];
        if (BWD) {
            const float4 dv = reinterpret_cast<const float4*>(a.dy)[i];
            float g[4] = {dv.x, dv.y, dv.z, dv.w};
            if (a.relu) {
                const float4 yv = reinterpret_cast<const float4*>(a.y)[i];
                const float y[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
            }
            const float4 sb = *reinterpret_cast<const float4*>(a.sums + c), sg = *reinterpret_cast<const float4*>(a.sums + a.C + c);
            const float db[4] = {sb.x, sb.y, sb.z, sb.w}, dg[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ga[e] * rs[e] * (g[e] - db[e] * inv_m - (x[e] - mu[e]) * rs[e] * dg[e] * inv_m);
        } else {
            const float4 b4 = *reinterpret_cast<const float4*>(a.beta + c);
            const float be[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[e] = fmaf((x[e] - mu[e]) * rs[e], ga[e], be[e]);
                if (a.relu) r[e] = fmaxf(r[e], 0.f);
            }
        }
        reinterpret_cast<float4*>(a.out)[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

int launch_bn(const BnArgs& a, int bwd, int apply, hipStream_t st) {
    if (a.M <= 0 || a.C <= 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(a.C % 4 == 0, "batch norm: C=%d must be a multiple of 4", a.C);
    if (apply) {
        const size_t n4 = (size_t)a.M * a.C / 4;
        const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
        if (bwd) hipLaunchKernelGGL((k_bn_apply<1>), dim3(blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_bn_apply<0>), dim3(blocks), dim3(256), 0, st, a);
    } else {
        const int col_tiles = cdiv(a.C, 256);
        long long want = (long long)a.M * col_tiles / 2048;
        int rows = (int)((want + 63) / 64) * 64;
        rows = rows < 64 ? 64 : (rows > 8192 ? 8192 : rows);
        if (bwd) hipLaunchKernelGGL((k_bn_reduce<1>), dim3(col_tiles, cdiv(a.M, rows)), dim3(256), 0, st, a, rows);
        else hipLaunchKernelGGL((k_bn_reduce<0>), dim3(col_tiles, cdiv(a.M, rows)), dim3(256), 0, st, a, rows);
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward
// ------------------------------------------------------------------------------------------
constexpr int LNB_TOK = 16, LNB_B = 4;   // tokens per wavefront, tokens in flight

// sum over the 64 lanes of a wavefront, in every lane: four DPP steps inside the rows of 16 lanes, two row swaps across them
// (as wave_umax, common.hpp; six ds_bpermute round trips with __shfl_xor)
__device__ __forceinline__ float wave_sum_dpp(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
    x += dpp(x, std::integral_constant<int, 0x141>{});      // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});      // row_mirror
    return xor32_sum(xor16_sum(x));
}

template <int VPL>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ gamma, AgentTypes types,
                                                        const float* dres, float* dx, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int P) {
    constexpr int C = VPL * 64;
    __shared__ float red[2][4][C];
    const int agent = blockIdx.y, t = types.t[agent];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float g[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g[i] = gamma[t * C + lane * VPL + i];
        dg[i] = db[i] = 0.f;
    }
    // LNB_B tokens at a time: their rows are requested together and their four wave sums run side by side (a token at a time the
    // wavefront waited for one row, then for 24 dependent ds_bpermute round trips: WAIT_ANY 79 %, 4.4 TB/s in the round-5 counters);
    // the sums themselves on DPP / permlane swaps (wave_sum_dpp)
    const int tok0 = (blockIdx.x * 4 + wave) * LNB_TOK;
    for (int tt = 0; tt < LNB_TOK; tt += LNB_B) {
        if (tok0 + tt >= P) break;
        float v[LNB_B][VPL], d[LNB_B][VPL], r[LNB_B][VPL];
        size_t base[LNB_B];
        bool on[LNB_B];
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) {
            const int tok = tok0 + tt + k;
            on[k] = tok < P;                                        // wave-uniform
            base[k] = ((size_t)agent * P + (on[k] ? tok : P - 1)) * C + lane * VPL;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { v[k][i] = x[base[k] + i]; d[k][i] = dy[base[k] + i]; r[k][i] = dres ? dres[base[k] + i] : 0.f; }
        }
        float mean[LNB_B], rstd[LNB_B], s1[LNB_B], s2[LNB_B];
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) s += v[k][i];
            mean[k] = s;
        }
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) mean[k] = wave_sum_dpp(mean[k]) * (1.f / C);
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { v[k][i] -= mean[k]; q += v[k][i] * v[k][i]; }
            rstd[k] = q;
        }
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) rstd[k] = rsqrtf(wave_sum_dpp(rstd[k]) * (1.f / C) + 1e-5f);
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                v[k][i] *= rstd[k];                 // xhat
                const float gd = g[i] * d[k][i];
                a1 += gd;
                a2 += gd * v[k][i];
                if (on[k]) { dg[i] += d[k][i] * v[k][i]; db[i] += d[k][i]; }
            }
            s1[k] = a1; s2[k] = a2;
        }
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) { s1[k] = wave_sum_dpp(s1[k]) * (1.f / C); s2[k] = wave_sum_dpp(s2[k]) * (1.f / C); }
#pragma unroll
        for (int k = 0; k < LNB_B; ++k) {
            if (on[k]) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) dx[base[k] + i] = rstd[k] * (g[i] * d[k][i] - s1[k] - v[k][i] * s2[k]) + r[k][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        red[0][wave][lane * VPL + i] = dg[i];
        red[1][wave][lane * VPL + i] = db[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        unsafeAtomicAdd(dgamma + t * C + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        unsafeAtomicAdd(dbeta + t * C + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

int launch_layernorm_bwd(const float* x, const float* dy, const float* gamma, const AgentTypes& types, int n_agents,
                         const float* dres, float* dx, float* dgamma, float* dbeta, int P, int C, hipStream_t st) {
    if (n_agents <= 0) return HMVIT_OK;
    dim3 grid(cdiv(P, 4 * LNB_TOK), n_agents);
    switch (C) {
        case 64: hipLaunchKernelGGL((k_layernorm_bwd<1>), grid, dim3(256), 0, st, x, dy, gamma, types, dres, dx, dgamma, dbeta, P); break;
        case 128: hipLaunchKernelGGL((k_layernorm_bwd<2>), grid, dim3(256), 0, st, x, dy, gamma, types, dres, dx, dgamma, dbeta, P); break;
        case 256: hipLaunchKernelGGL((k_layernorm_bwd<4>), grid, dim3(256), 0, st, x, dy, gamma, types, dres, dx, dgamma, dbeta, P); break;
        default: set_error("layernorm_bwd: C=%d unsupported (64, 128, 256)", C); return HMVIT_EINVAL;
    }
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// a-priori bound on |V'| (and |O|) of a stage from its weights: launch_v_bound (kernels.hpp)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_v_bound(const float* __restrict__ w_kv, const float* __restrict__ b_kv, const float* __restrict__ g,
                                                 const float* __restrict__ be, int n_pairs, int n_types, int C, float* out) {
    __shared__ float red[2][4];
    // ||LayerNorm(x)||_2 <= sqrt(C) (max|gamma| + max|beta|): the normalised row has 2-norm sqrt(C) exactly
    float gm = 0.f, bm = 0.f;
    for (int i = threadIdx.x; i < n_types * C; i += 256) { gm = fmaxf(gm, fabsf(g[i])); bm = fmaxf(bm, fabsf(be[i])); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, o, 64)); bm = fmaxf(bm, __shfl_xor(bm, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = gm; red[1][threadIdx.x >> 6] = bm; }
    __syncthreads();
    gm = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    bm = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const float xn2 = sqrtf((float)C) * (gm + bm);
    // one (V' matrix, head) block of 32 rows x C per wave at a time: Frobenius norm (>= the spectral norm) and the bias slice's 2-norm
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, heads = C / 32;
    float best = 0.f;
    for (int blk = blockIdx.x * 4 + wave; blk < n_pairs * heads; blk += gridDim.x * 4) {
        const int pair = blk / heads, h = blk - pair * heads;
        const float* w = w_kv + ((size_t)(pair * 2 + 1) * C + h * 32) * C;
        float f2 = 0.f;
        for (int k = lane; k < 32 * C; k += 64) f2 = fmaf(w[k], w[k], f2);
        const float bv = lane < 32 ? b_kv[(size_t)(pair * 2 + 1) * C + h * 32 + lane] : 0.f;
        float b2 = bv * bv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { f2 += __shfl_xor(f2, o, 64); b2 += __shfl_xor(b2, o, 64); }
        best = fmaxf(best, fmaf(sqrtf(f2), xn2, sqrtf(b2)) * 1.0001f);
    }
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(best));      // non-negative floats order like their bits
}

int launch_v_bound(const float* w_kv, const float* b_kv, const float* ln_gamma, const float* ln_beta, int n_pairs, int n_types, int C,
                   float* out, hipStream_t st) {
    HMVIT_CHECK_ARG(w_kv && b_kv && ln_gamma && ln_beta && out, "v_bound: null pointer%s", "");
    HMVIT_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
    hipLaunchKernelGGL(k_v_bound, dim3(64), dim3(256), 0, st, w_kv, b_kv, ln_gamma, ln_beta, n_pairs, n_types, C, out);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// attention backward
// ------------------------------------------------------------------------------------------
// Same decomposition as k_attention<float>: one workgroup per (sample, ego, window, group of HG heads), one wavefront
// per head, keys in chunks of 64, K / V tiles re-gathered into LDS.  MFMA tile algebra (v_mfma_f32_16x16x4_f32: a lane
// (l = lane & 15, g = lane >> 4) supplies A[row l][k g] and B[k g][col l] and receives D[row 4g + r][col l]): an
// accumulator tile can only be contracted over its ROW index, so both orientations of the logits are formed:
//   T tiles  S^T[key][query]   -> contraction over keys:    dQ^T = K^T dS^T
//   N tiles  S[query][key]     -> contraction over queries: dV = P^T dO, dK = dS^T Q
// The probabilities come from the saved log-sum-exp: P = exp(S + mask - lse), dS = P o (dP - D), D = rowsum(dO o O).
// Two wavefronts per head: each takes two of the four 16-key tiles of a chunk (its dK / dV tiles are complete, its dQ is a partial
// sum over its keys and meets the other wave's in LDS once at the end).  With one wave per head a workgroup had two waves and a
// CU four - one per SIMD, so the gather of a chunk and its products never overlapped.
// Occupancy (round 3).  Left to itself hipcc allocated this kernel 256 VGPRs + 96 AGPRs - ONE wavefront per SIMD, four per CU - and
// with `__launch_bounds__(.., 2)` it spilled.  The registers were not operands: (i) every token's store address, precomputed
// outside the chunk loop and kept alive (cured by taking the lane coordinates through an opaque asm copy where the addresses are
// formed); (ii) the fragments of all four query tiles of both orientations at once, the Q / dO tiles being loop-invariant (cured by
// two passes over the query tiles, each tile pair behind a compiler fence).  Now 254 VGPRs, no AGPRs, no scratch: two wavefronts per
// SIMD = two workgroups per CU, 7.3 -> 5.3 ms per launch at cfg2.  Two workgroups per CU also exposed the divergent gather (below).
// (History: an earlier `(.., 2)` build that spilled 8 VGPRs was not reproducible from run to run - tools/probe/r03_grad_bisect.sh -
// which was blamed on the spills; it was the same divergent gather.)
#ifdef HMVIT_PROBE
// cycle stamps of one workgroup (lane 0 of every wave): [wave][0 start, 1 prologue done, 2 + 3 c gather issued+staged, 3 + 3 c gather
// barrier passed, 4 + 3 c products + stores done, ..., 30 dq stored]; hmvit_debug_bwd_trace (tools/probe/bwd_trace.py)
__device__ unsigned long long g_bwd_trace[8 * 32];
#define BWD_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (blockIdx.x == 2001 && blockIdx.y == 1 && (threadIdx.x & 63) == 0 && (slot) < 32)         \
            g_bwd_trace[(threadIdx.x >> 6) * 32 + (slot)] = __builtin_readcyclecounter();            \
    } while (0)
int debug_bwd_trace(unsigned long long* host, int n) {
    HMVIT_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_trace), sizeof(unsigned long long) * (n < 256 ? n : 256)));
    return HMVIT_OK;
}
#else
#define BWD_STAMP(slot) do {} while (0)
#endif

template <int WIN, int HG>
__global__ __launch_bounds__(HG * 128, 2) void k_attention_bwd(AttnBwdParams bp) {
    const AttnParams& p = bp.f;
    constexpr int N = WIN * WIN, NQT = N / 16, SPC = 64 / N, NB = (WIN == 8) ? 7 : 1;
    constexpr int CH = HG * 32, QS = CH + 8, KS = CH + 8;             // halves per LDS row (16-byte aligned, conflict-free b128 reads)
    constexpr int THREADS = HG * 128;
    constexpr int TPK = CH / 8, KPP = THREADS / TPK;
    // every tile is kept as (hi, lo) f16 halves, split ONCE by the thread that stages it (round 3; the f32 tiles of round 2 were
    // re-split by every wave for every key tile: the kernel was bound by those conversions, not by its matrix products).  Operands
    // along the channels are 16-byte row reads; operands along the rows (the 16 x 16 x 16 products over keys / queries) come out
    // of the same row-major tiles through ds_read_b64_tr_b16.
    __shared__ __attribute__((aligned(16))) half_t Qh[N * QS], Ql[N * QS], dOh[N * QS], dOl[N * QS];
    __shared__ __attribute__((aligned(16))) half_t KVs[4 * 64 * KS];
    half_t *Kh = KVs, *Kl = KVs + 64 * KS, *Vh = KVs + 2 * 64 * KS, *Vl = KVs + 3 * 64 * KS;
    static_assert(sizeof(KVs) >= (size_t)HG * NQT * 2 * 64 * 16, "the dQ exchange at the end reuses the key / value tiles");
    __shared__ __attribute__((aligned(16))) float maskadd[64];
    __shared__ float Dl[N][HG], Lse[N][HG];

    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / WIN, Y = W / WIN, NG = C / CH, heads = C / 32;
    const int win = blockIdx.x / NG, hg = blockIdx.x - win * NG;
    const int ego = blockIdx.y, b = blockIdx.z;
    const int wx = win / Y, wy = win - wx * Y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) % HG, khalf = (tid >> 6) / HG;       // head of the group, half of the key tiles
    const int lq = lane & 15, g = lane >> 4;
    const int head = hg * HG + wave;
    const int te = p.mode[b * L + ego], ev = p.ego_e[b * L + ego];
    const int ch0 = hg * CH;
    const float* qplane = reinterpret_cast<const float*>(p.q) + (size_t)(b * L + ego) * P * C;
    const float* oplane = reinterpret_cast<const float*>(p.out) + (size_t)(b * L + ego) * P * C;
    const float* doplane = bp.d_out + (size_t)(b * L + ego) * P * C;
    const float* kvplanes = reinterpret_cast<const float*>(p.kv);

    BWD_STAMP(0);
    // ---- range of the gradient operands (round 5: no detect-and-retry) ----
    // dO arrives at whatever magnitude the pass has reached.  The workgroup brings ITS 64 x CH tile of dO to rho sigma in [T, 2T) with
    // powers of two sigma and T - rho = the largest 2-norm of a token's 32 channels of one head - and multiplies its results by
    // 1 / sigma at the stores (the pass is linear in dO).  T comes from the stage's a-priori bound vb on the 2-norm of a head's slice
    // of any V' row, hence of any O row (launch_v_bound: weights only): by Cauchy-Schwarz |dP| <= vb rho and |D| <= vb rho, so with
    // T <= 2^13 / vb every dS = P o (dP - D) - the one DERIVED f16 operand - stays below 2^15 whatever the data.  T is capped at 2^9
    // (the level the whole pass ran at before) and floored at 2^-10; with the shipped initialisation vb ~ 60 and T = 2^7.
    float do_scale = 1.f;
    {
        float T = 512.f;
        if (bp.v_bound) {
            const float vb = fmaxf(*bp.v_bound, 1e-30f);
            // power of two <= 2^13 / vb: exponent field arithmetic, then the clamps
            const float lim = __uint_as_float(__float_as_uint(8192.f / vb) & 0x7f800000u);
            T = fminf(512.f, fmaxf(lim, 0.0009765625f));
        }
        __shared__ float domax[THREADS / 64];
        const int cl = (tid % TPK) * 8;
        float m = 0.f;
        for (int n = tid / TPK; n < N; n += KPP) {
            int row, col;
            token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
            const float4* s4 = reinterpret_cast<const float4*>(doplane + (size_t)(row * W + col) * C + ch0 + cl);
            const float4 a = s4[0], c4 = s4[1];
            // this thread's 8 channels; the 4 threads of a head (lanes l, l ^ 1, l ^ 2, l ^ 3) complete the head's sum of squares.
            // Scaled by the thread's own largest element first: squares of 1e-25 or 1e25 must not leave f32's range
            const float am = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                   fmaxf(fmaxf(fabsf(c4.x), fabsf(c4.y)), fmaxf(fabsf(c4.z), fabsf(c4.w))));
            float hm = fmaxf(am, __shfl_xor(am, 1, 64));
            hm = fmaxf(hm, __shfl_xor(hm, 2, 64));
            const float sc = pow2_scale(hm);                        // the head slice's largest element -> [2^13, 2^14)
            float q2 = 0.f;
            q2 = fmaf(a.x * sc, a.x * sc, q2); q2 = fmaf(a.y * sc, a.y * sc, q2); q2 = fmaf(a.z * sc, a.z * sc, q2); q2 = fmaf(a.w * sc, a.w * sc, q2);
            q2 = fmaf(c4.x * sc, c4.x * sc, q2); q2 = fmaf(c4.y * sc, c4.y * sc, q2); q2 = fmaf(c4.z * sc, c4.z * sc, q2); q2 = fmaf(c4.w * sc, c4.w * sc, q2);
            q2 += __shfl_xor(q2, 1, 64);
            q2 += __shfl_xor(q2, 2, 64);
            m = fmaxf(m, sqrtf(q2) * pow2_inv(sc) * 1.0001f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) domax[tid >> 6] = m;
        __syncthreads();
        m = domax[0];
#pragma unroll
        for (int w = 1; w < THREADS / 64; ++w) m = fmaxf(m, domax[w]);
        // m = rho; pow2_scale brings it to [2^13, 2^14); T / 2^13 takes it to [T, 2T)
        do_scale = pow2_scale(m) * (T * (1.f / 8192.f));
    }
    const float do_inv = pow2_inv(do_scale);
    // ---- query tile (+ bias), dO tile, D and lse ----
    // D = rowsum over a head's 32 channels of dO o O comes out of the same pass: the thread that stages 8 channels of a token's dO
    // multiplies them with the 8 channels of O and the four threads of a head add up (round 5; a loop of its own walked 64
    // scalar loads per (token, head) with half the workgroup idle: 7 % of the kernel's cycles)
    {
        const float* bq = p.b_q + te * C + ch0;
        const int cl = (tid % TPK) * 8;
        for (int n = tid / TPK; n < N; n += KPP) {
            int row, col;
            token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
            const size_t o = (size_t)(row * W + col) * C + ch0 + cl;
            half8 qh_, ql_, dh_, dl_;
            float qv[8], dv8[8];
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qv[e] = qplane[o + e] + bq[cl + e];
                const float dov = doplane[o + e];
                dv8[e] = dov * do_scale;
#ifdef HMVIT_PROBE
                if (bp.probe != 3)
#endif
                d = fmaf(dov, oplane[o + e], d);
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if ((tid & 3) == 0) {
                const int hh = cl >> 5;
                Dl[n][hh] = d * do_scale;
                // kept times log2(e): the probabilities are rebuilt with v_exp_f32 (a base-2 exponential)
                Lse[n][hh] = p.lse[((size_t)(b * L + ego) * P + row * W + col) * heads + hg * HG + hh] * 1.4426950408889634f;
            }
            split_pk8(qv, qh_, ql_);
            split_pk8(dv8, dh_, dl_);
            *reinterpret_cast<half8*>(Qh + n * QS + cl) = qh_;
            *reinterpret_cast<half8*>(Ql + n * QS + cl) = ql_;
            *reinterpret_cast<half8*>(dOh + n * QS + cl) = dh_;
            *reinterpret_cast<half8*>(dOl + n * QS + cl) = dl_;
        }
    }
    // (the bias fragments themselves are re-read from global per tile pair: 14 fragment registers sets spilled the kernel)
    // bias-gradient accumulators: a wave's two key tiles (kt = 2 khalf, 2 khalf + 1) meet only NBW = 5 of the 7 tile offsets qt - kt + 3
    // (window 8): offsets [2, 6] for khalf 0, [0, 4] for khalf 1 - eight registers less than all seven (and with them the kernel
    // stays inside 256 registers at two waves per SIMD without spilling)
    constexpr int NBW = (WIN == 8) ? 5 : 1;
    float4v dbias[NBW];
    const float* biasT_g = p.bias_frag + ((size_t)head * NB * 64 + lane) * 4;
    const float* biasN_g = bp.bias_frag_neg + ((size_t)head * NB * 64 + lane) * 4;
#pragma unroll
    for (int v = 0; v < NBW; ++v) {
        dbias[v] = (float4v)(0.f);
    }
    float4v dq_acc[NQT][2];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq_acc[qt][dt] = (float4v)(0.f);
    __syncthreads();
    BWD_STAMP(1);

    const int hoff = wave * 32;
    const int n_chunks = (p.n_src + SPC - 1) / SPC;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        // ---- gather 64 keys x CH channels of K and V (as the forward does) ----
        int any_visible = 0;
        {
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            const int cl = (tid_o % TPK) * 8;
#pragma unroll 1
            for (int pass = 0; pass < 64 / KPP; ++pass) {
                const int kk = pass * KPP + tid_o / TPK;
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
                    if (ident) {
                        t.roi = 1.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { t.idx[k] = row * W + col; t.w[k] = k == 0 ? 1.f : 0.f; }
                    } else {
                        t = make_taps(a, col, row, H, W);
                    }
                    visible = (t.roi != 0.f) && (p.cav[b * L + src] != 0);
                    if (__any(visible)) {     // wave-uniform: a wave none of whose keys is visible loads nothing
                        // Every lane loads (pixel 0 where the key is invisible) and the result is selected afterwards: NO divergent region.
                        // With `if (visible) { loads }` the kernel was not reproducible at two workgroups per CU - one backward pass in
                        // two differed in keys at the edge of a source's field of view, where visibility differs between the lanes of a
                        // wave (tools/probe/bwd_repro.py, r03_bwd_occ.sh; the same source at one workgroup per CU was reproducible).
                        const int ts = p.mode[b * L + src];
                        const float* kpl = kvplanes + ((size_t)((b * L + src) * p.E + ev) * 2) * P * C + ch0 + cl;
                        const float* bk = p.b_kv + (size_t)(te * HMVIT_NUM_TYPES + ts) * 2 * C + ch0 + cl;
                        int ix[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) ix[k] = visible ? t.idx[k] : 0;
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            float acc[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float4* s4 = reinterpret_cast<const float4*>(kpl + (size_t)pl * P * C + (size_t)ix[k] * C);
                                const float4 r0 = s4[0], r1 = s4[1];
                                const float wk = t.w[k];
                                acc[0] = fmaf(wk, r0.x, acc[0]); acc[1] = fmaf(wk, r0.y, acc[1]); acc[2] = fmaf(wk, r0.z, acc[2]); acc[3] = fmaf(wk, r0.w, acc[3]);
                                acc[4] = fmaf(wk, r1.x, acc[4]); acc[5] = fmaf(wk, r1.y, acc[5]); acc[6] = fmaf(wk, r1.z, acc[6]); acc[7] = fmaf(wk, r1.w, acc[7]);
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) kvv[pl][e] = visible ? acc[e] + bk[pl * C + e] : 0.f;
                        }
                    }
                }
                half8 kh_, kl_, vh_, vl_;
                split_pk8(kvv[0], kh_, kl_);
                split_pk8(kvv[1], vh_, vl_);
                *reinterpret_cast<half8*>(Kh + kk * KS + cl) = kh_;
                *reinterpret_cast<half8*>(Kl + kk * KS + cl) = kl_;
                *reinterpret_cast<half8*>(Vh + kk * KS + cl) = vh_;
                *reinterpret_cast<half8*>(Vl + kk * KS + cl) = vl_;
                if (cl == 0) maskadd[kk] = visible ? 0.f : -INFINITY;
                any_visible |= visible ? 1 : 0;
            }
        }
        BWD_STAMP(2 + 3 * chunk);
        any_visible = __syncthreads_or(any_visible);
        BWD_STAMP(3 + 3 * chunk);
#ifdef HMVIT_PROBE
        if (bp.probe == 1) any_visible = 0;
#endif

        if (any_visible) {
            auto split4 = [](float a, float b, float c, float d, half4& h, half4& l) { split_pk4(a, b, c, d, h, l); };
            // rows row0 .. row0 + 3, column col0 + lq of a row-major f16 tile as one operand of v_mfma_f32_16x16x16_f16
            // (ds_read_b64_tr_b16: lane l of a 16-lane group points at row row0 + (l >> 2), columns col0 + 4 (l & 3) .. + 3 and
            // receives column l of the 4 x 16 block; tests/test_hip_ops.py::test_tr16_lane_mapping)
            auto col4 = [&](const half_t* tile, int stride, int row0, int col0) {
                const half_t* a = tile + (row0 + (lq >> 2)) * stride + col0 + 4 * (lq & 3);
                const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
                return half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            };
            constexpr float LOG2E = 1.4426950408889634f;
            // (the key tile index stays a compile-time constant: it selects the bias fragment)
            auto key_tile = [&](auto kt_c) {
                constexpr int kt = decltype(kt_c)::value;
                // operands of key tile kt: rows (kt*16 + lq) x channels 8 g .. 8 g + 7, and the "row 4g + r, column lq" form of K.
                // The four products over the 32 channels of a head (S^T, dP^T, S, dP) run on split-f16 operands (three
                // v_mfma_f32_16x16x32_f16 instead of eight v_mfma_f32_16x16x4_f32: lane (l, g) supplies channels 8 g .. 8 g + 7 of
                // row l, the accumulator layout is the same), and so do the products over the 16 keys / queries of a tile (v_mfma_f32_16x16x16_f16).
                const half8 kh = *reinterpret_cast<const half8*>(Kh + (kt * 16 + lq) * KS + hoff + 8 * g);
                const half8 kl = *reinterpret_cast<const half8*>(Kl + (kt * 16 + lq) * KS + hoff + 8 * g);
                const half8 vh = *reinterpret_cast<const half8*>(Vh + (kt * 16 + lq) * KS + hoff + 8 * g);
                const half8 vl = *reinterpret_cast<const half8*>(Vl + (kt * 16 + lq) * KS + hoff + 8 * g);
                half4 kdh[2], kdl[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    kdh[dt] = col4(Kh, KS, kt * 16 + 4 * g, hoff + dt * 16);
                    kdl[dt] = col4(Kl, KS, kt * 16 + 4 * g, hoff + dt * 16);
                }
                const float4v maddT = *reinterpret_cast<const float4v*>(maskadd + kt * 16 + 4 * g);
                const float maddN = maskadd[kt * 16 + lq];
                float4v dk_acc[2], dv_acc[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dk_acc[dt] = dv_acc[dt] = (float4v)(0.f);

#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) {
                    asm volatile("" ::: "memory");
                    const half8 qh = *reinterpret_cast<const half8*>(Qh + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 ql = *reinterpret_cast<const half8*>(Ql + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 doh = *reinterpret_cast<const half8*>(dOh + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 dol = *reinterpret_cast<const half8*>(dOl + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const int bvT = (WIN == 8) ? (qt - kt + 3) : 0, bvN = (WIN == 8) ? (kt - qt + 3) : 0;
                    // ---- T orientation: rows = keys 4g + r, column = query lq ----
                    float4v sT = *reinterpret_cast<const float4v*>(biasT_g + bvT * 256), dpT = (float4v)(0.f);
                    sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh, sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql, sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh, sT, 0, 0, 0);
                    dpT = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, doh, dpT, 0, 0, 0);
                    dpT = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, dol, dpT, 0, 0, 0);
                    dpT = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, doh, dpT, 0, 0, 0);
                    const float lseT = Lse[qt * 16 + lq][wave], dT = Dl[qt * 16 + lq][wave];
                    float4v dsT;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dsT[r] = __builtin_amdgcn_exp2f(fmaf(sT[r], LOG2E, maddT[r] - lseT)) * (dpT[r] - dT);
                    dbias[(WIN == 8) ? bvT - (kt < 2 ? 2 : 0) : 0] += dsT;
                    {   // dQ^T += K^T dS^T over this tile's 16 keys: the lane's four values ARE the operand of v_mfma_f32_16x16x16_f16
                        half4 dsh, dsl;
                        split4(dsT[0], dsT[1], dsT[2], dsT[3], dsh, dsl);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            dq_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kdl[dt], dsh, dq_acc[qt][dt], 0, 0, 0);
                            dq_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kdh[dt], dsl, dq_acc[qt][dt], 0, 0, 0);
                            dq_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kdh[dt], dsh, dq_acc[qt][dt], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) {
                    asm volatile("" ::: "memory");
                    const half8 qh = *reinterpret_cast<const half8*>(Qh + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 ql = *reinterpret_cast<const half8*>(Ql + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 doh = *reinterpret_cast<const half8*>(dOh + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const half8 dol = *reinterpret_cast<const half8*>(dOl + (qt * 16 + lq) * QS + hoff + 8 * g);
                    const int bvN = (WIN == 8) ? (kt - qt + 3) : 0;
                    // ---- N orientation: rows = queries 4g + r, column = key lq ----
                    float4v sN = *reinterpret_cast<const float4v*>(biasN_g + bvN * 256), dpN = (float4v)(0.f);
                    sN = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql, kh, sN, 0, 0, 0);
                    sN = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh, kl, sN, 0, 0, 0);
                    sN = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh, kh, sN, 0, 0, 0);
                    dpN = __builtin_amdgcn_mfma_f32_16x16x32_f16(dol, vh, dpN, 0, 0, 0);
                    dpN = __builtin_amdgcn_mfma_f32_16x16x32_f16(doh, vl, dpN, 0, 0, 0);
                    dpN = __builtin_amdgcn_mfma_f32_16x16x32_f16(doh, vh, dpN, 0, 0, 0);
                    float4v pN, dsN;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pN[r] = __builtin_amdgcn_exp2f(fmaf(sN[r], LOG2E, maddN - Lse[qt * 16 + 4 * g + r][wave]));
                        dsN[r] = pN[r] * (dpN[r] - Dl[qt * 16 + 4 * g + r][wave]);
                    }
                    {   // dV += P^T dO, dK += dS^T Q over this tile's 16 queries
                        half4 ph, pl, sh, sl;
                        split4(pN[0], pN[1], pN[2], pN[3], ph, pl);
                        split4(dsN[0], dsN[1], dsN[2], dsN[3], sh, sl);
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const half4 oh = col4(dOh, QS, qt * 16 + 4 * g, hoff + dt * 16), ol = col4(dOl, QS, qt * 16 + 4 * g, hoff + dt * 16);
                            const half4 qh4 = col4(Qh, QS, qt * 16 + 4 * g, hoff + dt * 16), ql4 = col4(Ql, QS, qt * 16 + 4 * g, hoff + dt * 16);
                            dv_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(pl, oh, dv_acc[dt], 0, 0, 0);
                            dv_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ph, ol, dv_acc[dt], 0, 0, 0);
                            dv_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ph, oh, dv_acc[dt], 0, 0, 0);
                            dk_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(sl, qh4, dk_acc[dt], 0, 0, 0);
                            dk_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(sh, ql4, dk_acc[dt], 0, 0, 0);
                            dk_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(sh, qh4, dk_acc[dt], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- store: lane holds keys kt*16 + 4g + r, channel head*32 + dt*16 + lq ----
                int g_o = g, lq_o = lq;
                asm volatile("" : "+v"(g_o), "+v"(lq_o));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = kt * 16 + 4 * g_o + r;
                    const int src = chunk * SPC + kk / N, n = kk % N;
                    if (src < p.n_src) {
                        int row, col;
                        token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
                        float* dst = bp.dkg + (((size_t)(b * p.n_ego + ego) * p.n_src + src) * 2) * P * C +
                                     (size_t)(row * W + col) * C + head * 32 + lq_o;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            dst[dt * 16] = dk_acc[dt][r] * do_inv;
                            dst[(size_t)P * C + dt * 16] = dv_acc[dt][r] * do_inv;
                        }
                    }
                }
            };
            if (khalf == 0) {
                key_tile(std::integral_constant<int, 0>{});
                key_tile(std::integral_constant<int, 1>{});
            } else {
                key_tile(std::integral_constant<int, 2>{});
                key_tile(std::integral_constant<int, 3>{});
            }
        } else {
            // a chunk without a visible key: its gathered-key gradients are zero.  Written here so that the caller does not have
            // to clear the whole (ego, source) gradient buffer first (7 GB per stage at cfg2)
            int g_o = g, lq_o = lq;
            asm volatile("" : "+v"(g_o), "+v"(lq_o));
#pragma unroll
            for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = (2 * khalf + kt2) * 16 + 4 * g_o + r;
                    const int src = chunk * SPC + kk / N, n = kk % N;
                    if (src < p.n_src) {
                        int row, col;
                        token_pixel(p.partition, WIN, X, Y, wx, wy, n, row, col);
                        float* dst = bp.dkg + (((size_t)(b * p.n_ego + ego) * p.n_src + src) * 2) * P * C +
                                     (size_t)(row * W + col) * C + head * 32 + lq_o;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            dst[dt * 16] = 0.f;
                            dst[(size_t)P * C + dt * 16] = 0.f;
                        }
                    }
                }
        }
        BWD_STAMP(4 + 3 * chunk);
        __syncthreads();
    }

    // the two key halves of a head meet: the second wave's partial dQ through LDS (the K tile is free after the last barrier)
    {
        float4v* xch = reinterpret_cast<float4v*>(KVs);           // [head][qt][dt][lane]
        if (khalf == 1) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) xch[((wave * NQT + qt) * 2 + dt) * 64 + lane] = dq_acc[qt][dt];
        }
        __syncthreads();
        if (khalf == 0) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq_acc[qt][dt] += xch[((wave * NQT + qt) * 2 + dt) * 64 + lane];
        }
    }
    // dQ^T tiles: lane holds channels dt*16 + 4g + (0..3) of query qt*16 + lq
    float* dqp = bp.dq + (size_t)(b * L + ego) * P * C;
    if (khalf == 0)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        int row, col;
        token_pixel(p.partition, WIN, X, Y, wx, wy, qt * 16 + lq, row, col);
        float* o = dqp + (size_t)(row * W + col) * C + head * 32 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            *reinterpret_cast<float4*>(o + dt * 16) = make_float4(dq_acc[qt][dt][0] * do_inv, dq_acc[qt][dt][1] * do_inv,
                                                                   dq_acc[qt][dt][2] * do_inv, dq_acc[qt][dt][3] * do_inv);
    }
    BWD_STAMP(30);
    const int vbase = (WIN == 8) ? (khalf == 0 ? 2 : 0) : 0;
#pragma unroll
    for (int v = 0; v < NBW; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (dbias[v][r] != 0.f) unsafeAtomicAdd(bp.d_bias_frag + ((size_t)(head * NB + vbase + v) * 64 + lane) * 4 + r, dbias[v][r] * do_inv);
}

template <int WIN, int HG>
static int launch_attn_bwd_t(const AttnBwdParams& p, hipStream_t st) {
    const int NG = p.f.C / (HG * 32);
    dim3 grid((p.f.H / WIN) * (p.f.W / WIN) * NG, p.f.n_ego, p.f.B);
    hipLaunchKernelGGL((k_attention_bwd<WIN, HG>), grid, dim3(HG * 128), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// attention backward, any window / dim_head (the reference trains every shape its config accepts, hetero_fusion.py:285-327):
// the adjoint of k_attention_any (attn.hip) in the same decomposition - one workgroup per (sample, ego, window, head), one thread
// per query, keys in chunks of 64 gathered into LDS, everything in exact f32 on the vector ALU.  Correctness path, not a tuned
// one: p = exp(s - lse), dS = p (dO . v - D), dq += dS k, and the per-key sums dk += dS q, dv += p dO over
// the window's queries meet in LDS (float atomics) before they leave as the chunk's rows of the gathered-key gradient.
// bias_frag / d_bias_frag are the DENSE (heads, N, N) tables of the generic forward.  Identity self transforms (as all training).
// ------------------------------------------------------------------------------------------
constexpr int ANYB_KC = 64;
template <int DHM>
__global__ __launch_bounds__(256) void k_attention_any_bwd(AttnBwdParams bp) {
    const AttnParams& p = bp.f;
    __shared__ float Ks[ANYB_KC][DHM + 1], Vs[ANYB_KC][DHM + 1], dKs[ANYB_KC][DHM + 1], dVs[ANYB_KC][DHM + 1], maskadd[ANYB_KC];
    const int WIN = p.window, N = WIN * WIN, DH = p.dim_head, C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int heads = C / DH, X = H / WIN, Y = W / WIN;
    const int win = blockIdx.x / heads, head = blockIdx.x - win * heads;
    const int ego = blockIdx.y, b = blockIdx.z;
    const int wx = win / Y, wy = win - wx * Y;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int te = p.mode[b * L + ego], ev = p.ego_e[b * L + ego];
    const int ch0 = head * DH;
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;
    const bool active = tid < N;

    float q[DHM], g[DHM], dq[DHM];
#pragma unroll
    for (int d = 0; d < DHM; ++d) q[d] = g[d] = dq[d] = 0.f;
    int qrow = 0, qcol = 0;
    float D = 0.f, lse = 0.f;
    if (active) {
        token_pixel(p.partition, WIN, X, Y, wx, wy, tid, qrow, qcol);
        const size_t o = ((size_t)(b * L + ego) * P + qrow * W + qcol) * C + ch0;
        const float* bq = p.b_q + te * C + ch0;
        const float* qp = reinterpret_cast<const float*>(p.q) + o;
        const float* op = reinterpret_cast<const float*>(p.out) + o;
        const float* gp = bp.d_out + o;
#pragma unroll
        for (int d = 0; d < DHM; ++d)
            if (d < DH) {
                q[d] = qp[d] + bq[d];
                g[d] = gp[d];
                D = fmaf(g[d], op[d], D);
            }
        lse = p.lse[((size_t)(b * L + ego) * P + qrow * W + qcol) * heads + head];
    }
    const float* bias = p.bias_frag + ((size_t)head * N + (active ? tid : 0)) * N;
    float* dbias = bp.d_bias_frag + ((size_t)head * N + (active ? tid : 0)) * N;

    const int n_keys = p.n_src * N;
    for (int k0 = 0; k0 < n_keys; k0 += ANYB_KC) {
        // ---- gather (as k_attention_any), and clear the chunk's key-gradient tiles ----
        const int kk = tid % ANYB_KC, key = k0 + kk;
        bool visible = false;
        Taps t;
        bool ident = false;
        int self_idx = 0, src = 0, kn = 0;
        if (key < n_keys) {
            src = key / N;
            kn = key - src * N;
            int row, col;
            token_pixel(p.partition, WIN, X, Y, wx, wy, kn, row, col);
            const float* a = p.ainv + ((size_t)(b * L + src) * L + ego) * 8;
            ident = a[6] != 0.f;
            self_idx = row * W + col;
            t.roi = 1.f;
            if (!ident) t = make_taps(a, col, row, H, W);
            visible = (t.roi != 0.f) && (p.cav[b * L + src] != 0);
        }
        {
            const int ts = p.mode[b * L + src];
            const float* kpl = reinterpret_cast<const float*>(p.kv) + ((size_t)((b * L + src) * p.E + ev) * 2) * P * C + ch0;
            const float* bk = p.b_kv + (size_t)(te * HMVIT_NUM_TYPES + ts) * 2 * C + ch0;
            for (int d = tid / ANYB_KC; d < DHM; d += nthr / ANYB_KC) {
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
                dKs[kk][d] = 0.f;
                dVs[kk][d] = 0.f;
            }
            if (tid < ANYB_KC) maskadd[kk] = visible ? 0.f : -INFINITY;
        }
        __syncthreads();
        if (active) {
            const int kmax = min(ANYB_KC, n_keys - k0);
            for (int j = 0; j < kmax; ++j) {
                if (maskadd[j] != 0.f) continue;                 // a masked key: p = 0, nothing flows
                float sdot = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DHM; ++d) {
                    sdot = fmaf(q[d], Ks[j][d], sdot);
                    dp = fmaf(g[d], Vs[j][d], dp);
                }
                const int jn = (k0 + j) % N;
                const float pr = expf(sdot * kl + bias[jn] - lse);
                const float ds = pr * (dp - D);
                atomicAdd(&dbias[jn], ds);
                const float dsk = ds * kl;
#pragma unroll
                for (int d = 0; d < DHM; ++d)
                    if (d < DH) {
                        dq[d] = fmaf(dsk, Ks[j][d], dq[d]);
                        atomicAdd(&dKs[j][d], dsk * q[d]);
                        atomicAdd(&dVs[j][d], pr * g[d]);
                    }
            }
        }
        __syncthreads();
        // ---- the chunk's rows of the gathered-key gradient (every row written: zeros where nothing was visible) ----
        if (key < n_keys) {
            int row, col;
            token_pixel(p.partition, WIN, X, Y, wx, wy, kn, row, col);
            float* dst = bp.dkg + (((size_t)(b * p.n_ego + ego) * p.n_src + src) * 2) * P * C + (size_t)(row * W + col) * C + ch0;
            for (int d = tid / ANYB_KC; d < DH; d += nthr / ANYB_KC) {
                dst[d] = dKs[kk][d];
                dst[(size_t)P * C + d] = dVs[kk][d];
            }
        }
        __syncthreads();
    }
    if (active) {
        float* o = bp.dq + ((size_t)(b * L + ego) * P + qrow * W + qcol) * C + ch0;
#pragma unroll
        for (int d = 0; d < DHM; ++d)
            if (d < DH) o[d] = dq[d];
    }
}

static int launch_attn_any_bwd(const AttnBwdParams& p, hipStream_t st) {
    const int N = p.f.window * p.f.window, DH = p.f.dim_head;
    HMVIT_CHECK_ARG(N <= 256, "attention_bwd (generic): window=%d gives %d tokens per window (at most 256)", p.f.window, N);
    HMVIT_CHECK_ARG(DH >= 1 && DH <= 64 && p.f.C % DH == 0, "attention_bwd (generic): dim_head=%d unsupported (1 .. 64, a divisor of C=%d)", DH, p.f.C);
    const int threads = N <= 64 ? 64 : (N + 63) / 64 * 64;
    dim3 grid((p.f.H / p.f.window) * (p.f.W / p.f.window) * (p.f.C / DH), p.f.n_ego, p.f.B);
    if (DH <= 16) hipLaunchKernelGGL(k_attention_any_bwd<16>, grid, dim3(threads), 0, st, p);
    else if (DH <= 32) hipLaunchKernelGGL(k_attention_any_bwd<32>, grid, dim3(threads), 0, st, p);
    else hipLaunchKernelGGL(k_attention_any_bwd<64>, grid, dim3(threads), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_attention_bwd(const AttnBwdParams& p, hipStream_t st) {
    HMVIT_CHECK_ARG(p.f.C == 64 || p.f.C == 128 || p.f.C == 256, "attention_bwd: C=%d unsupported", p.f.C);
    HMVIT_CHECK_ARG(p.f.lse && p.f.out && p.d_out && p.dq && p.dkg && p.d_bias_frag, "attention_bwd: null pointer");
    if (p.f.n_ego <= 0 || p.f.B <= 0) return HMVIT_OK;
    {
        const int dim_head = p.f.dim_head ? p.f.dim_head : 32;
        if ((p.f.window != 4 && p.f.window != 8) || dim_head != 32) {
            AttnBwdParams q = p;
            q.f.dim_head = dim_head;
            return launch_attn_any_bwd(q, st);
        }
    }
    HMVIT_CHECK_ARG(p.bias_frag_neg != nullptr, "attention_bwd: bias_frag_neg is null");
#ifdef HMVIT_PROBE
    if (const char* e = HMVIT_ENV("HMVIT_BWD_PROBE")) {
        AttnBwdParams q = p;
        q.probe = atoi(e);
        return q.f.window == 8 ? launch_attn_bwd_t<8, 2>(q, st) : launch_attn_bwd_t<4, 2>(q, st);
    }
#endif
    return p.f.window == 8 ? launch_attn_bwd_t<8, 2>(p, st) : launch_attn_bwd_t<4, 2>(p, st);
}

// ------------------------------------------------------------------------------------------
// adjoint of the bilinear gather
// ------------------------------------------------------------------------------------------
// One wavefront per source pixel s.  For an ego whose keys were sampled from this source, the ego pixels u with a tap on s
// satisfy |Ainv u - s|_inf < 1, i.e. u lies within (|A00| + |A01|, |A10| + |A11|) of A s (A = inverse of the sampling
// map): the candidates are the integer points of that box around round(A s), one per lane; lanes with a non-zero weight
// are then visited in turn by the whole wave (4 channels per lane and plane).
constexpr int WADJ_R = 3;   // supported candidate radius: 2 covers every rigid transform (radius < sqrt(2) + 0.5)
// Round 5: the pass doubles as the COLUMN SUMS of the gathered-key gradients (the K' / V' bias gradients: sum over the ego pixels of
// every (ego, source) pair).  Those were a pass of their own over the 7.2 GB buffer (k_colsum, 1.2 ms per stage at 6.7 TB/s, as long as
// this kernel).  Here every ego pixel with a tap in bounds is OWNED by exactly one source pixel - the first of its four taps with a
// non-zero weight - and the wavefront of that pixel, which reads the row for the adjoint anyway, adds it to its running sum; pixels
// without such a tap are invisible keys, whose rows are zero.  A workgroup takes WADJ_RUN x 4 pixels (its four wavefronts four ADJACENT
// pixels at a time: they share most of their candidate rows while those are in L1), then its wavefronts' sums meet in LDS: one atomic add
// per column, plane and workgroup, into one of WADJ_REP replicas of the gradient vector (k_fold_replicas adds them up afterwards:
// atomics on one address serialise at ~160 ns each).  Pixels per workgroup, measured at cfg2 (tools/probe/wadj_ab.sh; the kernel alone,
// no sums, one pixel per wavefront: 1.24 ms): 4 -> 1.63 ms, 8 -> 1.49, 16 -> 1.58, 32 -> 1.64, 128 -> 2.04 - few atomics against
// the balance of many small workgroups (pixels outside a source's footprint cost nothing, pixels inside ~25 row visits).
#ifndef HMVIT_WADJ_RUN
#define HMVIT_WADJ_RUN 2
#endif
constexpr int WADJ_RUN = HMVIT_WADJ_RUN, WADJ_REP = 64;

template <int VPL>
__global__ __launch_bounds__(256) void k_warp_adjoint(WarpAdjParams p) {
    constexpr int C = VPL * 64;
    __shared__ float red[4][2][C];
    const int P = p.H * p.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.y;                 // b * L + src
    const int b = slot / p.L, src = slot - b * p.L;
    if (src >= p.n_src) return;
    const int e = blockIdx.z;
    float csum[2][VPL];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < VPL; ++i) csum[pl][i] = 0.f;
    int te = -1;                                 // type of the egos of variant e (all the same: that is what a variant is)

    // the four wavefronts take four ADJACENT pixels at a time (they share most of their candidate rows while those are in L1: with
    // one run of 32 pixels per wavefront the launch took 2.0 ms against 1.2) and move along the row together
    for (int k = 0; k < WADJ_RUN; ++k) {
        const int s = blockIdx.x * (4 * WADJ_RUN) + 4 * k + wave;
        if (s >= P) break;
        const int sy = s / p.W, sx = s - sy * p.W;
        float acc[2][VPL];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[pl][i] = 0.f;

        for (int ego = 0; ego < p.n_ego; ++ego) {
            if (p.ego_e[b * p.L + ego] != e) continue;
            te = p.mode[b * p.L + ego];
            const float* a = p.ainv + ((size_t)(b * p.L + src) * p.L + ego) * 8;
            const float* g = p.dkg + (((size_t)(b * p.n_ego + ego) * p.n_src + src) * 2) * P * C;
            if (a[6] != 0.f) {   // identity: key u reads source pixel u (and is owned by it)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int i = 0; i < VPL; ++i) {
                        const float v = g[(size_t)pl * P * C + (size_t)s * C + lane * VPL + i];
                        acc[pl][i] += v;
                        csum[pl][i] += v;
                    }
                continue;
            }
            // forward map A = Ainv^-1 (2 x 3)
            const float det = a[0] * a[4] - a[1] * a[3], id = 1.f / det;
            const float f00 = a[4] * id, f01 = -a[1] * id, f10 = -a[3] * id, f11 = a[0] * id;
            const float f02 = -(f00 * a[2] + f01 * a[5]), f12 = -(f10 * a[2] + f11 * a[5]);
            const float ux = f00 * sx + f01 * sy + f02, uy = f10 * sx + f11 * sy + f12;
            const int cx = (int)rintf(fminf(fmaxf(ux, -1.0e6f), 1.0e6f)), cy = (int)rintf(fminf(fmaxf(uy, -1.0e6f), 1.0e6f));
            // lane -> candidate offset in a 7 x 7 box (49 of 64 lanes)
            const int oy = lane / (2 * WADJ_R + 1) - WADJ_R, ox = lane % (2 * WADJ_R + 1) - WADJ_R;
            const int u = cx + ox, v = cy + oy;
            float w = 0.f;
            bool own = false;
            if (lane < (2 * WADJ_R + 1) * (2 * WADJ_R + 1) && u >= 0 && u < p.W && v >= 0 && v < p.H) {
                const Taps t = make_taps(a, u, v, p.H, p.W);
#pragma unroll
                for (int k = 0; k < 4; ++k) w += (t.idx[k] == s) ? t.w[k] : 0.f;
                const int first = t.w[0] != 0.f ? 0 : t.w[1] != 0.f ? 1 : t.w[2] != 0.f ? 2 : 3;
                own = t.w[first] != 0.f && t.idx[first] == s;
            }
            unsigned long long live = __ballot(w != 0.f);
            const unsigned long long owned = __ballot(own);
            // (Measured and dropped: the rows of up to six live candidates requested together before they are used - 2.0 -> 2.3 ms per
            // launch: the kernel is bound by the rows it moves through L1 - each row of dkg is read by ~4 source pixels - not by their latency.)
            while (live) {
                const int i = __ffsll((long long)live) - 1;
                live &= live - 1;
                const float wi = __shfl(w, i, 64);
                const int ui = __shfl(v * p.W + u, i, 64);
                const bool mine = (owned >> i) & 1ull;              // wave-uniform
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int c = 0; c < VPL; ++c) {
                        const float val = g[(size_t)pl * P * C + (size_t)ui * C + lane * VPL + c];
                        acc[pl][c] = fmaf(wi, val, acc[pl][c]);
                        if (mine) csum[pl][c] += val;
                    }
            }
        }
        float* o = p.dkv + ((size_t)(slot * p.E + e) * 2) * P * C + (size_t)s * C + lane * VPL;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < VPL; ++i) o[(size_t)pl * P * C + i] = acc[pl][i];
    }
    if (!p.db_kv) return;                        // (uniform)
    // bias gradients: db_kv[(type of the egos, type of the source)][plane][column] += the workgroup's sum
    __shared__ int te_sh;
    if (threadIdx.x == 0) te_sh = -1;
    __syncthreads();
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < VPL; ++i) red[wave][pl][lane * VPL + i] = csum[pl][i];
    if (te >= 0 && lane == 0) te_sh = te;        // (a wave past the end of the map walked no pixel and knows no type: its sums are 0)
    __syncthreads();
    const int te_wg = te_sh;
    if (te_wg < 0) return;
    const int ts = p.mode[b * p.L + src];
    float* db = p.db_kv + ((size_t)(blockIdx.x % WADJ_REP) * p.T * p.T + (te_wg * p.T + ts)) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int pl = i / C, c = i - pl * C;
        const float v = (red[0][pl][c] + red[1][pl][c]) + (red[2][pl][c] + red[3][pl][c]);
        if (v != 0.f) unsafeAtomicAdd(db + i, v);
    }
}

// out[i] += sum_r rep[r][i], i < n (n a few thousand)
__global__ __launch_bounds__(256) void k_fold_replicas(const float* __restrict__ rep, float* __restrict__ out, int n, int R) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += rep[(size_t)r * n + i];
    out[i] += s;
}
size_t warp_adjoint_replica_floats(int T, int C) { return (size_t)WADJ_REP * T * T * 2 * C; }

int launch_warp_adjoint(const WarpAdjParams& p, hipStream_t st) {
    if (p.B <= 0 || p.n_src <= 0 || p.E <= 0) return HMVIT_OK;
    dim3 grid(cdiv(p.H * p.W, 4 * WADJ_RUN), p.B * p.L, p.E);
    WarpAdjParams q = p;
    const int n_db = p.T * p.T * 2 * p.C;
    if (p.db_kv) {
        HMVIT_CHECK_ARG(p.db_rep != nullptr, "warp_adjoint: bias gradients need the replica buffer%s", "");
        HMVIT_CHECK_HIP(hipMemsetAsync(p.db_rep, 0, warp_adjoint_replica_floats(p.T, p.C) * 4, st));
        q.db_kv = p.db_rep;                     // the kernel adds into the replicas
    }
    switch (p.C) {
        case 64: hipLaunchKernelGGL((k_warp_adjoint<1>), grid, dim3(256), 0, st, q); break;
        case 128: hipLaunchKernelGGL((k_warp_adjoint<2>), grid, dim3(256), 0, st, q); break;
        case 256: hipLaunchKernelGGL((k_warp_adjoint<4>), grid, dim3(256), 0, st, q); break;
        default: set_error("warp_adjoint: C=%d unsupported", p.C); return HMVIT_EINVAL;
    }
    HMVIT_CHECK_LAUNCH();
    if (p.db_kv) hipLaunchKernelGGL(k_fold_replicas, dim3(cdiv(n_db, 256)), dim3(256), 0, st, p.db_rep, p.db_kv, n_db, WADJ_REP);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
