// Typed nn.Linear as an MFMA GEMM: y = act(a w^T + bias) (+ residual).
//
// One launch runs a list of independent jobs (one per agent / weight set), so the per-type
// dispatch of the reference's ModuleLists (hetero_fusion.py:111-152, base_transformer.py:138-192)
// becomes a pointer table in the kernel arguments instead of host-side Python loops.
//
// Tile: 128 tokens x 128 output columns per 256-thread workgroup, 4 wavefronts in a 2x2
// arrangement, each owning a 64x64 block = 2x2 MFMA 32x32 accumulators.
//   f16 mode: v_mfma_f32_32x32x16_f16, K staged in 64-wide slabs, LDS rows padded to 72
//             halves (144 B) so ds_read_b128 of 16 different rows hits 16 different 16-B slots.
//   f32 mode: v_mfma_f32_32x32x2_f32 (exact f32), K staged in 32-wide slabs, rows padded to 33.
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

template <typename T>
struct GemmCfg;
template <>
struct GemmCfg<half_t> {
    static constexpr int BK = 64, LDS_STRIDE = 72;
};
template <>
struct GemmCfg<float> {
    static constexpr int BK = 32, LDS_STRIDE = 33;
};

constexpr int BM = 128, BN = 128;

// ---- global -> LDS staging of a (128 x BK) slab whose rows are `ld` elements apart ----
template <typename T, typename TS>
__device__ __forceinline__ void stage_slab(T* __restrict__ lds, const TS* __restrict__ g, int row0,
                                           int rows_valid, int ld, int k0) {
    constexpr int BK = GemmCfg<T>::BK, LS = GemmCfg<T>::LDS_STRIDE;
    const int tid = threadIdx.x;
    if constexpr (sizeof(T) == 2 && sizeof(TS) == 2) {
        // 128 x 64 halves: 8 chunks of 16 B per row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            half8 v = (half8)(half_t)0;
            if (row0 + row < rows_valid)
                v = *reinterpret_cast<const half8*>(g + (size_t)(row0 + row) * ld + k0 + kc);
            *reinterpret_cast<half8*>(lds + row * LS + kc) = v;
        }
    } else if constexpr (sizeof(T) == 2 && sizeof(TS) == 4) {
        // f32 source converted on the way in: 128 x 64 floats, 16 float4 per row
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i, row = c >> 4, kc = (c & 15) * 4;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + row < rows_valid)
                f = *reinterpret_cast<const float4*>(g + (size_t)(row0 + row) * ld + k0 + kc);
            half4 h;
            h[0] = (half_t)f.x; h[1] = (half_t)f.y; h[2] = (half_t)f.z; h[3] = (half_t)f.w;
            *reinterpret_cast<half4*>(lds + row * LS + kc) = h;
        }
    } else {
        // 128 x 32 floats: 8 float4 per row, scalar LDS writes (odd row stride)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 4;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + row < rows_valid)
                f = *reinterpret_cast<const float4*>(g + (size_t)(row0 + row) * ld + k0 + kc);
            float* d = reinterpret_cast<float*>(lds) + row * LS + kc;
            d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
        }
    }
}

template <typename T, bool A_F32, bool GELU, bool OUT_F32>
__global__ __launch_bounds__(256) void k_gemm(GemmJobs jobs) {
    constexpr int BK = GemmCfg<T>::BK, LS = GemmCfg<T>::LDS_STRIDE;
    __shared__ __attribute__((aligned(16))) T As[BM * LS];
    __shared__ __attribute__((aligned(16))) T Ws[BN * LS];

    const GemmJob& J = jobs.j[blockIdx.y];
    const int M = J.M, N = J.N, K = J.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = blockIdx.x;
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hi = lane >> 5;

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        if constexpr (A_F32)
            stage_slab<T, float>(As, reinterpret_cast<const float*>(J.a), m0, M, K, k0);
        else
            stage_slab<T, T>(As, reinterpret_cast<const T*>(J.a), m0, M, K, k0);
        stage_slab<T, T>(Ws, reinterpret_cast<const T*>(J.w), n0, N, K, k0);
        __syncthreads();
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                half8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const half8*>(As + (wm * 64 + i * 32 + r) * LS + kk * 16 + hi * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *reinterpret_cast<const half8*>(Ws + (wn * 64 + j * 32 + r) * LS + kk * 16 + hi * 8);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* Af = reinterpret_cast<const float*>(As);
            const float* Wf = reinterpret_cast<const float*>(Ws);
#pragma unroll 4
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = Af[(wm * 64 + i * 32 + r) * LS + kk * 2 + hi];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Wf[(wn * 64 + j * 32 + r) * LS + kk * 2 + hi];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue: acc[i][j][e] is row (e&3) + 8*(e>>2) + 4*hi, column r of its 32x32 block
    const int npp = J.n_per_plane;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= N) continue;
        const float bias = J.bias ? J.bias[n] : 0.f;
        const int plane = n / npp, nc = n - plane * npp;
        const size_t ybase = (size_t)plane * J.plane_stride + nc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (m >= M) continue;
                float v = acc[i][j][e] + bias;
                if constexpr (GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                if (J.residual) v += J.residual[(size_t)m * N + n];
                const size_t o = ybase + (size_t)m * npp;
                if constexpr (OUT_F32)
                    reinterpret_cast<float*>(J.y)[o] = v;
                else
                    reinterpret_cast<T*>(J.y)[o] = (T)v;
            }
        }
    }
}

// ---- split mode: f32 a (M, K) and f32 w (N, K), every product as (hi + lo) halves on the f16 pipes ----
// a w^T = a_hi w_hi + a_lo w_hi + a_hi w_lo (+ 2^-22 relative), f32 accumulate, f32 result.  The slabs are split on the way into
// LDS (hi = f16(x), lo = f16(x - hi)); same 128 x 128 tile and wave arrangement as above.  Used by the training path
// (csrc/capi_train.hip), whose operands are f32 activations and f32 master weights.
// a 128 x 64 f32 slab: 8 float4 per thread, fetched into registers one slab ahead of the products (fetch_slab) and split into
// the LDS planes at the top of the next iteration (put_slab)
__device__ __forceinline__ void fetch_slab(float4 (&f)[8], const float* __restrict__ g, int row0, int rows_valid, int ld, int k0) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 4, kc = (c & 15) * 4;
        f[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < rows_valid) f[i] = *reinterpret_cast<const float4*>(g + (size_t)(row0 + row) * ld + k0 + kc);
    }
}
__device__ __forceinline__ void put_slab(half_t* __restrict__ lds_hi, half_t* __restrict__ lds_lo, const float4 (&f)[8], float sc) {
    constexpr int LS = GemmCfg<half_t>::LDS_STRIDE;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 4, kc = (c & 15) * 4;
        half4 h, l;
        split_pk4(f[i].x * sc, f[i].y * sc, f[i].z * sc, f[i].w * sc, h, l);
        *reinterpret_cast<half4*>(lds_hi + row * LS + kc) = h;
        *reinterpret_cast<half4*>(lds_lo + row * LS + kc) = l;
    }
}
// max |.| of the slab a workgroup holds in registers -> every thread (one LDS round trip, two barriers around the partials)
__device__ __forceinline__ float slab_absmax(const float4 (&f)[8], float* partial) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fmaxf(fmaxf(fabsf(f[i].x), fabsf(f[i].y)), fmaxf(fabsf(f[i].z), fabsf(f[i].w))));
    m = wave_absmax(m);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(partial[0], partial[1]), fmaxf(partial[2], partial[3]));
    return m;
}

template <bool GELU>
__global__ __launch_bounds__(256) void k_gemm_split(GemmJobs jobs) {
    constexpr int BK = GemmCfg<half_t>::BK, LS = GemmCfg<half_t>::LDS_STRIDE;
    __shared__ __attribute__((aligned(16))) half_t Ah[BM * LS], Al[BM * LS], Wh[BN * LS], Wl[BN * LS];
    const GemmJob& J = jobs.j[blockIdx.y];
    const int M = J.M, N = J.N, K = J.K;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = blockIdx.x;
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hi = lane >> 5;

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const float* Ag = reinterpret_cast<const float*>(J.a);
    const float* Wg = reinterpret_cast<const float*>(J.w);
    float4 fa[8], fw[8];
    fetch_slab(fa, Ag, m0, M, K, 0);
    fetch_slab(fw, Wg, n0, N, K, 0);
    // Range (round 5): every K slab is split at its own powers of two (operands of the training path are gradients at whatever
    // magnitude the pass has reached); the accumulator runs at the current pair of scales and is multiplied by the ratio - a
    // power of two, exact - when a slab needs another pair (csrc/train.hip k_gemm_tn_split does the same along the tokens)
    __shared__ float pmax[2][4];
    float sa_run = 1.f, sw_run = 1.f;
    bool live = false;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const float ma = slab_absmax(fa, pmax[0]), mw = slab_absmax(fw, pmax[1]);
        const float ta = ma * sa_run, tw = mw * sw_run;
        const bool keep = live && ta < 32768.f && tw < 32768.f && (ta >= 256.f || ma == 0.f) && (tw >= 256.f || mw == 0.f);
        if (!keep) {
            const float sa = pow2_scale(ma), sw = pow2_scale(mw);
            if (live) {
                // two factors, applied one after the other: each is a power of two within 2^+-80, their product can leave f32 and an
                // infinite factor would turn an accumulator that holds 0 into NaN
                const float f1 = sa * pow2_inv(sa_run), f2 = sw * pow2_inv(sw_run);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * f1) * f2;
            }
            sa_run = sa; sw_run = sw; live = true;
        }
        put_slab(Ah, Al, fa, sa_run);
        put_slab(Wh, Wl, fw, sw_run);
        __syncthreads();
        if (k0 + BK < K) {                      // the next slab travels while this one is multiplied
            fetch_slab(fa, Ag, m0, M, K, k0 + BK);
            fetch_slab(fw, Wg, n0, N, K, k0 + BK);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32 + r) * LS + kk * 16 + hi * 8;
                ah[i] = *reinterpret_cast<const half8*>(Ah + o);
                al[i] = *reinterpret_cast<const half8*>(Al + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 64 + j * 32 + r) * LS + kk * 16 + hi * 8;
                bh[j] = *reinterpret_cast<const half8*>(Wh + o);
                bl[j] = *reinterpret_cast<const half8*>(Wl + o);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    const int npp = J.n_per_plane;
    const float unscale_a = pow2_inv(sa_run), unscale_w = pow2_inv(sw_run);     // (applied one after the other, as above)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= N) continue;
        const float bias = J.bias ? J.bias[n] : 0.f;
        const int plane = n / npp, nc = n - plane * npp;
        const size_t ybase = (size_t)plane * J.plane_stride + nc;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (m >= M) continue;
                float v = fmaf(acc[i][j][e] * unscale_a, unscale_w, bias);
                if constexpr (GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                if (J.residual) v += J.residual[(size_t)m * N + n];
                reinterpret_cast<float*>(J.y)[ybase + (size_t)m * npp] = v;
            }
        }
    }
}

static int launch_gemm_split(const GemmJobs& jobs, bool gelu, hipStream_t st) {
    int max_tiles = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const GemmJob& j = jobs.j[i];
        HMVIT_CHECK_ARG(j.K > 0 && j.K % 64 == 0, "gemm (split): K=%d must be a positive multiple of 64", j.K);
        max_tiles = max(max_tiles, cdiv(j.M, BM) * cdiv(j.N, BN));
    }
    if (jobs.n == 0 || max_tiles == 0) return HMVIT_OK;
    dim3 grid(max_tiles, jobs.n), block(256);
    if (gelu) hipLaunchKernelGGL((k_gemm_split<true>), grid, block, 0, st, jobs);
    else hipLaunchKernelGGL((k_gemm_split<false>), grid, block, 0, st, jobs);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

template <typename T>
static int launch_gemm_t(const GemmJobs& jobs, bool a_f32, bool gelu, bool out_f32, hipStream_t st) {
    int max_tiles = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const GemmJob& j = jobs.j[i];
        if (j.K % GemmCfg<T>::BK != 0 || j.K <= 0) {
            set_error("gemm: K=%d must be a positive multiple of %d", j.K, GemmCfg<T>::BK);
            return HMVIT_EINVAL;
        }
        const int t = cdiv(j.M, BM) * cdiv(j.N, BN);
        if (t > max_tiles) max_tiles = t;
    }
    if (jobs.n == 0 || max_tiles == 0) return HMVIT_OK;
    dim3 grid(max_tiles, jobs.n), block(256);
#define HMVIT_GEMM_CASE(AF, GE, OF)                                                    \
    if (a_f32 == AF && gelu == GE && out_f32 == OF) {                                  \
        hipLaunchKernelGGL((k_gemm<T, AF, GE, OF>), grid, block, 0, st, jobs);         \
        HMVIT_CHECK_LAUNCH();                                                          \
        return HMVIT_OK;                                                               \
    }
    HMVIT_GEMM_CASE(false, false, false)
    HMVIT_GEMM_CASE(false, false, true)
    HMVIT_GEMM_CASE(false, true, false)
    HMVIT_GEMM_CASE(true, false, false)
    HMVIT_GEMM_CASE(true, true, false)
    HMVIT_GEMM_CASE(true, false, true)
    HMVIT_GEMM_CASE(false, true, true)
    HMVIT_GEMM_CASE(true, true, true)
#undef HMVIT_GEMM_CASE
    return HMVIT_EINVAL;
}

int launch_gemm(const GemmJobs& jobs, bool a_f32, bool gelu, bool out_f32, int precision,
                hipStream_t st) {
    if (precision == HMVIT_PREC_F32) return launch_gemm_t<float>(jobs, a_f32, gelu, out_f32, st);
    if (precision == HMVIT_PREC_SPLIT) {
        HMVIT_CHECK_ARG(a_f32 && out_f32, "gemm (split): f32 operands and f32 result only%s", "");
        return launch_gemm_split(jobs, gelu, st);
    }
    return launch_gemm_t<half_t>(jobs, a_f32, gelu, out_f32, st);
}

}  // namespace hmvit
