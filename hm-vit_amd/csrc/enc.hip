// LiDAR BEV encoder kernels (PointPillar branch):
//
//   k_pfn_scatter  PillarVFE (one PFN layer) fused with PointPillarScatter
//                  (sub_modules/pillar_vfe.py:31-53,105-146, point_pillar_scatter.py:14-47):
//                  HBM-bound, one wavefront per pillar, writes the dense NHWC canvas directly.
//   k_conv         Conv2d / ConvTranspose2d(kernel = stride) + folded BatchNorm + ReLU as an
//                  implicit GEMM on MFMA (backbones/base_bev_backbone.py:6-122,
//                  sub_modules/downsample_conv.py:20-51): NHWC activations, the im2col row of an
//                  output pixel is gathered on the fly (one 3x3 tap x BK input channels per K slab).
//
// Same tile as gemm.hip: 128 output pixels x 128 output channels per workgroup, 2x2 wavefronts,
// v_mfma_f32_32x32x16_f16 (f16 mode) or v_mfma_f32_32x32x2_f32 (exact f32 mode).
#include <type_traits>
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

// max |y| of a launch into one device slot (f32 bit pattern; non-negative floats order like unsigned integers).  Every wavefront
// of every workgroup ends here, on ONE address: as unconditional atomics that serialisation was 5 - 50 % of the split
// convolutions (round-4 ablation on the PointPillar layers: 64 -> 64 channels on 256 x 256 x 5 maps 143 -> 68 us without it).
// The slot only ever grows, so a workgroup whose maximum does not exceed a value the slot has already held has nothing to add.
// absmax_peek: thread 0's L2-coherent look at the slot when the workgroup STARTS (nobody waits for it); absmax_raise_wg: the
// workgroup's maximum against that value at its end, against a second look where it is larger, and an atomic only if it still is.
// A stale look can only be too small, i.e. cost a load or an atomic that was not needed.
__device__ __forceinline__ unsigned absmax_peek(const unsigned* slot) {
    return (slot && threadIdx.x == 0) ? __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
__device__ __forceinline__ void absmax_raise(unsigned* slot, float m, unsigned seen) {
#ifndef HMVIT_EXP_NOAMAX
    // larger than the early look: look again (the slot has usually caught up by now) before paying for an atomic
    if (m > 0.f && __float_as_uint(m) > seen && __float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(slot, __float_as_uint(m));
#endif
}
// 256-thread workgroup, every wavefront past its last LDS read; `scratch`: 4 floats of LDS that nobody reads any more
__device__ __forceinline__ void absmax_raise_wg(unsigned* slot, float m, unsigned seen, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) absmax_raise(slot, fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3])), seen);
}

// ------------------------------------------------------------------------------------------
// PFN + scatter
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void k_pfn_scatter(PfnParams p) {
    __shared__ float feat[4][32][12];
    __shared__ float amax_scratch[4];      // its own words: a wavefront may still be reading feat[0] when another one reports
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + wave;
    // split-mode consumers want max |canvas| (hmvit_conv_range before this call): one look at the slot per workgroup; the last,
    // partial workgroup (wavefronts return early: no workgroup barrier there) raises per wavefront
    const bool full_wg = blockIdx.x * 4 + 4 <= p.n_pillars;
    const unsigned amax_seen = full_wg ? absmax_peek(p.canvas_absmax) : 0u;
    if (v >= p.n_pillars) return;
    const int npts = p.num_points[v];
    const int4 co = *reinterpret_cast<const int4*>(p.coords + (size_t)v * 4);   // [agent, z, y, x]

    // lanes 0..31: one point each; sums of x, y, z over all 32 rows (padding rows are zero)
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < 32) pt = *reinterpret_cast<const float4*>(p.voxels + ((size_t)v * 32 + lane) * 4);
    float sx = pt.x, sy = pt.y, sz = pt.z;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64);
        sy += __shfl_xor(sy, o, 64);
        sz += __shfl_xor(sz, o, 64);
    }
    const float inv = 1.f / (float)npts;
    if (lane < 32) {
        const float m = lane < npts ? 1.f : 0.f;      // padded points are zeroed AFTER augmentation
        float* f = feat[wave][lane];
        f[0] = pt.x * m; f[1] = pt.y * m; f[2] = pt.z * m; f[3] = pt.w * m;
        f[4] = (pt.x - sx * inv) * m; f[5] = (pt.y - sy * inv) * m; f[6] = (pt.z - sz * inv) * m;
        f[7] = (pt.x - ((float)co.w * p.vx + p.x_off)) * m;
        f[8] = (pt.y - ((float)co.z * p.vy + p.y_off)) * m;
        f[9] = (pt.z - ((float)co.y * p.vz + p.z_off)) * m;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // lane = output channel: Linear(10 -> 64, BN scale folded) + shift, ReLU, max over the points
    float w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = p.w[lane * 10 + k];
    const float shift = p.shift[lane];
    float best = 0.f;   // ReLU output is >= 0
    for (int j = 0; j < 32; ++j) {
        const float* f = feat[wave][j];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(w[k], f[k], acc);
        best = fmaxf(best, acc + shift);
    }
    if (p.pillar_out) p.pillar_out[(size_t)v * 64 + lane] = best;
    // the reference's indexed scatter raises on an index outside the canvas (point_pillar_scatter.py:30-40); here such a
    // pillar is dropped and counted (a stale agent index or a coordinate outside the grid must not write out of bounds)
    const bool inside = (unsigned)co.x < (unsigned)p.n_agents && (unsigned)co.z < (unsigned)p.ny &&
                        (unsigned)co.w < (unsigned)p.nx && co.y == 0;
    if (!inside && p.oob_count && lane == 0) atomicAdd(p.oob_count, 1);
    if (p.canvas && inside) {
        const size_t cell = ((size_t)co.x * p.ny + co.z) * p.nx + co.w + co.y;   // index z + y * nx + x
        reinterpret_cast<TO*>(p.canvas)[cell * 64 + lane] = (TO)best;
    }
    if (p.canvas_absmax) {
        float m = inside ? best : 0.f;
        if (full_wg) {
            absmax_raise_wg(p.canvas_absmax, m, amax_seen, amax_scratch);
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) absmax_raise(p.canvas_absmax, m, 0u);
        }
    }
}

int launch_pfn_scatter(const PfnParams& p, int precision, hipStream_t st) {
    if (p.n_pillars <= 0) return HMVIT_OK;
    dim3 grid(cdiv(p.n_pillars, 4)), block(256);
    if (precision != HMVIT_PREC_F16)                       // f32 canvas for the exact-f32 and the split-operand convolutions
        hipLaunchKernelGGL((k_pfn_scatter<float>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((k_pfn_scatter<half_t>), grid, block, 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// ------------------------------------------------------------------------------------------
template <typename T>
struct ConvCfg;
template <>
struct ConvCfg<half_t> {
    static constexpr int BK = 64, LS = 72;
};
template <>
struct ConvCfg<float> {
    static constexpr int BK = 32, LS = 33;
};

// CBN = 128 or 64 output channels per workgroup (64: the 64-channel layers of the first backbone block would leave half
// of a 128-wide tile empty).  2 x 2 wavefronts, each 64 pixels x CBN / 2 channels.  The K slabs are double-buffered in
// LDS: slab k + 1 travels global -> registers while slab k is multiplied, registers -> the other buffer afterwards, one
// barrier per slab.
typedef int int4v __attribute__((ext_vector_type(4)));
__device__ int4v llvm_raw_buffer_load_b128(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ __forceinline__ int4v conv_rsrc(const void* base, size_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    int4v rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)(unsigned)(bytes < 0xfffffff0ull ? bytes : 0xfffffff0ull));
    rs.w = 0x00020000;
    return rs;
}

// Staged row of 16-byte piece c (8 pieces per row).  f16 rows go to LDS as one 16-byte store per piece (groups of 8 lanes = one
// row).  The split kernels store 8-byte (hi) + 8-byte (lo) halves: ds_write_b64 is served in groups of 16 lanes over 32 banks, and
// with rows of 144 bytes two CONSECUTIVE rows in a group collide on 12 of their 16 banks (measured: 33-40 % of the LDS cycles of
// the split convolutions were bank conflicts) - so a group takes rows r and r + 4 of its wave's 8 rows (144 * 4 = 16 banks apart).
template <bool SPLIT>
__device__ __forceinline__ int stage_row(int c) {
    if constexpr (SPLIT) return ((c >> 6) << 3) | (((c >> 3) & 1) << 2) | ((c >> 4) & 3);
    else return c >> 3;
}

// channel (inside a 32-channel tile) whose weight row is staged as LDS / MFMA row rho: an accumulator lane (pixel r, half
// hi) owns rows 8 q + 4 hi + e, which become two runs of eight consecutive channels (cf. weights.py store_row_order)
__device__ __forceinline__ int conv_row_channel(int rho) {
    const int q = rho >> 3, h = (rho >> 2) & 1, e = rho & 3;
    return 16 * (q >> 1) + 8 * h + 4 * (q & 1) + e;
}

// SPLIT (T = float): f32 map, f32 weights, f32 result like the exact-f32 instantiation, but every product runs on the f16 pipes
// as (hi + lo) halves - x w = x_hi w_hi + x_lo w_hi + x_hi w_lo, f32 accumulate (the scheme of the fusion's split mode, DESIGN
// 5.0).  A 32-deep K slab is split on its way into LDS: a staged row holds [32 hi | 32 lo | pad] halves, i.e. the f16
// kernel's 72-half row, so the LDS footprint and the conflict-free fragment reads are the f16 kernel's.
// RING (split only): the weight slabs come out of the GEMM-order image (launch_conv_pack, kind 1) by LDS-DMA - slab g + 1 is
// requested into the other of two unpadded, swizzled buffers at the start of iteration g; no weight registers, no hi / lo split,
// no ds_write for them.  Confirmation rides on hipcc's own wait for the input rows: the request is issued BEFORE this iteration's
// row loads, and the vmcnt(rows of one slab) in front of the row stores at the end of the iteration completes everything older.
template <typename T, int CBN, int CBM, bool SPLIT = false, bool RING = false>
__global__ __launch_bounds__(256, 2) void k_conv(ConvParams p) {
    static_assert(!SPLIT || sizeof(T) == 4, "the split instantiation reads f32 operands");
    static_assert(!RING || SPLIT, "the weight ring is built for the split instantiation");
    using TL = typename std::conditional<SPLIT, half_t, T>::type;        // LDS element
    constexpr int BK = ConvCfg<T>::BK, LS = SPLIT ? 72 : ConvCfg<T>::LS;
    constexpr int MI = CBM / 64;         // 32-pixel MFMA tiles per wave (pixel tile of 128 or, for small maps, 64)
    constexpr int RPT = CBM / 32;        // staged A rows per thread (8 chunks per row)
    constexpr int RPW = CBN / 32;        // staged W rows per thread
    constexpr int NJ = CBN / 64;         // 32-channel MFMA tiles per wave
    __shared__ __attribute__((aligned(16))) TL As[2][CBM * LS];
    __shared__ __attribute__((aligned(1024))) TL Ws[2][CBN * (RING ? 64 : LS)];

    const int M = p.N * p.Ho * p.Wo;
    const int Ncols = p.deconv_s ? p.deconv_s * p.deconv_s * p.Cout : p.Cout;
    const int Ktot = p.rowpack ? p.KH * 32 : p.KH * p.KW * p.Cin;
    const int tiles_n = (Ncols + CBN - 1) / CBN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * CBM, n0 = tn * CBN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, hi = lane >> 5;
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* w = reinterpret_cast<const T*>(p.w);
    // split mode: range normalisation of both operands (ConvParams::absmax)
    float sx = 1.f, sw = 1.f, s_inv = 1.f;
    if constexpr (SPLIT) {
        if (p.x_absmax) {
            sx = pow2_scale(__uint_as_float(p.x_absmax[0]));
            if (p.w_absmax < 0.f) {            // the weights arrive pre-multiplied by the power of two -w_absmax: nothing to do at staging
                s_inv = pow2_inv(sx) * pow2_inv(-p.w_absmax);
            } else {
                sw = pow2_scale(p.w_absmax > 0.f ? p.w_absmax : __uint_as_float(p.x_absmax[1]));
                s_inv = pow2_inv(sx) * pow2_inv(sw);
            }
        }
    }
    const bool w_mul = sw != 1.f;
    const unsigned amax_seen = SPLIT ? absmax_peek(p.y_absmax) : 0u;

    // the rows this thread stages: decode (image, oy, ox) once
    int rn[RPT], roy[RPT], rox[RPT];
    bool rvalid[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = stage_row<SPLIT>(tid + 256 * i);
        const int m = m0 + row;
        rvalid[i] = m < M;
        const int mm = rvalid[i] ? m : 0;
        rn[i] = mm / (p.Ho * p.Wo);
        const int rem = mm - rn[i] * p.Ho * p.Wo;
        roy[i] = rem / p.Wo;
        rox[i] = rem - roy[i] * p.Wo;
    }

    // (not up2) per staged row: coordinates and element offset of tap (0, 0), 16-byte piece of the slab included
    int riy0[RPT], rix0[RPT], rowoff[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        riy0[i] = roy[i] * p.stride - p.pad;
        rix0[i] = rox[i] * p.stride - p.pad;
        rowoff[i] = ((rn[i] * p.H + riy0[i]) * p.W + rix0[i]) * p.Cin + ((tid + 256 * i) & 7) * (int)(16 / sizeof(T));
    }

    float16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // global -> registers (the next K slab is fetched while the current one is multiplied), registers -> LDS
    using Vec = typename std::conditional<sizeof(T) == 2, half8, float4v>::type;   // 16 bytes of a row
    constexpr int VE = 16 / sizeof(T);
    Vec ra[2][RPT], rw[2][RPW];     // two slabs in flight: slab k + 2 is requested while slab k is multiplied
    int s_ky = 0, s_kx = 0, s_ci0 = 0;       // tap and first channel of the next slab to be requested
    unsigned wrow[RPW];                      // byte offset of the weight rows staged by this thread (16-byte piece included)
    bool wvalid[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int c = tid + 256 * i, row = stage_row<SPLIT>(c);
        const int nrow = n0 + (row & ~31) + conv_row_channel(row & 31);   // output channel whose weights go to LDS row `row`
        wvalid[i] = nrow < Ncols;
        wrow[i] = (unsigned)(((size_t)(wvalid[i] ? nrow : 0) * Ktot + (c & 7) * VE) * sizeof(T));
    }
    // raw buffer descriptors of the input map and the weight matrix (an offset >= num_records reads as zero)
    const size_t x_bytes = (size_t)p.N * p.H * p.W * p.Cin * sizeof(T) >> (p.up2 ? 2 : 0), w_bytes = (size_t)Ncols * Ktot * sizeof(T);
    const int4v rs_x = conv_rsrc(x, x_bytes), rs_w = conv_rsrc(w, w_bytes);
    auto load_slab = [&](int k0, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        if (p.rowpack) {
            // few-channel stem: k = ky * 32 + px * 4 + ci over rows of 8 pixels x 4 channels, which are contiguous in the
            // physically padded NHWC4 input (no bounds checks; pixel 7 and any row beyond the kernel carry zero weights)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int k = k0 + ((tid + 256 * i) & 7) * VE;
                const int ky = k >> 5, px = (k & 31) >> 2;
                ra[SET][i] = (Vec)(T)0;
                if (rvalid[i]) {
                    const size_t pix = ((size_t)rn[i] * p.H + roy[i] * p.stride + ky) * p.W + rox[i] * p.stride + px;
                    ra[SET][i] = *reinterpret_cast<const Vec*>(x + pix * 4);
                }
            }
        } else {
            // slabs are requested in order, so (ky, kx, first channel) advance by counting instead of dividing k0 out
            const int ky = s_ky, kx = s_kx, ci0 = s_ci0;
            s_ci0 += BK;
            if (s_ci0 >= p.Cin) { s_ci0 = 0; if (++s_kx == p.KW) { s_kx = 0; ++s_ky; } }
            if (!p.up2) {
                // the usual case: element offset of the row's tap (0, 0) + a wave-uniform tap offset (32-bit: NHWC maps
                // stay below 2^31 elements), range test on the two coordinates
                // buffer loads: a row outside the map asks for an out-of-range offset and gets zeros - no branch per row
                const int toff = (ky * p.W + kx) * p.Cin + ci0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const bool ok = rvalid[i] && (unsigned)(riy0[i] + ky) < (unsigned)p.H && (unsigned)(rix0[i] + kx) < (unsigned)p.W;
                    const unsigned off = ok ? (unsigned)(rowoff[i] + toff) * (unsigned)sizeof(T) : 0xffffffffu;
                    ra[SET][i] = __builtin_bit_cast(Vec, llvm_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
                }
            } else {
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int c = tid + 256 * i, kc = (c & 7) * VE;
                    // A slab: im2col rows gathered from the NHWC input
                    const int iy = roy[i] * p.stride + ky - p.pad, ix = rox[i] * p.stride + kx - p.pad;
                    const bool ok = rvalid[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    ra[SET][i] = (Vec)(T)0;
                    if (ok) {
                        // up2: logical pixel (iy, ix) of the upsampled map is physical pixel (iy / 2, ix / 2)
                        const size_t pix = ((size_t)rn[i] * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + (ix >> 1);
                        ra[SET][i] = *reinterpret_cast<const Vec*>(x + pix * p.Cin + ci0 + kc);
                    }
                }
            }
        }
        if constexpr (!RING) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
                rw[SET][i] = __builtin_bit_cast(Vec, llvm_raw_buffer_load_b128(rs_w, (int)(wvalid[i] ? wrow[i] + (unsigned)k0 * (unsigned)sizeof(T) : 0xffffffffu), 0, 0));
        }
    };
    // RING: slab g of this channel tile -> buffer `buf`; this wavefront's CBN / 32 pieces of 1 KB
    const unsigned ws_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&Ws[0][0];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* img = RING ? reinterpret_cast<const char*>(p.w_image) + (size_t)tn * (Ktot / BK) * (CBN * 128) + lane * 16 : nullptr;
    auto request = [&](int g, int buf) {
        if constexpr (RING) {
            const char* src = img + (size_t)g * (CBN * 128);
#pragma unroll
            for (int i = 0; i < CBN / 32; ++i) {
                const char* a = src + (i * 4 + wave_u) * 1024;
                const unsigned dst = __builtin_amdgcn_readfirstlane(ws_lds + (unsigned)buf * (CBN * 128) + (i * 4 + wave_u) * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(a), "s"(dst) : "memory");
            }
        }
    };
    auto store_slab = [&](int buf, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int c = tid + 256 * i, row = stage_row<SPLIT>(c), kc = (c & 7) * VE;
            if constexpr (SPLIT) {
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ra[SET][i][e] * sx;
                    h[e] = (half_t)v;
                    l[e] = (half_t)(v - (float)h[e]);
                }
                *reinterpret_cast<half4*>(As[buf] + row * LS + kc) = h;
                *reinterpret_cast<half4*>(As[buf] + row * LS + 32 + kc) = l;
            } else if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<half8*>(As[buf] + row * LS + kc) = ra[SET][i];
            } else {
                float* da = reinterpret_cast<float*>(As[buf]) + row * LS + kc;   // LS = 33 floats: rows are not 16-byte aligned
#pragma unroll
                for (int e = 0; e < 4; ++e) da[e] = ra[SET][i][e];
            }
        }
        if constexpr (!RING)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int c = tid + 256 * i, row = stage_row<SPLIT>(c), kc = (c & 7) * VE;
            if constexpr (SPLIT) {
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = w_mul ? rw[SET][i][e] * sw : rw[SET][i][e];
                    h[e] = (half_t)v;
                    l[e] = (half_t)(v - (float)h[e]);
                }
                *reinterpret_cast<half4*>(Ws[buf] + row * LS + kc) = h;
                *reinterpret_cast<half4*>(Ws[buf] + row * LS + 32 + kc) = l;
            } else if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<half8*>(Ws[buf] + row * LS + kc) = rw[SET][i];
            } else {
                float* dw = reinterpret_cast<float*>(Ws[buf]) + row * LS + kc;
#pragma unroll
                for (int e = 0; e < 4; ++e) dw[e] = rw[SET][i][e];
            }
        }
    };

    // Software pipeline, two slabs deep in registers (the loop body is written for a fixed slab parity so that register
    // sets and LDS buffers are static): slab k sits in LDS buffer k & 1; while it is multiplied, slab k + 2 is requested
    // into register set k & 1 and slab k + 1 - requested one iteration earlier, so it has had a whole slab time to arrive
    // - goes from set (k + 1) & 1 to the other LDS buffer.  (One slab of lookahead left every iteration waiting for
    // its global loads.)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    request(0, 0);
    load_slab(0, I0{});
    if (BK < Ktot) load_slab(BK, I1{});
    store_slab(0, I0{});
    __syncthreads();
    auto body = [&](int k0, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        using Same = std::integral_constant<int, PAR>;
        using Other = std::integral_constant<int, 1 - PAR>;
        if (k0 + BK < Ktot) request(k0 / BK + 1, 1 - PAR);     // (that buffer was last read before the previous barrier)
        if (k0 + 2 * BK < Ktot) load_slab(k0 + 2 * BK, Same{});
        // D[channel][pixel]: the weights are the A operand, so that an accumulator lane owns one output pixel and
        // runs of 4 consecutive channels (vector stores in the epilogue instead of 2-byte ones)
        if constexpr (SPLIT) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                half8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const half_t* q = As[PAR] + (wm * (CBM / 2) + i * 32 + r) * LS + kk * 16 + hi * 8;
                    ah[i] = *reinterpret_cast<const half8*>(q);
                    al[i] = *reinterpret_cast<const half8*>(q + 32);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (RING) {       // unpadded 128-byte rows, 16-byte pieces swizzled by (row >> 1) & 7
                        const int row = wn * (CBN / 2) + j * 32 + r, key = (row >> 1) & 7;
                        bh[j] = *reinterpret_cast<const half8*>(Ws[PAR] + row * 64 + (((kk * 2 + hi) ^ key) << 3));
                        bl[j] = *reinterpret_cast<const half8*>(Ws[PAR] + row * 64 + (((kk * 2 + hi + 4) ^ key) << 3));
                    } else {
                        const half_t* q = Ws[PAR] + (wn * (CBN / 2) + j * 32 + r) * LS + kk * 16 + hi * 8;
                        bh[j] = *reinterpret_cast<const half8*>(q);
                        bl[j] = *reinterpret_cast<const half8*>(q + 32);
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    }
            }
        } else if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                half8 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    a[i] = *reinterpret_cast<const half8*>(As[PAR] + (wm * (CBM / 2) + i * 32 + r) * LS + kk * 16 + hi * 8);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[j] = *reinterpret_cast<const half8*>(Ws[PAR] + (wn * (CBN / 2) + j * 32 + r) * LS + kk * 16 + hi * 8);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* Af = reinterpret_cast<const float*>(As[PAR]);
            const float* Wf = reinterpret_cast<const float*>(Ws[PAR]);
#pragma unroll 4
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = Af[(wm * (CBM / 2) + i * 32 + r) * LS + kk * 2 + hi];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = Wf[(wn * (CBN / 2) + j * 32 + r) * LS + kk * 2 + hi];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        if (k0 + BK < Ktot) store_slab(1 - PAR, Other{});   // that buffer was last read before the previous barrier
        __syncthreads();
    };
    for (int k0 = 0; k0 < Ktot; k0 += 2 * BK) {
        body(k0, I0{});
        if (k0 + BK < Ktot) body(k0 + BK, I1{});
    }

    // ---- epilogue: bias (folded BN shift), residual, ReLU, NHWC store with channel offset / deconv scatter ----
    // The weight rows of a 32-channel tile sit in LDS in the order conv_row_channel (see wrow): MFMA output row
    // 8 q + 4 hi + e of lane (r, hi) is then channel 16 (q >> 1) + 8 hi + 4 (q & 1) + e, i.e. accumulator elements
    // 8 qq .. 8 qq + 7 are the eight consecutive channels 16 qq + 8 hi .. + 7 of pixel m0 + wm * (CBM / 2) + i * 32 + r:
    // one 16-byte store per run (8-byte stores are issue-bound, and this kernel is short of issue slots).
    const int s = p.deconv_s;
    const bool vec_ok = (p.Cout % 8 == 0) && (p.y_coff % 8 == 0) && (p.y_ctot % 8 == 0);
    float ymax = 0.f;                      // max |y| of this lane's stores (ConvParams::y_absmax)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (CBM / 2) + i * 32 + r;
        if (m >= M) continue;
        const int n = m / (p.Ho * p.Wo), rem = m - n * p.Ho * p.Wo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int col = n0 + wn * (CBN / 2) + j * 32 + 16 * qq + 8 * hi;
                if (col >= Ncols) continue;
                if (vec_ok && col + 7 < Ncols) {
                    const int sub = s ? col / p.Cout : 0, co = s ? col - sub * p.Cout : col;
                    const int dy = s ? sub / s : 0, dx = s ? sub - dy * s : 0;
                    const size_t pix = s ? ((size_t)n * p.Ho * s + oy * s + dy) * (p.Wo * s) + ox * s + dx : (size_t)m;
                    const size_t o = pix * p.y_ctot + p.y_coff + co;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = SPLIT ? acc[i][j][8 * qq + e] * s_inv : acc[i][j][8 * qq + e];
                    if (p.bias) {
                        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if (p.res) {
                        const T* rp = reinterpret_cast<const T*>(p.res) + pix * p.Cout + co;
                        if constexpr (sizeof(T) == 2) {
                            const half8 rv = *reinterpret_cast<const half8*>(rp);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += (float)rp[e];
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ymax = fmaxf(ymax, fabsf(v[e]));
                    }
                    if (p.out_f32 || sizeof(T) == 4) {
                        float* yp = reinterpret_cast<float*>(p.y) + o;
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        half8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
                        *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.y) + o) = h;
                    }
                } else {
                    // ragged channel count: element stores (the channels of one `sub` block stay together only when
                    // Cout is a multiple of 8, so recompute the scatter per element)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ce = col + e;
                        if (ce >= Ncols) continue;
                        const int sube = s ? ce / p.Cout : 0, coe = s ? ce - sube * p.Cout : ce;
                        const int dye = s ? sube / s : 0, dxe = s ? sube - dye * s : 0;
                        const size_t pixe = s ? ((size_t)n * p.Ho * s + oy * s + dye) * (p.Wo * s) + ox * s + dxe : (size_t)m;
                        const size_t oe = pixe * p.y_ctot + p.y_coff + coe;
                        float ve = (SPLIT ? acc[i][j][8 * qq + e] * s_inv : acc[i][j][8 * qq + e]) + (p.bias ? p.bias[coe] : 0.f);
                        if (p.res) ve += (float)reinterpret_cast<const T*>(p.res)[pixe * p.Cout + coe];
                        if (p.relu) ve = fmaxf(ve, 0.f);
                        if constexpr (SPLIT) ymax = fmaxf(ymax, fabsf(ve));
                        if (p.out_f32) reinterpret_cast<float*>(p.y)[oe] = ve;
                        else reinterpret_cast<T*>(p.y)[oe] = (T)ve;
                    }
                }
            }
    }
    if constexpr (SPLIT) {
        if (p.y_absmax) {
            absmax_raise_wg(p.y_absmax, ymax, amax_seen, reinterpret_cast<float*>(As[0]));
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_conv3: 3 x 3 / stride 1 / pad 1 convolutions in f16 (most layers of both backbones).  k_conv fetches every input
// pixel once per tap, and the bytes it pulls through the texture addresser are what bounds it; here a workgroup owns an
// 8 x 16 block of output pixels of one image, stages the (8 + 2) x (16 + 2) input patch of a 64-channel slab in LDS ONCE
// and lets the nine taps read their shifted windows out of it (the MFMA B fragment of pixel (py, px) and tap (ky, kx) is
// patch row (py + ky) * 18 + px + kx).  Per tap only the 64-deep weight slab travels (double-buffered as in k_conv, one tap
// ahead in registers); the patch of the next channel slab is requested at tap 0 and stored after tap 8.  Same MFMA
// tiling (2 x 2 waves, 64 pixels x CBN / 2 channels each), same permuted weight rows and 16-byte epilogue as k_conv.
// Needs Cin % 64 == 0, Cout % 8 == 0 (vector epilogue), no deconvolution; the decoder's nearest x2 upsampling of the input
// is an addressing mode of the patch gather.
// ------------------------------------------------------------------------------------------
// SPLIT: f32 map / weights / result, products on split-f16 operands as in k_conv<float, ..., SPLIT>: a channel slab is 32 deep, a
// staged row (patch pixel or weight row) holds [32 hi | 32 lo | pad] halves - the same 72-half rows, LDS footprint and
// fragment addressing as the f16 kernel; a 16-byte piece is 4 floats instead of 8 halves.
template <int CBN, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void k_conv3(ConvParams p) {
    using T = half_t;                                            // LDS element
    using TG = typename std::conditional<SPLIT, float, half_t>::type;   // element in global memory
    constexpr int BK = SPLIT ? 32 : 64, PE = 16 / (int)sizeof(TG);      // channels per slab, elements per 16-byte piece
    constexpr int LS = 72, TH = 8, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 180 patch pixels
    constexpr int RPW = CBN / 32, NJ = CBN / 64, MI = 2;
    constexpr int NPP = (NPIX * 8 + 255) / 256;           // 16-byte patch pieces per thread (6, the last one partial)
    __shared__ __attribute__((aligned(16))) T Ps[NPIX * LS];
    __shared__ __attribute__((aligned(16))) T Ws[2][CBN * LS];

    const int Ktot = 9 * p.Cin;
    const int tiles_n = (p.Cout + CBN - 1) / CBN, tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int tm = blockIdx.x / tiles_n;
    const int tn = blockIdx.x - tm * tiles_n, n0 = tn * CBN;
    const int tx = tm % tiles_x; tm /= tiles_x;
    const int ty = tm % tiles_y, n = tm / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, hi = lane >> 5;
    const TG* x = reinterpret_cast<const TG*>(p.x);
    const TG* w = reinterpret_cast<const TG*>(p.w);
    const int4v rs_x = conv_rsrc(x, (size_t)p.N * p.H * p.W * p.Cin * sizeof(TG) >> (p.up2 ? 2 : 0)), rs_w = conv_rsrc(w, (size_t)p.Cout * Ktot * sizeof(TG));

    // patch pieces of this thread: byte offset of (pixel, 16-byte chunk) for channel slab 0, out-of-map pixels read zeros
    unsigned poff[NPP];
    int plds[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int q = tid + 256 * i, pp = stage_row<SPLIT>(q), ch = q & 7;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = oy0 + pr - 1, ix = ox0 + pc - 1;
        const bool ok = pp < NPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        // up2: logical pixel (iy, ix) of the nearest-upsampled map is physical pixel (iy / 2, ix / 2) of the half-size input
        const int pixel = p.up2 ? (n * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + (ix >> 1) : (n * p.H + iy) * p.W + ix;
        poff[i] = ok ? (unsigned)(pixel * p.Cin + ch * PE) * (unsigned)sizeof(TG) : 0xffffffffu;
        plds[i] = pp < NPIX ? pp * LS + ch * PE : -1;
    }
    unsigned wrow[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int c = tid + 256 * i, row = stage_row<SPLIT>(c);
        const int nrow = n0 + (row & ~31) + conv_row_channel(row & 31);
        wrow[i] = nrow < p.Cout ? (unsigned)(((size_t)nrow * Ktot + (c & 7) * PE) * sizeof(TG)) : 0xffffffffu;
    }
    // patch row (in halves) of the two 32-pixel tiles of this wave: pixel wm * 64 + i * 32 + r of the 8 x 16 block
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int pix = wm * 64 + i * 32 + r;
        pbase[i] = ((pix >> 4) * PW + (pix & 15)) * LS + hi * 8;
    }

    float16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // split mode: range normalisation of both operands (ConvParams::absmax)
    float sx = 1.f, sw = 1.f, s_inv = 1.f;
    if constexpr (SPLIT) {
        if (p.x_absmax) {
            sx = pow2_scale(__uint_as_float(p.x_absmax[0]));
            if (p.w_absmax < 0.f) {            // the weights arrive pre-multiplied by the power of two -w_absmax: nothing to do at staging
                s_inv = pow2_inv(sx) * pow2_inv(-p.w_absmax);
            } else {
                sw = pow2_scale(p.w_absmax > 0.f ? p.w_absmax : __uint_as_float(p.x_absmax[1]));
                s_inv = pow2_inv(sx) * pow2_inv(sw);
            }
        }
    }
    const bool w_mul = sw != 1.f;
    const unsigned amax_seen = SPLIT ? absmax_peek(p.y_absmax) : 0u;
    int4v rp[NPP], rw[RPW];                        // 16-byte pieces: 8 halves, or 4 floats (SPLIT)
    auto put = [&](T* dst, const int4v& piece, float sc) {   // one piece into its LDS row (SPLIT: scaled, as hi / lo halves)
        if constexpr (SPLIT) {
            const float4v f = __builtin_bit_cast(float4v, piece) * sc;
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)f[e];
                l[e] = (half_t)(f[e] - (float)h[e]);
            }
            *reinterpret_cast<half4*>(dst) = h;
            *reinterpret_cast<half4*>(dst + 32) = l;
        } else {
            *reinterpret_cast<half8*>(dst) = __builtin_bit_cast(half8, piece);
        }
    };
    auto put_plain = [&](T* dst, const int4v& piece) {      // the same without the scale (pre-scaled weights)
        if constexpr (SPLIT) {
            const float4v f = __builtin_bit_cast(float4v, piece);
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)f[e];
                l[e] = (half_t)(f[e] - (float)h[e]);
            }
            *reinterpret_cast<half4*>(dst) = h;
            *reinterpret_cast<half4*>(dst + 32) = l;
        } else {
            *reinterpret_cast<half8*>(dst) = __builtin_bit_cast(half8, piece);
        }
    };
    auto load_patch = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            rp[i] = llvm_raw_buffer_load_b128(rs_x, (int)(poff[i] == 0xffffffffu ? poff[i] : poff[i] + (unsigned)ci0 * (unsigned)sizeof(TG)), 0, 0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            if (plds[i] >= 0) put(Ps + plds[i], rp[i], sx);
    };
    auto load_w = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
            rw[i] = llvm_raw_buffer_load_b128(rs_w, (int)(wrow[i] == 0xffffffffu ? wrow[i] : wrow[i] + (unsigned)k0 * (unsigned)sizeof(TG)), 0, 0);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int c = tid + 256 * i;
            if (w_mul) put(Ws[buf] + stage_row<SPLIT>(c) * LS + (c & 7) * PE, rw[i], sw);
            else put_plain(Ws[buf] + stage_row<SPLIT>(c) * LS + (c & 7) * PE, rw[i]);
        }
    };

    load_patch(0);
    load_w(0);
    store_patch();
    store_w(0);
    __syncthreads();
    int cur = 0;
    for (int ci0 = 0; ci0 < p.Cin; ci0 += BK) {
        const bool more_slabs = ci0 + BK < p.Cin;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool more = tap < 8 || more_slabs;
            if (tap == 0 && more_slabs) load_patch(ci0 + BK);                     // lands during the nine taps
            if (more) load_w(tap < 8 ? (tap + 1) * p.Cin + ci0 : ci0 + BK);      // next tap's weight slab
            const int tapoff = ((tap / 3) * PW + (tap % 3)) * LS;                 // compile-time: the loop is unrolled
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                half8 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const half8*>(Ps + pbase[i] + tapoff + kk * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    b[j] = *reinterpret_cast<const half8*>(Ws[cur] + (wn * (CBN / 2) + j * 32 + r) * LS + kk * 16 + hi * 8);
                if constexpr (SPLIT) {
                    half8 al[MI], bl[NJ];
#pragma unroll
                    for (int i = 0; i < MI; ++i) al[i] = *reinterpret_cast<const half8*>(Ps + pbase[i] + tapoff + kk * 16 + 32);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        bl[j] = *reinterpret_cast<const half8*>(Ws[cur] + (wn * (CBN / 2) + j * 32 + r) * LS + kk * 16 + hi * 8 + 32);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], a[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], al[i], acc[i][j], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
            if (more) store_w(cur ^ 1);      // that buffer was last read before the previous barrier
            __syncthreads();
            cur ^= 1;
            if (tap == 8 && more_slabs) {    // every wave is done with the patch: replace it
                store_patch();
                __syncthreads();
            }
        }
    }

    // ---- epilogue (k_conv's vector path): lane (r, hi) owns pixel wm * 64 + i * 32 + r and channels 16 qq + 8 hi .. + 7 ----
    float ymax = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int pix = wm * 64 + i * 32 + r;
        const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int co = n0 + wn * (CBN / 2) + j * 32 + 16 * qq + 8 * hi;
                if (co >= p.Cout) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = SPLIT ? acc[i][j][8 * qq + e] * s_inv : acc[i][j][8 * qq + e];
                if (p.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (p.res) {
                    const TG* rp8 = reinterpret_cast<const TG*>(p.res) + opix * p.Cout + co;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rp8[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (SPLIT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ymax = fmaxf(ymax, fabsf(v[e]));
                }
                const size_t o = opix * p.y_ctot + p.y_coff + co;
                if (p.out_f32 || SPLIT) {
                    float* yp = reinterpret_cast<float*>(p.y) + o;
                    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
                    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.y) + o) = h;
                }
            }
    }
    if constexpr (SPLIT) {
        if (p.y_absmax) {
            absmax_raise_wg(p.y_absmax, ymax, amax_seen, reinterpret_cast<float*>(Ps));
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_conv3r: k_conv3 with the weight slabs on an LDS-DMA ring.
// k_conv3 spends a tap like this: ~0.35 us of matrix work per wavefront, then every wavefront converts / stores the next weight
// slab from registers and meets the others at a barrier - and that slab was requested only ONE tap earlier, so on the deep layers
// (72 taps per workgroup, one or two workgroups per CU) a tap lasts 1.2 - 1.5 us whatever the precision (round-4 kernel trace:
// 256 -> 256 channels on 64 x 64 x 5 maps, 110 us split / 44 us f16 for 24 GF).  Here the weights exist once as an IMAGE of what
// the ring holds (conv3_pack below: per (channel tile, slab) CBN rows x 128 bytes = [32 hi | 32 lo] halves in split mode, 64
// halves in f16, rows in conv_row_channel order, 16-byte pieces XOR-swizzled by (row >> 1) & 7 so that rows 128 bytes apart
// read conflict-free).  A slab travels global -> LDS by global_load_lds_dwordx4, 1 KB per instruction and wavefront, no staging
// registers, no conversion, requested TWO taps ahead into a three-slot ring; a wavefront confirms only its own pieces of the
// next slab (counted vmcnt) before the tap's single barrier.  The input patch is k_conv3's (registers -> split -> LDS, once per
// channel slab); its loads are compiler-visible, so hipcc puts a full vmcnt(0) in front of their use: at a channel slab's last
// tap the slab two ahead is therefore requested AFTER the patch has been stored (it flies one tap instead of two).
// Same MFMA order as k_conv3 -> bit-identical results (tests/test_hip_encoder.py).
// ------------------------------------------------------------------------------------------
template <int CBN, bool SPLIT>
struct Conv3rCfg {
    static constexpr int BK = SPLIT ? 32 : 64;
    static constexpr int SLAB_HALVES = CBN * 64;          // CBN rows x 128 bytes
    static constexpr int NSLOT = 3;
    static constexpr int PPW = CBN / 32;                  // 1 KB DMA pieces per wavefront and slab (4 wavefronts)
};
__host__ __device__ inline size_t conv3_image_bytes(int Cout, int Cin, bool split) {
    const int cbn = Cout <= 64 ? 64 : 128, bk = split ? 32 : 64;
    return (size_t)((Cout + cbn - 1) / cbn) * 9 * (Cin / bk) * cbn * 128;
}
// kind 1: the same slabs in the GEMM's own k order (slab g = columns 32 g .. 32 g + 31 of the (Ncols, Ktot) weight matrix), for the
// generic implicit-GEMM kernel k_conv<..., SPLIT, RING>: strided / 1 x 1 / transposed convolutions
__host__ __device__ inline size_t conv_gemm_image_bytes(int Ncols, int Ktot) {
    const int cbn = Ncols <= 64 ? 64 : 128;
    return (size_t)((Ncols + cbn - 1) / cbn) * (Ktot / 32) * cbn * 128;
}
// one thread per 16-byte piece of the image
// gemm_ktot > 0: kind 1 (Cout = Ncols rows of gemm_ktot columns, slabs in k order; split only)
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_conv3_pack(const void* __restrict__ w_, half_t* __restrict__ image, int Cout, int Cin, int gemm_ktot) {
    const int CBN = Cout <= 64 ? 64 : 128, BK = SPLIT ? 32 : 64;
    const int G = gemm_ktot ? gemm_ktot / BK : 9 * (Cin / BK), Ktot = gemm_ktot ? gemm_ktot : 9 * Cin;
    const size_t n_pieces = (size_t)((Cout + CBN - 1) / CBN) * G * CBN * 8;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_pieces) return;
    const int slot = idx & 7, row = (int)((idx >> 3) % CBN);
    const size_t sl = (idx >> 3) / CBN;
    const int g = (int)(sl % G), tile = (int)(sl / G);
    const int pc = slot ^ ((row >> 1) & 7);               // the piece that sits in this slot
    const int nrow = tile * CBN + (row & ~31) + conv_row_channel(row & 31);
    const int cs = g / 9, tap = g - cs * 9;
    half8 out = (half8)(half_t)0.f;
    if (nrow < Cout) {
        if constexpr (SPLIT) {
            const float* src = reinterpret_cast<const float*>(w_) + (size_t)nrow * Ktot + (gemm_ktot ? g * BK : tap * Cin + cs * BK) + (pc & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = src[e];
                const half_t h = (half_t)v;
                out[e] = (pc & 4) ? (half_t)(v - (float)h) : h;
            }
        } else {
            out = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(w_) + (size_t)nrow * Ktot + tap * Cin + cs * BK + pc * 8);
        }
    }
    *reinterpret_cast<half8*>(image + idx * 8) = out;
}
size_t conv3_image_size(int Cout, int Cin, int precision) { return conv3_image_bytes(Cout, Cin, precision == HMVIT_PREC_SPLIT); }
int launch_conv3_pack(const void* w, int Cout, int Cin, int precision, void* image, hipStream_t st) {
    const bool split = precision == HMVIT_PREC_SPLIT;
    HMVIT_CHECK_ARG(split || precision == HMVIT_PREC_F16, "conv3 image: split or f16 (precision %d)", precision);
    HMVIT_CHECK_ARG(w && image && Cout > 0 && Cin > 0 && Cin % (split ? 32 : 64) == 0, "conv3 image: Cout=%d Cin=%d", Cout, Cin);
    const size_t n_pieces = conv3_image_bytes(Cout, Cin, split) / 16;
    if (split) hipLaunchKernelGGL(k_conv3_pack<true>, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<half_t*>(image), Cout, Cin, 0);
    else hipLaunchKernelGGL(k_conv3_pack<false>, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<half_t*>(image), Cout, Cin, 0);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
size_t conv_gemm_image_size(int Ncols, int Ktot) { return conv_gemm_image_bytes(Ncols, Ktot); }
int launch_conv_gemm_pack(const float* w, int Ncols, int Ktot, void* image, hipStream_t st) {
    HMVIT_CHECK_ARG(w && image && Ncols > 0 && Ktot > 0 && Ktot % 32 == 0, "conv image (GEMM order): Ncols=%d Ktot=%d", Ncols, Ktot);
    const size_t n_pieces = conv_gemm_image_bytes(Ncols, Ktot) / 16;
    hipLaunchKernelGGL(k_conv3_pack<true>, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<half_t*>(image), Ncols, 0, Ktot);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// STRIDE = 2: the 3 x 3 / stride 2 / pad 1 layers (the first layer of every backbone block, the shrink header's 384 -> 256 layer)
// with the same tile and ring: the patch is the 17 x 33 input pixels under the 8 x 16 outputs (81 KB per channel slab: one
// workgroup per CU, 18 pieces per thread in flight), output pixel (py, px) and tap (ky, kx) read patch pixel (2 py + ky, 2 px + kx).
// The generic kernel fetches every input pixel 2.25 times per channel slab through the address path for these layers.
template <int CBN, bool SPLIT = false, int STRIDE = 1>
__global__ __launch_bounds__(256, STRIDE == 1 ? 2 : 1) void k_conv3r(ConvParams p) {
    using T = half_t;
    using TG = typename std::conditional<SPLIT, float, half_t>::type;
    using Cfg = Conv3rCfg<CBN, SPLIT>;
    constexpr int BK = Cfg::BK, PE = 16 / (int)sizeof(TG);
    constexpr int LS = 72, TH = 8, TW = 16, PW = TW * STRIDE + 3 - STRIDE, NPIX = (TH * STRIDE + 3 - STRIDE) * PW;
    constexpr int NJ = CBN / 64, MI = 2, PPW = Cfg::PPW, NSLOT = Cfg::NSLOT, SLAB = Cfg::SLAB_HALVES;
    constexpr int NPP = (NPIX * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) T Ps[NPIX * LS];
    __shared__ __attribute__((aligned(1024))) T ring[NSLOT * SLAB];

    const int NCS = p.Cin / BK, G = 9 * NCS;
    const int tiles_n = (p.Cout + CBN - 1) / CBN, tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int tm = blockIdx.x / tiles_n;
    const int tn = blockIdx.x - tm * tiles_n, n0 = tn * CBN;
    const int tx = tm % tiles_x; tm /= tiles_x;
    const int ty = tm % tiles_y, n = tm / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, hi = lane >> 5;
    const TG* x = reinterpret_cast<const TG*>(p.x);
    const int4v rs_x = conv_rsrc(x, (size_t)p.N * p.H * p.W * p.Cin * sizeof(TG) >> (p.up2 ? 2 : 0));

    unsigned poff[NPP];
    int plds[NPP];
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int q = tid + 256 * i, pp = stage_row<SPLIT>(q), ch = q & 7;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = oy0 * STRIDE + pr - 1, ix = ox0 * STRIDE + pc - 1;
        const bool ok = pp < NPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int pixel = p.up2 ? (n * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + (ix >> 1) : (n * p.H + iy) * p.W + ix;
        poff[i] = ok ? (unsigned)(pixel * p.Cin + ch * PE) * (unsigned)sizeof(TG) : 0xffffffffu;
        plds[i] = pp < NPIX ? pp * LS + ch * PE : -1;
    }
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int pix = wm * 64 + i * 32 + r;
        pbase[i] = ((pix >> 4) * STRIDE * PW + (pix & 15) * STRIDE) * LS + hi * 8;
    }
    // weight fragments: row wn * (CBN / 2) + j * 32 + r of the slab, piece kk * 2 + hi (+ 4 for the low halves), swizzled
    int wrow_h[NJ], wkey[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = wn * (CBN / 2) + j * 32 + r;
        wrow_h[j] = row * 64;
        wkey[j] = (row >> 1) & 7;
    }

    float16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float sx = 1.f, s_inv = 1.f;
    if constexpr (SPLIT) {
        if (p.x_absmax) {
            sx = pow2_scale(__uint_as_float(p.x_absmax[0]));
            s_inv = pow2_inv(sx) * pow2_inv(-p.w_absmax);      // the image holds weights pre-multiplied by the power of two -w_absmax
        }
    }
    const unsigned amax_seen = SPLIT ? absmax_peek(p.y_absmax) : 0u;
    int4v rp[NPP];
    auto put = [&](T* dst, const int4v& piece, float sc) {
        if constexpr (SPLIT) {
            const float4v f = __builtin_bit_cast(float4v, piece) * sc;
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)f[e];
                l[e] = (half_t)(f[e] - (float)h[e]);
            }
            *reinterpret_cast<half4*>(dst) = h;
            *reinterpret_cast<half4*>(dst + 32) = l;
        } else {
            *reinterpret_cast<half8*>(dst) = __builtin_bit_cast(half8, piece);
        }
    };
    auto load_patch = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            rp[i] = llvm_raw_buffer_load_b128(rs_x, (int)(poff[i] == 0xffffffffu ? poff[i] : poff[i] + (unsigned)ci0 * (unsigned)sizeof(TG)), 0, 0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            if (plds[i] >= 0) put(Ps + plds[i], rp[i], sx);
    };
    // slab g of this channel tile -> ring slot g % NSLOT: this wavefront's PPW pieces of 1 KB (piece i * 4 + wave)
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring;
    const char* img = reinterpret_cast<const char*>(p.w_image) + (size_t)tn * G * (SLAB * 2) + lane * 16;
    auto request = [&](int g) {
#ifdef HMVIT_EXP_C3_NODMA
        return;
#endif
        const char* src = img + (size_t)g * (SLAB * 2);
        const unsigned dst0 = ring_lds + (unsigned)(g % NSLOT) * (SLAB * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const char* a = src + (i * 4 + wave) * 1024;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (i * 4 + wave) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(a), "s"(dst) : "memory");
        }
    };
    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef HMVIT_EXP_C3_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    // "all but the newest n vector-memory operations of this wavefront are complete" (requests and patch loads share the in-order counter)
    auto confirm = [&](bool requested, bool patch_in_flight) {
        if (!requested) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (!patch_in_flight) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW + NPP) : "memory");
    };

    request(0);
    if (G > 1) request(1);
    load_patch(0);
    store_patch();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    int g = 0;
    for (int cs = 0; cs < NCS; ++cs) {
        const bool more_slabs = cs + 1 < NCS;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++g) {
            const bool defer = tap == 8 && more_slabs;      // the patch is replaced at the end of this tap: request after that
            const bool req = g + 2 < G && !defer;
            if (tap == 0 && more_slabs) load_patch((cs + 1) * BK);
            const T* Wc = ring + (g % NSLOT) * SLAB;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) * LS;
            // every fragment of the tap first (16 ds_read_b128 in split mode), then its 24 matrix instructions: a wavefront that
            // is alone on its SIMD - the deep layers leave most CUs one workgroup - waits for LDS once per tap instead of per k-step
            constexpr int NKK = BK / 16;
            half8 a[NKK][MI], b[NKK][NJ], al[SPLIT ? NKK : 1][MI], bl[SPLIT ? NKK : 1][NJ];
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kk][i] = *reinterpret_cast<const half8*>(Ps + pbase[i] + tapoff + kk * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[kk][j] = *reinterpret_cast<const half8*>(Wc + wrow_h[j] + (((kk * 2 + hi) ^ wkey[j]) << 3));
                if constexpr (SPLIT) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) al[kk][i] = *reinterpret_cast<const half8*>(Ps + pbase[i] + tapoff + kk * 16 + 32);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bl[kk][j] = *reinterpret_cast<const half8*>(Wc + wrow_h[j] + (((kk * 2 + hi + 4) ^ wkey[j]) << 3));
                }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                // the request of the slab two taps ahead goes out in the middle of the tap's matrix work, not in front of it: four
                // DMA instructions per wavefront back to back stall on the address path (7 ns each per CU), and at the head of a
                // tap nothing of this wavefront is running yet that the stall could hide behind (-3 ... -7 % per layer)
                if (kk == NKK / 2 && req) request(g + 2);
                if constexpr (SPLIT) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], al[kk][i], acc[i][j], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
            }
            // this wavefront's pieces of slab g + 1; what it has issued since: the request of this tap, and - in the first tap of a
            // channel slab - the six patch loads before it (they are older than the request confirmed at the end of tap 1: two taps)
#ifndef HMVIT_EXP_C3_NOWAIT
            if (g + 1 < G) confirm(req, tap == 0 && more_slabs);
#endif
            barrier();
            if (defer) {                                  // every wavefront is done with the patch: replace it, then the deferred request
                store_patch();
                if (g + 2 < G) request(g + 2);
                barrier();
            }
        }
    }

    // ---- epilogue (k_conv's vector path): lane (r, hi) owns pixel wm * 64 + i * 32 + r and channels 16 qq + 8 hi .. + 7 ----
    float ymax = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int pix = wm * 64 + i * 32 + r;
        const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int co = n0 + wn * (CBN / 2) + j * 32 + 16 * qq + 8 * hi;
                if (co >= p.Cout) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = SPLIT ? acc[i][j][8 * qq + e] * s_inv : acc[i][j][8 * qq + e];
                if (p.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (p.res) {
                    const TG* rp8 = reinterpret_cast<const TG*>(p.res) + opix * p.Cout + co;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rp8[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (SPLIT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ymax = fmaxf(ymax, fabsf(v[e]));
                }
                const size_t o = opix * p.y_ctot + p.y_coff + co;
                if (p.out_f32 || SPLIT) {
                    float* yp = reinterpret_cast<float*>(p.y) + o;
                    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
                    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.y) + o) = h;
                }
            }
    }
    if constexpr (SPLIT) {
        if (p.y_absmax) {
            absmax_raise_wg(p.y_absmax, ymax, amax_seen, reinterpret_cast<float*>(Ps));
        }
    }
}

// ------------------------------------------------------------------------------------------
// max pooling, NHWC, 8 channels per thread (padding never wins: torch pads with -inf)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                                                 int ks, int stride, int pad) {
    const int cg = C / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * Ho * Wo * cg) return;
    const int c8 = (int)(idx % cg) * 8;
    size_t r = idx / cg;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const T* src = x + (((size_t)n * H + iy) * W + ix) * C + c8;
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], (float)src[e]);
        }
    T* dst = y + (((size_t)n * Ho + oy) * Wo + ox) * C + c8;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = (T)best[e];
}

int launch_maxpool(const void* x, void* y, int N, int H, int W, int C, int ksize, int stride, int pad, int precision, hipStream_t st) {
    HMVIT_CHECK_ARG(C % 8 == 0 && ksize > 0 && stride > 0, "maxpool: C=%d must be a multiple of 8", C);
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const size_t n = (size_t)N * Ho * Wo * (C / 8);
    if (n == 0) return HMVIT_OK;
    if (precision != HMVIT_PREC_F16)
        hipLaunchKernelGGL((k_maxpool<float>), dim3((unsigned)cdiv((long long)n, 256)), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho, Wo, ksize, stride, pad);
    else
        hipLaunchKernelGGL((k_maxpool<half_t>), dim3((unsigned)cdiv((long long)n, 256)), dim3(256), 0, st, (const half_t*)x, (half_t*)y, N, H, W, C, Ho, Wo, ksize, stride, pad);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// Adjoint of nn.MaxPool2d on f32 NHWC maps, as a GATHER (deterministic, no atomics): input pixel (iy, ix) collects dy of every
// window it lies in AND whose maximum it is - the first maximum in the window's row-major scan, the element torch's backward routes to.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                     int H, int W, int C, int Ho, int Wo, int ks, int stride, int pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * H * W * C) return;
    const int c = (int)(idx % C);
    size_t r = idx / C;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    const float* xn = x + (size_t)n * H * W * C + c;
    const float mine = xn[((size_t)iy * W + ix) * C];
    float acc = 0.f;
    // output rows whose window covers iy: oy * stride - pad <= iy < oy * stride - pad + ks
    const int oy_hi = min((iy + pad) / stride, Ho - 1), ox_hi = min((ix + pad) / stride, Wo - 1);
    for (int oy = oy_hi; oy >= 0 && oy * stride - pad + ks > iy; --oy)
        for (int ox = ox_hi; ox >= 0 && ox * stride - pad + ks > ix; --ox) {
            bool is_arg = true;
            for (int ky = 0; ky < ks && is_arg; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    const int yy = oy * stride + ky - pad, xx = ox * stride + kx - pad;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const float v = xn[((size_t)yy * W + xx) * C];
                    const bool before = yy < iy || (yy == iy && xx < ix);
                    if (v > mine || (before && v == mine)) { is_arg = false; break; }
                }
            if (is_arg) acc += dy[(((size_t)n * Ho + oy) * Wo + ox) * C + c];
        }
    dx[idx] = acc;
}
int launch_maxpool_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int ksize, int stride, int pad, hipStream_t st) {
    HMVIT_CHECK_ARG(ksize > 0 && stride > 0, "maxpool_bwd: bad kernel / stride");
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const size_t n = (size_t)N * H * W * C;
    if (n == 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)cdiv((long long)n, 256)), dim3(256), 0, st, x, dy, dx, N, H, W, C, Ho, Wo, ksize, stride, pad);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// max |a| as an f32 bit pattern (non-negative floats order like unsigned integers), atomicMax into slot[which(blockIdx.y)]:
// the range information of the split-mode convolutions when their caller did not provide it
__global__ __launch_bounds__(256) void k_absmax2(const float* __restrict__ x, size_t nx, const float* __restrict__ w, size_t nw,
                                                 unsigned* __restrict__ slot) {
    const bool is_w = blockIdx.y != 0;
    const float* a = is_w ? w : x;
    const size_t n = is_w ? nw : nx, n4 = n / 4;
    if (a == nullptr || n == 0) return;
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {            // four 16-byte loads in flight per thread
        const float4 v0 = reinterpret_cast<const float4*>(a)[i], v1 = reinterpret_cast<const float4*>(a)[i + stride];
        const float4 v2 = reinterpret_cast<const float4*>(a)[i + 2 * stride], v3 = reinterpret_cast<const float4*>(a)[i + 3 * stride];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                           fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                           fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
    }
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(a)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(a[n4 * 4 + threadIdx.x]));
    // NaN / Inf inputs: fmaxf drops NaN; an Inf maximum gives the smallest scale and the Inf propagates through the products
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) absmax_raise(slot + (is_w ? 1 : 0), m, 0u);
}
int launch_absmax(const float* x, size_t n, unsigned* slot, hipStream_t st) {
    if (n == 0) return HMVIT_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (n / 16 + 255) / 256));
    hipLaunchKernelGGL(k_absmax2, dim3(blocks, 1), dim3(256), 0, st, x, n, nullptr, (size_t)0, slot);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
// ring of result slots of the fallback pass: a launch takes the next one (stream-ordered use: zeroed, filled, read by its own
// three stream operations)
constexpr int kAbsmaxSlots = 4096;
__device__ unsigned g_conv_absmax[kAbsmaxSlots][2];

int launch_conv(const ConvParams& p_in, int precision, hipStream_t st) {
    ConvParams p = p_in;
    if (precision != HMVIT_PREC_SPLIT) {
        p.x_absmax = nullptr; p.y_absmax = nullptr;
    } else if (!p.x_absmax) {
        // no range information from the caller (hmvit_conv_range): measure max |x| (and max |w| unless given) here - one more pass
        // over the input; the convolutional modules of hm-vit_amd chain the information instead (pointpillar.py, decoder.py, camera.py)
        static std::atomic<unsigned> next_slot{0};
        static unsigned* bases[64] = {};      // per device: the symbol's address is looked up once (not during a graph capture)
        int dev = 0;
        HMVIT_CHECK_HIP(hipGetDevice(&dev));
        HMVIT_CHECK_ARG(dev >= 0 && dev < 64, "conv: device ordinal %d", dev);
        if (!bases[dev]) HMVIT_CHECK_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&bases[dev]), HIP_SYMBOL(g_conv_absmax)));
        unsigned* slot = bases[dev] + 2 * (next_slot.fetch_add(1) % kAbsmaxSlots);
        const size_t nx = ((size_t)p.N * p.H * p.W * p.Cin) >> (p.up2 ? 2 : 0);
        const int ncols = p.deconv_s ? p.deconv_s * p.deconv_s * p.Cout : p.Cout;
        const size_t nw = p.w_absmax != 0.f ? 0 : (size_t)ncols * (p.rowpack ? p.KH * 32 : p.KH * p.KW * p.Cin);
        HMVIT_CHECK_HIP(hipMemsetAsync(slot, 0, 2 * sizeof(unsigned), st));
        const unsigned blocks = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (std::max(nx, nw) / 16 + 255) / 256));
        hipLaunchKernelGGL(k_absmax2, dim3(blocks, nw ? 2 : 1), dim3(256), 0, st, reinterpret_cast<const float*>(p.x), nx,
                           reinterpret_cast<const float*>(p.w), nw, slot);
        HMVIT_CHECK_LAUNCH();
        p.x_absmax = slot;
    }
    const bool f32_maps = precision == HMVIT_PREC_F32 || precision == HMVIT_PREC_SPLIT;   // element type of x / w / residual
    const int bk = f32_maps ? ConvCfg<float>::BK : ConvCfg<half_t>::BK;
    HMVIT_CHECK_ARG(p.rowpack || (p.Cin > 0 && p.Cin % bk == 0), "conv: Cin=%d must be a multiple of %d", p.Cin, bk);
    HMVIT_CHECK_ARG(!p.rowpack || (p.Cin == 4 && (p.KH * 32) % bk == 0 && !p.deconv_s && !p.up2 &&
                                   (p.Ho - 1) * p.stride + p.KH <= p.H && (p.Wo - 1) * p.stride + 8 <= p.W),
                    "conv (row-packed stem): needs a 4-channel input padded to cover %d rows x 8 pixels per output", p.KH);
    HMVIT_CHECK_ARG(p.N > 0 && p.Ho > 0 && p.Wo > 0 && p.Cout > 0 && p.KH > 0 && p.KW > 0 && p.stride > 0,
                    "conv: bad geometry");
    HMVIT_CHECK_ARG(!p.deconv_s || (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0),
                    "deconv is expressed as a 1x1 GEMM with scatter");
    const int M = p.N * p.Ho * p.Wo;
    const int Ncols = p.deconv_s ? p.deconv_s * p.deconv_s * p.Cout : p.Cout;
    const bool narrow = Ncols <= 64;
    {   // the gathers address the input and the weights with 32-bit byte offsets (buffer loads)
        const size_t es = f32_maps ? 4 : 2;
        const size_t xb = (size_t)p.N * p.H * p.W * p.Cin * es, wb = (size_t)Ncols * (p.rowpack ? p.KH * 32 : p.KH * p.KW * p.Cin) * es;
        HMVIT_CHECK_ARG(xb < 0xfffffff0ull && wb < 0xfffffff0ull, "conv: input (%zu bytes) or weights (%zu bytes) exceed the 4 GB a launch can address",
                        xb, wb);
    }
    const bool split = precision == HMVIT_PREC_SPLIT;
    // 3 x 3 / stride 2 / pad 1 with a kind-0 weight image: the ring kernel on a 17 x 33 patch
    if ((!f32_maps || split) && p.KH == 3 && p.KW == 3 && p.stride == 2 && p.pad == 1 && !p.deconv_s && !p.rowpack && !p.up2 &&
        p.Cin % (split ? 32 : 64) == 0 && p.Cout % 8 == 0 && p.y_coff % 8 == 0 && p.y_ctot % 8 == 0 && !p.no_patch &&
        p.w_image && p.w_image_kind == 0 && (!split || p.w_absmax < 0.f) && !HMVIT_ENV("HMVIT_CONV_NO_RING")) {
        const int tiles = p.N * cdiv(p.Ho, 8) * cdiv(p.Wo, 16);
        // measured on the PointPillar layers (split): 384 -> 256 channels 591 -> 515 us; 64 -> 64, 64 -> 128 and 128 -> 256 channels no
        // better than the generic kernel with its own weight ring (185 / 72 / 67 -> 191 / 72 / 74 us: with few channel slabs the
        // 81 KB patch is replaced too often for what it saves) - so deep inputs only (tests: HMVIT_CONV_S2_ALL in probe builds)
        if (tiles * cdiv(p.Cout, narrow ? 64 : 128) >= 128 && (p.Cin >= 256 || p.force_patch || HMVIT_ENV("HMVIT_CONV_S2_ALL"))) {
            dim3 grid3(tiles * cdiv(p.Cout, narrow ? 64 : 128));
            if (split) {
                if (narrow) hipLaunchKernelGGL((k_conv3r<64, true, 2>), grid3, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((k_conv3r<128, true, 2>), grid3, dim3(256), 0, st, p);
            } else if (narrow) hipLaunchKernelGGL((k_conv3r<64, false, 2>), grid3, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_conv3r<128, false, 2>), grid3, dim3(256), 0, st, p);
            HMVIT_CHECK_LAUNCH();
            return HMVIT_OK;
        }
    }
    // 3 x 3 / stride 1 / pad 1 in f16: the patch-in-LDS kernel (one fetch per input pixel and channel slab instead of nine)
    if ((!f32_maps || split) && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && !p.deconv_s && !p.rowpack &&
        p.Cin % (split ? 32 : 64) == 0 && p.Cout % 8 == 0 && p.y_coff % 8 == 0 && p.y_ctot % 8 == 0 && p.Ho == p.H && p.Wo == p.W &&
        !p.no_patch && !HMVIT_ENV("HMVIT_CONV_NO_PATCH")) {
        const int tiles = p.N * cdiv(p.Ho, 8) * cdiv(p.Wo, 16);
        const bool ring = p.w_image && p.w_image_kind == 0 && (!split || p.w_absmax < 0.f) && !HMVIT_ENV("HMVIT_CONV_NO_RING");
        // enough workgroups to cover the CUs.  The ring kernel is taken from half a cover on: on the deep ResNet layers
        // (512 channels, 16 x 16 maps, 160 workgroups of 144 taps) it still beats the generic kernel's 64-pixel tiles 2 : 1
        if (tiles * cdiv(p.Cout, narrow ? 64 : 128) >= (ring ? 128 : 256)) {
            dim3 grid3(tiles * cdiv(p.Cout, narrow ? 64 : 128));
            // the weights also exist as a ring image (hmvit_conv3x3_image; split: of the pre-scaled weights): the LDS-DMA kernel
            if (ring) {
                if (split) {
                    if (narrow) hipLaunchKernelGGL((k_conv3r<64, true>), grid3, dim3(256), 0, st, p);
                    else hipLaunchKernelGGL((k_conv3r<128, true>), grid3, dim3(256), 0, st, p);
                } else if (narrow) hipLaunchKernelGGL((k_conv3r<64>), grid3, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((k_conv3r<128>), grid3, dim3(256), 0, st, p);
                HMVIT_CHECK_LAUNCH();
                return HMVIT_OK;
            }
            if (split) {
                if (narrow) hipLaunchKernelGGL((k_conv3<64, true>), grid3, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((k_conv3<128, true>), grid3, dim3(256), 0, st, p);
            } else if (narrow) hipLaunchKernelGGL((k_conv3<64>), grid3, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_conv3<128>), grid3, dim3(256), 0, st, p);
            HMVIT_CHECK_LAUNCH();
            return HMVIT_OK;
        }
    }
    // small maps (the deep ResNet layers on 16 x 16 features): 64-pixel tiles, or the launch would not cover the CUs
    const bool small = cdiv(M, 128) * cdiv(Ncols, narrow ? 64 : 128) < 256;
    dim3 grid(cdiv(M, small ? 64 : 128) * cdiv(Ncols, narrow ? 64 : 128)), block(256);
#define HMVIT_CONV_CASE(T, N_, M_) hipLaunchKernelGGL((k_conv<T, N_, M_>), grid, block, 0, st, p)
    if (precision == HMVIT_PREC_SPLIT) {
        // the weights also exist as the GEMM-order ring image (hmvit_conv_gemm_image, of the pre-scaled weights): weight slabs by LDS-DMA
        const bool gring = p.w_image && p.w_image_kind == 1 && p.w_absmax < 0.f && !p.rowpack && !HMVIT_ENV("HMVIT_CONV_NO_RING");
#define HMVIT_CONV_SPLIT(N_, M_) do { if (gring) hipLaunchKernelGGL((k_conv<float, N_, M_, true, true>), grid, block, 0, st, p); \
                                      else hipLaunchKernelGGL((k_conv<float, N_, M_, true>), grid, block, 0, st, p); } while (0)
        if (narrow) { if (small) HMVIT_CONV_SPLIT(64, 64); else HMVIT_CONV_SPLIT(64, 128); }
        else { if (small) HMVIT_CONV_SPLIT(128, 64); else HMVIT_CONV_SPLIT(128, 128); }
#undef HMVIT_CONV_SPLIT
    } else if (precision == HMVIT_PREC_F32) {
        if (narrow) { if (small) HMVIT_CONV_CASE(float, 64, 64); else HMVIT_CONV_CASE(float, 64, 128); }
        else { if (small) HMVIT_CONV_CASE(float, 128, 64); else HMVIT_CONV_CASE(float, 128, 128); }
    } else {
        if (narrow) { if (small) HMVIT_CONV_CASE(half_t, 64, 64); else HMVIT_CONV_CASE(half_t, 64, 128); }
        else { if (small) HMVIT_CONV_CASE(half_t, 128, 64); else HMVIT_CONV_CASE(half_t, 128, 128); }
    }
#undef HMVIT_CONV_CASE
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
