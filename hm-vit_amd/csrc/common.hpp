// Shared device/host helpers for libhmvit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/hmvit.h"

namespace hmvit {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
// x -> (hi, lo) f16 pair (a macro: vector elements cannot bind to references)
#define split_h(x, H, L)                     \
    do {                                     \
        const float _x = (x);                \
        const half_t _h = (half_t)_x;        \
        (H) = _h;                            \
        (L) = (half_t)(_x - (float)_h);      \
    } while (0)

// The same split for a PAIR of values on packed instructions (v_cvt_pk_f16_f32, one v_pk_add_f32 for the two residuals): five
// instructions per pair.  hi, lo: packed f16 pairs (element 0 in the low half).
// Round 5 first wrote this as three instructions of inline asm (v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16 forming
// x * 1.0 - hi straight into the halves of the destination; bit-identical to split_h in isolation, tools/probe/split2_probe.hip)
// and measured NO gain in the tails / 1 % in the attention - and k_attention_bwd, where the split is followed at once by the
// matrix instruction that reads it, produced wrong gradients with it in one schedule and right ones in another (the same source
// with a device-side comparison added passed): hipcc pads hazards between an asm block and an MFMA by its generic rule, not by
// what the block's partial-register writes need.  Plain C++ leaves the hazard bookkeeping to the compiler.
__device__ __forceinline__ void split_pk2(float a, float b, unsigned& hi, unsigned& lo) {
    const float2v x = {a, b};
    const half2v h = __builtin_convertvector(x, half2v);
    const half2v l = __builtin_convertvector(x - __builtin_convertvector(h, float2v), half2v);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_pk4(float a, float b, float c, float d, half4& h, half4& l) {
    unsigned h0, l0, h1, l1;
    split_pk2(a, b, h0, l0);
    split_pk2(c, d, h1, l1);
    h = __builtin_bit_cast(half4, uint2v{h0, h1});
    l = __builtin_bit_cast(half4, uint2v{l0, l1});
}
__device__ __forceinline__ void split_pk8(const float (&v)[8], half8& h, half8& l) {
    unsigned hh[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pk2(v[2 * i], v[2 * i + 1], hh[i], ll[i]);
    h = __builtin_bit_cast(half8, uint4v_{hh[0], hh[1], hh[2], hh[3]});
    l = __builtin_bit_cast(half8, uint4v_{ll[0], ll[1], ll[2], ll[3]});
}

// Tuning / ablation switches read from the environment exist only in probe builds (`make PROBE=1` -> -DHMVIT_PROBE, used by
// tools/probe and tests/tools): the shipped library never calls getenv and never takes a pointer out of the environment.
#ifdef HMVIT_PROBE
#include <stdlib.h>
#define HMVIT_ENV(name) getenv(name)
#else
#define HMVIT_ENV(name) ((const char*)nullptr)
#endif

// Timing-only ablation switches (results wrong by construction) and debug fingerprints must never reach a shipped library:
// they compile only together with -DHMVIT_PROBE or -DHMVIT_ALLOW_EXP (tools/probe/build_var.sh adds the latter: probe stamps off).  tests/test_host_cpu.py checks the Makefile's
// default flags carry none of them.
#if !defined(HMVIT_PROBE) && !defined(HMVIT_ALLOW_EXP) && (                                                                                          \
    defined(HMVIT_EXP_C3_NOBAR) || defined(HMVIT_EXP_C3_NODMA) || defined(HMVIT_EXP_C3_NOWAIT) || defined(HMVIT_EXP_DYN_ALL) || \
    defined(HMVIT_EXP_NOAMAX) || defined(HMVIT_EXP_NOBAR) || defined(HMVIT_EXP_NOMMA) || defined(HMVIT_EXP_NOSTORE) ||       \
    defined(HMVIT_EXP_NOWAIT) || defined(HMVIT_EXP_PCS_NOEPI) || defined(HMVIT_EXP_PCS_NOLOAD) ||                           \
    defined(HMVIT_EXP_PCS_NOLOADER) || defined(HMVIT_EXP_PCS_NOMATH) || defined(HMVIT_EXP_PCS_NOQ) ||                       \
    defined(HMVIT_EXP_PCS_NOSTORE) || defined(HMVIT_EXP_PCS_NOTABLES) || defined(HMVIT_EXP_STATIC_ITEMS) ||                 \
    defined(HMVIT_EXP_PATCH_NODMA) || defined(HMVIT_EXP_PATCH_NOBLEND) || defined(HMVIT_EXP_PATCH_NOMATH) || defined(HMVIT_EXP_PATCH_NOTABLES) || \
    defined(HMVIT_EXP_X16_NODMA) || defined(HMVIT_EXP_PCS_Q4) || defined(HMVIT_EXP_PCS_LPRIO) || defined(HMVIT_EXP_X16_NOSTORE) || defined(HMVIT_DBG_SUMS))
#error "HMVIT_EXP_* / HMVIT_DBG_* are timing / debug experiments: build them with -DHMVIT_ALLOW_EXP (tools/probe/build_var.sh), never into the shipped library"
#endif

// ---- thread-local error string (hmvit_last_error) ----
void set_error(const char* fmt, ...);
#define HMVIT_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            ::hmvit::set_error(__VA_ARGS__);  \
            return HMVIT_EINVAL;              \
        }                                     \
    } while (0)
#define HMVIT_CHECK_HIP(expr)                                                         \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::hmvit::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                   \
            return HMVIT_EHIP;                                                        \
        }                                                                             \
    } while (0)
#define HMVIT_CHECK_LAUNCH() HMVIT_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <typename T>
struct ElemOf;
template <>
struct ElemOf<float> {
    static constexpr int prec = HMVIT_PREC_F32;
};
template <>
struct ElemOf<half_t> {
    static constexpr int prec = HMVIT_PREC_F16;
};

// ---- sampling geometry shared by the warp operator and the attention gather ----
// ainv record: [a00 a01 a02 a10 a11 a12 is_identity pad]; src = Ainv [u, v, 1] in pixels,
// u = column (x, width axis), v = row (y, height axis).
struct Taps {
    int idx[4];    // token index y * W + x of the 4 bilinear taps (clamped when weight is 0)
    float w[4];    // bilinear weights, 0 for out-of-range taps (zeros padding)
    float roi;     // 1 when the nearest source pixel is inside the map
};

// The same sample in pixel coordinates: x0 / y0 = floor of the sampling position (clamped far outside the map so that the integer
// conversion stays defined), w[k] the weight of tap (x0 + (k & 1), y0 + (k >> 1)), already 0 where that pixel lies outside the
// map.  make_taps is built on it, so both forms hold bit-identical weights and visibility.
struct TapsXY {
    int x0, y0;
    float w[4];
    float roi;
};
__device__ __forceinline__ TapsXY make_taps_xy(const float* __restrict__ a, int u, int v, int H, int W) {
    TapsXY t;
    const float fu = (float)u, fv = (float)v;
    const float sx = fmaf(a[0], fu, fmaf(a[1], fv, a[2]));
    const float sy = fmaf(a[3], fu, fmaf(a[4], fv, a[5]));
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float wx1 = sx - x0f, wy1 = sy - y0f;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    // keep the integer conversion defined for far-away samples
    const float lim = 1.0e6f;
    const int x0 = (int)fminf(fmaxf(x0f, -lim), lim), y0 = (int)fminf(fmaxf(y0f, -lim), lim);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = (x0 >= 0) & (x0 < W), vx1 = (x1 >= 0) & (x1 < W);
    const bool vy0 = (y0 >= 0) & (y0 < H), vy1 = (y1 >= 0) & (y1 < H);
    t.x0 = x0; t.y0 = y0;
    t.w[0] = (vx0 & vy0) ? wx0 * wy0 : 0.f;
    t.w[1] = (vx1 & vy0) ? wx1 * wy0 : 0.f;
    t.w[2] = (vx0 & vy1) ? wx0 * wy1 : 0.f;
    t.w[3] = (vx1 & vy1) ? wx1 * wy1 : 0.f;
    // nearest (round-half-even, as std::nearbyint in grid_sample)
    const float nx = rintf(sx), ny = rintf(sy);
    t.roi = (nx >= 0.f && nx <= (float)(W - 1) && ny >= 0.f && ny <= (float)(H - 1)) ? 1.f : 0.f;
    return t;
}
__device__ __forceinline__ Taps make_taps(const float* __restrict__ a, int u, int v, int H, int W) {
    const TapsXY q = make_taps_xy(a, u, v, H, W);
    Taps t;
    const int cx0 = min(max(q.x0, 0), W - 1), cx1 = min(max(q.x0 + 1, 0), W - 1);
    const int cy0 = min(max(q.y0, 0), H - 1), cy1 = min(max(q.y0 + 1, 0), H - 1);
    t.idx[0] = cy0 * W + cx0; t.idx[1] = cy0 * W + cx1; t.idx[2] = cy1 * W + cx0; t.idx[3] = cy1 * W + cx1;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.w[k] = q.w[k];
    t.roi = q.roi;
    return t;
}

// Exchanges with lane ^ 32 / lane ^ 16 on the VALU (v_permlane32_swap / v_permlane16_swap, gfx950): with both operands
// holding x, the swap leaves the value of the lower partner in one register and of the upper partner in the other, in
// every lane - no LDS round trip as with ds_bpermute (__shfl_xor), whose wait sits on the critical path of every softmax
// row maximum and LayerNorm sum.  The elements of the builtin's result go through scalar copies: a bit_cast straight
// from a vector element is miscompiled by hipcc 7.2 (it reads element 0).
__device__ __forceinline__ void xor32_pair(float x, float& lo, float& hi) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    lo = __builtin_bit_cast(float, r0);
    hi = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void xor16_pair(float x, float& lo, float& hi) {
    const unsigned a = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    lo = __builtin_bit_cast(float, r0);
    hi = __builtin_bit_cast(float, r1);
}
// two-register forms (rows of 16 lanes: r0 .. r3): swap16: a.r1 <-> b.r0 and a.r3 <-> b.r2; swap32: a.r2, a.r3 <-> b.r0, b.r1
__device__ __forceinline__ void swap16_rows(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap32_rows(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
// max over the 64 lanes of a wavefront, in every lane, of non-negative floats taken as their bit patterns (they order like unsigned
// integers): four DPP steps inside the rows of 16 lanes, two row swaps across them - no LDS round trips (six ds_bpermute with
// __shfl_xor), no canonicalising v_max pairs
__device__ __forceinline__ unsigned wave_umax(unsigned u) {
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, 0xB1, 0xF, 0xF, false));      // quad_perm [1, 0, 3, 2]
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, 0x4E, 0xF, 0xF, false));      // quad_perm [2, 3, 0, 1]
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, 0x141, 0xF, 0xF, false));     // row_half_mirror
    u = max(u, (unsigned)__builtin_amdgcn_update_dpp((int)u, (int)u, 0x140, 0xF, 0xF, false));     // row_mirror
    {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        u = max(r0, r1);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        u = max(r0, r1);
    }
    return u;
}
__device__ __forceinline__ float wave_absmax(float m) { return __uint_as_float(wave_umax(__float_as_uint(m))); }   // m >= 0
__device__ __forceinline__ float max_raw(float a, float b) {   // v_max_f32 without the canonicalising max(x, x) pair
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float xor32_sum(float x) { float a, b; xor32_pair(x, a, b); return a + b; }
__device__ __forceinline__ float xor16_sum(float x) { float a, b; xor16_pair(x, a, b); return a + b; }
// maximum over the four lanes l, l ^ 16, l ^ 32, l ^ 48
__device__ __forceinline__ float max_over_lane_groups(float x) {
    float a, b;
    xor16_pair(x, a, b);
    xor32_pair(max_raw(a, b), a, b);
    return max_raw(a, b);
}

// Range normalisation of the split operands (HmvitStageScales, include/hmvit.h): power-of-two factors, so every scaling and
// its inverse are exact.  pow2_scale(y): the power of two s with y s in [2^13, 2^14) (y >= 0; exponent clamped to 2^+-40);
// pow2_inv(s) = 1 / s for a power of two.
__device__ __forceinline__ float pow2_scale(float y) {
    int e = (int)((__float_as_uint(y) >> 23) & 0xffu);
    e = min(max(e, 127 - 40), 127 + 40);
    return __uint_as_float((unsigned)(127 + 13 + 127 - e) << 23);
}
__device__ __forceinline__ float pow2_inv(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }

// pixel (row, col) of token `n` (row-major inside the w x w window) of window (wx, wy)
// for the two partitions (hetero_fusion.py:387-389 / :430-431)
__device__ __forceinline__ void token_pixel(int partition, int window, int X, int Y, int wx, int wy,
                                            int n, int& row, int& col) {
    const int w1 = n / window, w2 = n - w1 * window;
    if (partition == HMVIT_PART_WINDOW) {
        row = wx * window + w1;
        col = wy * window + w2;
    } else {
        row = w1 * X + wx;
        col = w2 * Y + wy;
    }
}

}  // namespace hmvit
