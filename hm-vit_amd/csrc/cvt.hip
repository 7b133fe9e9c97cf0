// Kernels of the camera -> BEV lift (CVT encoder): the parts of CrossViewAttention.forward
// (opencood/models/sub_modules/cvt_modules.py:216-280) and CrossAttention.forward (:118-173) that are not plain
// LayerNorm / Linear (those run on k_layernorm / the GEMM of gemm.hip).  Correctness-first f32 versions.
//
//   k_cvt_embed       camera-aware positional embeddings: image-ray embedding of every feature pixel (img_embed(E_inv
//                     [I_inv pixel; 1]) - cam_embed(camera centre), L2-normalised over channels, :236-250) and the BEV
//                     query embedding (bev_embed(world xy) - cam_embed(centre), normalised, + x, :252-269)
//   k_bn_relu_tokens  BatchNorm2d (eval) + ReLU of feature_linear / feature_proj (:197-207), NCHW -> token-major
//   k_cross_attention softmax over the keys of ALL cameras of one agent, queries differ per camera (:148-158)
#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per output token; lane owns channels lane, lane + 64, ...
__global__ __launch_bounds__(256) void k_cvt_embed(CvtEmbedParams p) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n_tok = p.bn * p.P;
    if (tok >= n_tok) return;
    const int bn = tok / p.P, pix = tok - bn * p.P;
    const float* E = p.E_inv + (size_t)bn * 16;
    float in[4];
    int nin;
    if (p.mode == 0) {
        // image plane: generate_grid(h, w) scaled by the image size (square maps: x along columns, y along rows)
        const int row = pix / p.W, col = pix - row * p.W;
        const float px = p.img_w * ((float)col / (float)(p.W - 1)), py = p.img_h * ((float)row / (float)(p.H - 1));
        const float* I = p.I_inv + (size_t)bn * 9;
        const float cam[4] = {I[0] * px + I[1] * py + I[2], I[3] * px + I[4] * py + I[5], I[6] * px + I[7] * py + I[8], 1.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) in[r] = E[4 * r] * cam[0] + E[4 * r + 1] * cam[1] + E[4 * r + 2] * cam[2] + E[4 * r + 3] * cam[3];
        nin = 4;
    } else {
        in[0] = p.grid[pix]; in[1] = p.grid[p.P + pix]; in[2] = in[3] = 0.f;
        nin = 2;
    }
    const float ctr[4] = {E[3], E[7], E[11], E[15]};
    float sq = 0.f;
    float e[8];
    for (int k = 0, c = lane; c < p.dim; c += 64, ++k) {
        float v = p.w_bias ? p.w_bias[c] : 0.f;
        for (int r = 0; r < nin; ++r) v = fmaf(p.w_in[c * nin + r], in[r], v);
        float ce = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) ce = fmaf(p.w_cam[c * 4 + r], ctr[r], ce);
        e[k] = v - ce;
        sq = fmaf(e[k], e[k], sq);
    }
    const float inv = 1.f / (sqrtf(wave_sum(sq)) + 1e-7f);
    for (int k = 0, c = lane; c < p.dim; c += 64, ++k) {
        float v = e[k] * inv;
        if (p.x) v += p.x[((size_t)(bn / p.n_cam) * p.dim + c) * p.P + pix];    // query = query_pos + x[:, None]
        p.out[(size_t)tok * p.dim + c] = v;
    }
}

int launch_cvt_embed(const CvtEmbedParams& p, hipStream_t st) {
    HMVIT_CHECK_ARG(p.dim > 0 && p.dim <= 512, "cvt_embed: dim=%d (1..512)", p.dim);
    const int n_tok = p.bn * p.P;
    if (n_tok <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_cvt_embed, dim3(cdiv(n_tok, 4)), dim3(256), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// x (n, C, P) -> y (n, P, C) = relu(x * scale[c] + shift[c]); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_bn_relu_tokens(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y, int C, int P) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pp = p0 + tx;
        float v = 0.f;
        if (c < C && pp < P) v = fmaxf(fmaf(x[((size_t)n * C + c) * P + pp], scale[c], shift[c]), 0.f);
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int pp = p0 + r, c = c0 + tx;
        if (pp < P && c < C) y[((size_t)n * P + pp) * C + c] = tile[tx][r];
    }
}

int launch_bn_relu_tokens(const float* x, const float* scale, const float* shift, float* y, int n, int C, int P, hipStream_t st) {
    if (n <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_bn_relu_tokens, dim3(cdiv(P, 32), cdiv(C, 32), n), dim3(256), 0, st, x, scale, shift, y, C, P);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// q (b, n, Q, heads * 32), k (b, n, K, heads * 32), v (b, n * K, heads * 32) -> out (b, Q, heads * 32), exact f32.
// One workgroup = 64 queries of one head of one agent, one wavefront per 16 queries; K / V tiles of 64 keys staged in LDS and
// shared by the four waves; online softmax over the keys of ALL cameras (the queries differ per camera).
// v_mfma_f32_16x16x4_f32 (lane (l, g) supplies A[row l][k g], B[k g][col l], receives D[row 4 g + r][col l]):
//   logits transposed, S^T[key][query] = K Q^T: a lane owns ONE query (column l) and keys 4 g + r of each 16-key sub-tile, so
//   the row maximum / sum are lane-local plus two cross-lane steps (lanes l, l ^ 16, l ^ 32);
//   O^T[d][query] = V^T P^T: the accumulator of S^T is directly the B operand, step r of a sub-tile contracting over the keys
//   {4 g + r}; the A operand V^T is read from LDS in that order.
// (Round 1 ran one thread per query with scalar FMAs: 7.3 ms at the shipped level-1 size - 5 agents x 4 cameras x 4096 keys,
// 1024 queries - half of the camera branch's fp32-parity time.)
// bias: optional (heads, Q, K) additive logit bias (n_cam = 1): the relative-position bias of FAX's self-attention
// (fax_modules.py:122-160).
typedef float float4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float xlane_max4(float v) {      // over the four lanes that share l = lane & 15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xlane_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__global__ __launch_bounds__(256) void k_cross_attention(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ out, int n_cam, int Q,
                                                         int K, int heads, float scale, const float* __restrict__ bias,
                                                         float* __restrict__ lse) {
    constexpr int D = 32, KS = D + 1;
    __shared__ float Ks[64 * KS], Vs[64 * KS];
    const int b = blockIdx.z, head = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l = lane & 15, g = lane >> 4;
    const int HD = heads * D;
    const int qi = blockIdx.x * 64 + wave * 16 + l;
    const bool valid = qi < Q;
    const int qc = valid ? qi : Q - 1;
    float4m o_acc[2] = {(float4m)(0.f), (float4m)(0.f)};
    float m_run = -INFINITY, l_part = 0.f;       // l_part: this lane's share of the denominator (keys 4 g + r)
    const int skey = tid >> 2, sd0 = (tid & 3) * 8;          // staging: key, first channel
    for (int cam = 0; cam < n_cam; ++cam) {
        float qf[8];
        const float* qp = q + (((size_t)(b * n_cam + cam) * Q + qc) * HD + head * D);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = qp[4 * ks + g] * scale;
        for (int k0 = 0; k0 < K; k0 += 64) {
            __syncthreads();
            {
                const int kk = k0 + skey;
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, b0 = a0, b1 = a0;
                if (kk < K) {
                    const float* kp = k + (((size_t)(b * n_cam + cam) * K + kk) * HD + head * D + sd0);
                    const float* vp = v + (((size_t)b * n_cam * K + (size_t)cam * K + kk) * HD + head * D + sd0);
                    a0 = *reinterpret_cast<const float4*>(kp); a1 = *reinterpret_cast<const float4*>(kp + 4);
                    b0 = *reinterpret_cast<const float4*>(vp); b1 = *reinterpret_cast<const float4*>(vp + 4);
                }
                float* kd = Ks + skey * KS + sd0;
                float* vd = Vs + skey * KS + sd0;
                kd[0] = a0.x; kd[1] = a0.y; kd[2] = a0.z; kd[3] = a0.w; kd[4] = a1.x; kd[5] = a1.y; kd[6] = a1.z; kd[7] = a1.w;
                vd[0] = b0.x; vd[1] = b0.y; vd[2] = b0.z; vd[3] = b0.w; vd[4] = b1.x; vd[5] = b1.y; vd[6] = b1.z; vd[7] = b1.w;
            }
            __syncthreads();
            float4m s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = (float4m)(0.f);
                if (bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + 4 * g + r;
                        s[kt][r] = key < K ? bias[((size_t)head * Q + qc) * K + key] : 0.f;
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(kt * 16 + l) * KS + 4 * ks + g], qf[ks], s[kt], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (k0 + kt * 16 + 4 * g + r >= K) s[kt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            const float m_new = fmaxf(m_run, xlane_max4(mx));
            const float alpha = __expf(m_run - m_new);          // m_new is finite: a tile holds at least one real key
            l_part *= alpha;
            o_acc[0] *= alpha; o_acc[1] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __expf(s[kt][r] - m_new);
                    l_part += pj;
                    // contraction step over the keys {4 g' + r}: A = V^T[d = dt * 16 + l][key 4 g + r], B = this lane's p
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(kt * 16 + 4 * g + r) * KS + dt * 16 + l], pj, o_acc[dt], 0, 0, 0);
                }
            }
            m_run = m_new;
        }
    }
    const float l_sum = xlane_sum4(l_part);
    const float inv = 1.f / l_sum;
    // training: the row's log-sum-exp (logits in scaled units, q . k / sqrt(d)) lets the backward pass rebuild the probabilities
    if (lse && valid && g == 0) lse[((size_t)b * heads + head) * Q + qi] = m_run + logf(l_sum);
    if (valid) {
        // o_acc[dt][r] = O^T[d = dt * 16 + 4 g + r][query l]
        float* op = out + (((size_t)b * Q + qi) * HD + head * D + 4 * g);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            *reinterpret_cast<float4*>(op + dt * 16) = make_float4(o_acc[dt][0] * inv, o_acc[dt][1] * inv, o_acc[dt][2] * inv, o_acc[dt][3] * inv);
    }
}

int launch_cross_attention(const float* q, const float* k, const float* v, float* out, int b, int n_cam, int Q, int K, int heads,
                           int dim_head, const float* bias, hipStream_t st, float* lse) {
    HMVIT_CHECK_ARG(dim_head == 32, "cross_attention: dim_head=%d (32)", dim_head);
    HMVIT_CHECK_ARG(!bias || n_cam == 1, "cross_attention: a logit bias needs n_cam = 1 (got %d)", n_cam);
    if (b <= 0 || Q <= 0) return HMVIT_OK;
    hipLaunchKernelGGL(k_cross_attention, dim3(cdiv(Q, 64), heads, b), dim3(256), 0, st, q, k, v, out, n_cam, Q, K, heads,
                       1.f / sqrtf((float)dim_head), bias, lse);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// Backward pass of the joint-softmax cross attention (training of the camera lift, cvt_modules.py:95-165 under autograd).
//   P[q, (cam, k)] = exp(s Q_cam[q] . K_cam[k] - lse[q]),   D[q] = dO[q] . O[q]
//   dV[cam, k] = sum_q P dO[q],   dS = P (dO[q] . V[cam, k] - D[q]) s,   dQ_cam[q] = sum_k dS K_cam[k],   dK_cam[k] = sum_q dS Q_cam[q]
// Two launches, neither with atomics: k_cross_attention_bwd_dq owns 64 queries of one head and walks every key tile,
// k_cross_attention_bwd_dkv owns 64 keys of one camera and walks every query tile; both rebuild the 64 x 64 tile of P / dS from
// the saved log-sum-exp (thread (tq, tk) = 4 x 4 query-key pairs, f32 FMAs over the 32 channels out of LDS) and contract it
// through LDS.  Correctness first: plain f32 FMAs, about a third of the f32 vector peak.
// ------------------------------------------------------------------------------------------
namespace {
constexpr int XB_D = 32, XB_T = 64, XB_LS = XB_D + 1;

// tile rows [r0, r0 + 64) of a (rows, heads * 32) matrix -> LDS (64 x 33), zero beyond n_rows
__device__ __forceinline__ void xb_load_tile(float* dst, const float* __restrict__ src, int r0, int n_rows, int HD, int head, float mul) {
    const int row = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 8;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (r0 + row < n_rows) {
        const float* p = src + (size_t)(r0 + row) * HD + head * XB_D + c0;
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    float* d = dst + row * XB_LS + c0;
    d[0] = a.x * mul; d[1] = a.y * mul; d[2] = a.z * mul; d[3] = a.w * mul; d[4] = b.x * mul; d[5] = b.y * mul; d[6] = b.z * mul; d[7] = b.w * mul;
}

// P and dS of the 4 x 4 pairs (queries 4 tq + i, keys 4 tk + j) of a 64 x 64 tile
// bias_tile: optional, &bias[(head Q + q0) K + k0] with row stride K (the additive logit bias of hmvit_attention_bias)
__device__ __forceinline__ void xb_tile_probs(const float* Qs, const float* dOs, const float* Ks, const float* Vs, const float* lse_s,
                                              const float* d_s, int tq, int tk, int q_valid, int k_valid, float scale,
                                              float (&P)[4][4], float (&dS)[4][4], const float* __restrict__ bias_tile = nullptr,
                                              int bias_ld = 0) {
    float s[4][4], dp[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[i][j] = dp[i][j] = 0.f;
#pragma unroll 8
    for (int c = 0; c < XB_D; ++c) {
        float qv[4], ov[4], kv[4], vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { qv[i] = Qs[(4 * tq + i) * XB_LS + c]; ov[i] = dOs[(4 * tq + i) * XB_LS + c]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { kv[j] = Ks[(4 * tk + j) * XB_LS + c]; vv[j] = Vs[(4 * tk + j) * XB_LS + c]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[i][j] = fmaf(qv[i], kv[j], s[i][j]);
                dp[i][j] = fmaf(ov[i], vv[j], dp[i][j]);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (4 * tq + i < q_valid) && (4 * tk + j < k_valid);
            const float bv = (bias_tile && ok) ? bias_tile[(size_t)(4 * tq + i) * bias_ld + 4 * tk + j] : 0.f;
            const float p = ok ? __expf(s[i][j] * scale + bv - lse_s[4 * tq + i]) : 0.f;
            P[i][j] = p;
            dS[i][j] = p * (dp[i][j] - d_s[4 * tq + i]) * scale;
        }
}
}  // namespace

// grid (Q / 64, heads, b): dq (b, n_cam, Q, HD)
__global__ __launch_bounds__(256) void k_cross_attention_bwd_dq(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ out,
                                                                const float* __restrict__ lse, const float* __restrict__ d_out,
                                                                float* __restrict__ dq, int n_cam, int Q, int K, int heads, float scale,
                                                                const float* __restrict__ bias) {
    __shared__ float Qs[XB_T * XB_LS], dOs[XB_T * XB_LS], Ks[XB_T * XB_LS], Vs[XB_T * XB_LS], Ss[XB_T * (XB_T + 1)];
    __shared__ float lse_s[XB_T], d_s[XB_T];
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * XB_T, HD = heads * XB_D;
    const int tid = threadIdx.x, tq = tid >> 4, tk = tid & 15;
    const int q_valid = min(XB_T, Q - q0);
    xb_load_tile(dOs, d_out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);
    xb_load_tile(Ss, out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);          // O, only for D
    __syncthreads();
    if (tid < XB_T) {
        float d = 0.f;
        for (int c = 0; c < XB_D; ++c) d = fmaf(dOs[tid * XB_LS + c], Ss[tid * XB_LS + c], d);
        d_s[tid] = d;
        lse_s[tid] = tid < q_valid ? lse[((size_t)b * heads + head) * Q + q0 + tid] : 0.f;
    }
    const int orow = tid >> 2, oc0 = (tid & 3) * 8;       // output role: query row, 8 channels
    for (int cam = 0; cam < n_cam; ++cam) {
        __syncthreads();
        xb_load_tile(Qs, q + (size_t)(b * n_cam + cam) * Q * HD, q0, Q, HD, head, 1.f);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int k0 = 0; k0 < K; k0 += XB_T) {
            __syncthreads();
            xb_load_tile(Ks, k + (size_t)(b * n_cam + cam) * K * HD, k0, K, HD, head, 1.f);
            xb_load_tile(Vs, v + ((size_t)b * n_cam + cam) * K * HD, k0, K, HD, head, 1.f);
            __syncthreads();
            float P[4][4], dS[4][4];
            xb_tile_probs(Qs, dOs, Ks, Vs, lse_s, d_s, tq, tk, q_valid, min(XB_T, K - k0), scale, P, dS,
                          bias ? bias + ((size_t)head * Q + q0) * K + k0 : nullptr, K);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Ss[(4 * tq + i) * (XB_T + 1) + 4 * tk + j] = dS[i][j];
            __syncthreads();
            for (int kk = 0; kk < XB_T; ++kk) {
                const float ds = Ss[orow * (XB_T + 1) + kk];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(ds, Ks[kk * XB_LS + oc0 + e], acc[e]);
            }
        }
        if (orow < q_valid) {
            float* o = dq + ((size_t)(b * n_cam + cam) * Q + q0 + orow) * HD + head * XB_D + oc0;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

// grid (K / 64, heads * n_cam, b): dk (b, n_cam, K, HD), dv (b, n_cam * K, HD)
__global__ __launch_bounds__(256) void k_cross_attention_bwd_dkv(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ d_out,
                                                                 float* __restrict__ dk, float* __restrict__ dv, int n_cam, int Q, int K,
                                                                 int heads, float scale, const float* __restrict__ bias) {
    __shared__ float Qs[XB_T * XB_LS], dOs[XB_T * XB_LS], Ks[XB_T * XB_LS], Vs[XB_T * XB_LS], Ss[XB_T * (XB_T + 1)], Ps[XB_T * (XB_T + 1)];
    __shared__ float lse_s[XB_T], d_s[XB_T];
    const int b = blockIdx.z, head = blockIdx.y % heads, cam = blockIdx.y / heads, k0 = blockIdx.x * XB_T, HD = heads * XB_D;
    const int tid = threadIdx.x, tq = tid >> 4, tk = tid & 15;
    const int k_valid = min(XB_T, K - k0);
    xb_load_tile(Ks, k + (size_t)(b * n_cam + cam) * K * HD, k0, K, HD, head, 1.f);
    xb_load_tile(Vs, v + ((size_t)b * n_cam + cam) * K * HD, k0, K, HD, head, 1.f);
    const int orow = tid >> 2, oc0 = (tid & 3) * 8;       // output role: key row, 8 channels
    float acck[8], accv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acck[e] = accv[e] = 0.f;
    for (int q0 = 0; q0 < Q; q0 += XB_T) {
        __syncthreads();
        const int q_valid = min(XB_T, Q - q0);
        xb_load_tile(Qs, q + (size_t)(b * n_cam + cam) * Q * HD, q0, Q, HD, head, 1.f);
        xb_load_tile(dOs, d_out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);
        xb_load_tile(Ss, out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);      // O (row stride XB_LS inside Ss), only for D
        __syncthreads();
        if (tid < XB_T) {
            float d = 0.f;
            for (int c = 0; c < XB_D; ++c) d = fmaf(dOs[tid * XB_LS + c], Ss[tid * XB_LS + c], d);
            d_s[tid] = d;
            lse_s[tid] = tid < q_valid ? lse[((size_t)b * heads + head) * Q + q0 + tid] : 0.f;
        }
        __syncthreads();
        float P[4][4], dS[4][4];
        xb_tile_probs(Qs, dOs, Ks, Vs, lse_s, d_s, tq, tk, q_valid, k_valid, scale, P, dS,
                      bias ? bias + ((size_t)head * Q + q0) * K + k0 : nullptr, K);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Ss[(4 * tq + i) * (XB_T + 1) + 4 * tk + j] = dS[i][j];
                Ps[(4 * tq + i) * (XB_T + 1) + 4 * tk + j] = P[i][j];
            }
        __syncthreads();
        for (int qq = 0; qq < XB_T; ++qq) {
            const float ds = Ss[qq * (XB_T + 1) + orow], pp = Ps[qq * (XB_T + 1) + orow];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acck[e] = fmaf(ds, Qs[qq * XB_LS + oc0 + e], acck[e]);
                accv[e] = fmaf(pp, dOs[qq * XB_LS + oc0 + e], accv[e]);
            }
        }
    }
    if (orow < k_valid) {
        const size_t row = ((size_t)(b * n_cam + cam) * K + k0 + orow) * HD + head * XB_D + oc0;
        *reinterpret_cast<float4*>(dk + row) = make_float4(acck[0], acck[1], acck[2], acck[3]);
        *reinterpret_cast<float4*>(dk + row + 4) = make_float4(acck[4], acck[5], acck[6], acck[7]);
        *reinterpret_cast<float4*>(dv + row) = make_float4(accv[0], accv[1], accv[2], accv[3]);
        *reinterpret_cast<float4*>(dv + row + 4) = make_float4(accv[4], accv[5], accv[6], accv[7]);
    }
}

// d_bias[head][q][k] = sum over the batch of dS / scale: one 64 x 64 tile per workgroup, the batch walked inside (no atomics)
__global__ __launch_bounds__(256) void k_attention_dbias(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         const float* __restrict__ out, const float* __restrict__ lse,
                                                         const float* __restrict__ d_out, const float* __restrict__ bias,
                                                         float* __restrict__ d_bias, int batch, int Q, int K, int heads, float scale) {
    __shared__ float Qs[XB_T * XB_LS], dOs[XB_T * XB_LS], Ks[XB_T * XB_LS], Vs[XB_T * XB_LS], Os[XB_T * XB_LS];
    __shared__ float lse_s[XB_T], d_s[XB_T];
    const int head = blockIdx.y, q0 = blockIdx.x * XB_T, k0 = blockIdx.z * XB_T, HD = heads * XB_D;
    const int tid = threadIdx.x, tq = tid >> 4, tk = tid & 15;
    const int q_valid = min(XB_T, Q - q0), k_valid = min(XB_T, K - k0);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const float inv_scale = 1.f / scale;
    for (int b = 0; b < batch; ++b) {
        __syncthreads();
        xb_load_tile(Qs, q + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);
        xb_load_tile(dOs, d_out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);
        xb_load_tile(Os, out + (size_t)b * Q * HD, q0, Q, HD, head, 1.f);
        xb_load_tile(Ks, k + (size_t)b * K * HD, k0, K, HD, head, 1.f);
        xb_load_tile(Vs, v + (size_t)b * K * HD, k0, K, HD, head, 1.f);
        __syncthreads();
        if (tid < XB_T) {
            float d = 0.f;
            for (int c = 0; c < XB_D; ++c) d = fmaf(dOs[tid * XB_LS + c], Os[tid * XB_LS + c], d);
            d_s[tid] = d;
            lse_s[tid] = tid < q_valid ? lse[((size_t)b * heads + head) * Q + q0 + tid] : 0.f;
        }
        __syncthreads();
        float P[4][4], dS[4][4];
        xb_tile_probs(Qs, dOs, Ks, Vs, lse_s, d_s, tq, tk, q_valid, k_valid, scale, P, dS, bias + ((size_t)head * Q + q0) * K + k0, K);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += dS[i][j] * inv_scale;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * tq + i < q_valid && 4 * tk + j < k_valid) d_bias[((size_t)head * Q + q0 + 4 * tq + i) * K + k0 + 4 * tk + j] = acc[i][j];
}

int launch_cross_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* d_out,
                               float* dq, float* dk, float* dv, int b, int n_cam, int Q, int K, int heads, int dim_head, hipStream_t st,
                               const float* bias, float* d_bias) {
    HMVIT_CHECK_ARG(dim_head == 32, "cross_attention_bwd: dim_head=%d (32)", dim_head);
    HMVIT_CHECK_ARG(!bias || n_cam == 1, "cross_attention_bwd: a logit bias needs n_cam = 1 (got %d)", n_cam);
    if (b <= 0 || Q <= 0 || K <= 0) return HMVIT_OK;
    const float scale = 1.f / sqrtf((float)dim_head);
    hipLaunchKernelGGL(k_cross_attention_bwd_dq, dim3(cdiv(Q, 64), heads, b), dim3(256), 0, st, q, k, v, out, lse, d_out, dq, n_cam, Q, K,
                       heads, scale, bias);
    hipLaunchKernelGGL(k_cross_attention_bwd_dkv, dim3(cdiv(K, 64), heads * n_cam, b), dim3(256), 0, st, q, k, v, out, lse, d_out, dk, dv,
                       n_cam, Q, K, heads, scale, bias);
    if (bias && d_bias)
        hipLaunchKernelGGL(k_attention_dbias, dim3(cdiv(Q, 64), heads, cdiv(K, 64)), dim3(256), 0, st, q, k, v, out, lse, d_out, bias, d_bias,
                           b, Q, K, heads, scale);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit

namespace hmvit {

// ------------------------------------------------------------------------------------------
// MFMA version of the cross attention (f16 operands, f32 accumulate / softmax): one workgroup = 64 queries of one head
// (4 waves x 16 queries), K / V tiles of 64 keys staged in LDS and shared by the waves, the next tile's rows prefetched
// into registers while the current one is multiplied.  Same fragment scheme as the fusion's attention (attn.hip):
// transposed logits S^T = K Q^T (v_mfma_f32_16x16x32_f16) so a lane owns one query column, exp2 with log2(e) / sqrt(d)
// folded into the query operand, P^T straight into O^T = V^T P^T with V^T fragments from ds_read_b64_tr_b16, the softmax
// denominator from an all-ones tile.  Q and K must be multiples of 64 (the f32 kernel above handles the rest).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cross_attention_mfma(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                              const half_t* __restrict__ v, float* __restrict__ out, int n_cam,
                                                              int Q, int K, int heads, float qscale) {
    constexpr int D = 32, KS = D + 8, VS = D + 16;
    __shared__ __attribute__((aligned(16))) half_t Ks[2][64 * KS];
    __shared__ __attribute__((aligned(16))) half_t Vs[2][64 * VS];
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
    const int HD = heads * D;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lq = lane & 15, g = lane >> 4;
    // staging role of this thread: key row and 8-channel piece of a 64 x 32 tile
    const int srow = threadIdx.x >> 2, spiece = (threadIdx.x & 3) * 8;

    float m_run = -INFINITY;
    float4v o_acc[2], l_acc;
    o_acc[0] = o_acc[1] = l_acc = (float4v)(0.f);
    const half8 ones = (half8)(half_t)1.0f;
    const int n_tiles = K / 64, total = n_cam * n_tiles;

    auto load_tile = [&](int it, half8& rk, half8& rv) {
        const int cam = it / n_tiles, kt = it - cam * n_tiles;
        const size_t key = (size_t)(b * n_cam + cam) * K + (size_t)kt * 64 + srow;
        rk = *reinterpret_cast<const half8*>(k + key * HD + head * D + spiece);
        rv = *reinterpret_cast<const half8*>(v + key * HD + head * D + spiece);      // v is (b, n K, HD): same row index
    };
    auto store_tile = [&](int buf, const half8& rk, const half8& rv) {
        *reinterpret_cast<half8*>(Ks[buf] + srow * KS + spiece) = rk;
        *reinterpret_cast<half8*>(Vs[buf] + srow * VS + spiece) = rv;
    };

    half8 rk, rv, qh;
    load_tile(0, rk, rv);
    store_tile(0, rk, rv);
    __syncthreads();
    int cam_loaded = -1;
    for (int it = 0; it < total; ++it) {
        const int buf = it & 1, cam = it / n_tiles;
        if (it + 1 < total) load_tile(it + 1, rk, rv);
        if (cam != cam_loaded) {      // the query operand of this camera, pre-scaled by log2(e) / sqrt(d)
            const half8 raw = *reinterpret_cast<const half8*>(q + ((size_t)(b * n_cam + cam) * Q + q0 + wave * 16 + lq) * HD + head * D + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[e] = (half_t)((float)raw[e] * qscale);
            cam_loaded = cam;
        }
        const half_t* Kb = Ks[buf];
        const half_t* Vb = Vs[buf];
        half8 kh[4], vh[2][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kh[kt] = *reinterpret_cast<const half8*>(Kb + (kt * 16 + lq) * KS + g * 8);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half_t* base = Vb + (ks * 32 + 4 * g + (lq >> 2)) * VS + (lq & 3) * 8 + dt * 4;
                const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                half8 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) { t[e] = (half_t)lo[e]; t[4 + e] = (half_t)hi[e]; }
                vh[dt][ks] = t;
            }
        float4v s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[kt], qh, (float4v)(0.f), 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = max_over_lane_groups(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m_new);
        m_run = m_new;
        o_acc[0] *= alpha; o_acc[1] *= alpha; l_acc *= alpha;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ph;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ph[e] = (half_t)s[2 * ks][e]; ph[4 + e] = (half_t)s[2 * ks + 1][e]; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[dt][ks], ph, o_acc[dt], 0, 0, 0);
            l_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, l_acc, 0, 0, 0);
        }
        if (it + 1 < total) store_tile(buf ^ 1, rk, rv);    // that buffer was last read before the previous barrier
        __syncthreads();
    }
    // lane (query lq, g) holds channels 8 g + 4 dt + r (row order of the V^T tiles)
    const float inv = 1.f / l_acc[0];
    float* op = out + ((size_t)b * Q + q0 + wave * 16 + lq) * HD + head * D + 8 * g;
    *reinterpret_cast<float4*>(op) = make_float4(o_acc[0][0] * inv, o_acc[0][1] * inv, o_acc[0][2] * inv, o_acc[0][3] * inv);
    *reinterpret_cast<float4*>(op + 4) = make_float4(o_acc[1][0] * inv, o_acc[1][1] * inv, o_acc[1][2] * inv, o_acc[1][3] * inv);
}

// ------------------------------------------------------------------------------------------
// The same joint-softmax cross attention on f32 operands with split-f16 products (HMVIT_PREC_SPLIT): every product x y as
// x_hi y_hi + x_lo y_hi + x_hi y_lo on v_mfma_f32_16x16x32_f16, f32 accumulate - fp32-class results at a fifth of the matrix cycles
// of the exact-f32 kernel above (v_mfma_f32_16x16x4_f32), which the split model used until round 4: 0.87 ms of the 6.3 ms CVT
// encoder for its largest launch (5 agents x 4 cameras x 4096 keys, 1024 queries: 43 GF at 49 TF/s).
// Workgroup = 64 queries of one head of one agent, EIGHT wavefronts: wavefront (qg, kh) takes queries 16 qg .. 16 qg + 15 and the
// 32 keys kh of every 64-key tile (the launch has only Q / 64 x heads x agents = 320 workgroups: two wavefronts per SIMD and
// half the matrix work per wavefront instead of one workgroup of four on most CUs); the two halves keep their own running
// maximum / denominator / output and are merged once, through LDS, at the end.  K / V rows travel global (f32) -> registers ->
// (hi, lo) halves -> LDS, one tile ahead, one barrier per tile; fragments and the V^T transposed reads as in
// k_cross_attention_mfma.  Q and K multiples of 64, no logit bias, no saved log-sum-exp (those calls keep the f32 kernel).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_cross_attention_split(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, float* __restrict__ out, int n_cam,
                                                               int Q, int K, int heads, float qscale) {
    constexpr int D = 32, KS = D + 8, VS = D + 16;
    __shared__ __attribute__((aligned(16))) half_t Kh[2][64 * KS], Kl[2][64 * KS];
    __shared__ __attribute__((aligned(16))) half_t Vh[2][64 * VS], Vl[2][64 * VS];
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 64;
    const int HD = heads * D;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, lq = lane & 15, g = lane >> 4;
    const int qg = wave & 3, kh = wave >> 2;
    // staging role of this thread: key row and 4-channel piece of a 64 x 32 tile
    const int srow = threadIdx.x >> 3, spiece = (threadIdx.x & 7) * 4;

    float m_run = -INFINITY;
    float4v o_acc[2], l_acc;
    o_acc[0] = o_acc[1] = l_acc = (float4v)(0.f);
    const half8 ones = (half8)(half_t)1.0f;
    const int n_tiles = K / 64, total = n_cam * n_tiles;

    auto load_tile = [&](int it, float4& rk, float4& rv) {
        const int cam = it / n_tiles, kt = it - cam * n_tiles;
        const size_t key = (size_t)(b * n_cam + cam) * K + (size_t)kt * 64 + srow;
        rk = *reinterpret_cast<const float4*>(k + key * HD + head * D + spiece);
        rv = *reinterpret_cast<const float4*>(v + key * HD + head * D + spiece);      // v is (b, n K, HD): same row index
    };
    auto split4 = [&](const float4& x, half_t* hi, half_t* lo) {
        const float f[4] = {x.x, x.y, x.z, x.w};
        half4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (half_t)f[e];
            l[e] = (half_t)(f[e] - (float)h[e]);
        }
        *reinterpret_cast<half4*>(hi) = h;
        *reinterpret_cast<half4*>(lo) = l;
    };
    auto store_tile = [&](int buf, const float4& rk, const float4& rv) {
        split4(rk, Kh[buf] + srow * KS + spiece, Kl[buf] + srow * KS + spiece);
        split4(rv, Vh[buf] + srow * VS + spiece, Vl[buf] + srow * VS + spiece);
    };

    // rows one tile ahead in registers.  (Tried on top of this, measured on the 4 x 4096-key launch: four tiles ahead with the
    // query operands in LDS and hipcc's waits made countable - vmcnt(5 / 4) instead of vmcnt(0) in front of the row stores -
    // 350 -> 460 us: the tile time is not the global round trip.)
    float4 rk, rv;
    half8 qh, ql;
    load_tile(0, rk, rv);
    store_tile(0, rk, rv);
    __syncthreads();
    int cam_loaded = -1;
    for (int it = 0; it < total; ++it) {
        {
            const int buf = it & 1, cam = it / n_tiles;
            if (it + 1 < total) load_tile(it + 1, rk, rv);
            if (cam != cam_loaded) {      // the query operand of this camera, pre-scaled by log2(e) / sqrt(d), as (hi, lo)
                const float* qp = q + ((size_t)(b * n_cam + cam) * Q + q0 + qg * 16 + lq) * HD + head * D + g * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(qp), a1 = *reinterpret_cast<const float4*>(qp + 4);
                const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = f[e] * qscale;
                    qh[e] = (half_t)x;
                    ql[e] = (half_t)(x - (float)qh[e]);
                }
                cam_loaded = cam;
            }
            // this wavefront's 32 keys of the tile: key sub-tiles 2 kh, 2 kh + 1
            half8 k_h[2], k_l[2], v_h[2], v_l[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                k_h[kt] = *reinterpret_cast<const half8*>(Kh[buf] + ((2 * kh + kt) * 16 + lq) * KS + g * 8);
                k_l[kt] = *reinterpret_cast<const half8*>(Kl[buf] + ((2 * kh + kt) * 16 + lq) * KS + g * 8);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const half_t* base = (hl ? Vl[buf] : Vh[buf]) + (kh * 32 + 4 * g + (lq >> 2)) * VS + (lq & 3) * 8 + dt * 4;
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                    half8 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { t[e] = (half_t)lo[e]; t[4 + e] = (half_t)hi[e]; }
                    if (hl) v_l[dt] = t; else v_h[dt] = t;
                }
            float4v s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                float4v acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_l[kt], qh, (float4v)(0.f), 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_h[kt], ql, acc, 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_h[kt], qh, acc, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            mx = max_over_lane_groups(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            half8 ph, pl;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[kt][r] - m_new);
                    const half_t eh = (half_t)e;
                    ph[4 * kt + r] = eh;
                    pl[4 * kt + r] = (half_t)(e - (float)eh);
                }
            m_run = m_new;
            o_acc[0] *= alpha; o_acc[1] *= alpha; l_acc *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_l[dt], ph, o_acc[dt], 0, 0, 0);
                o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_h[dt], pl, o_acc[dt], 0, 0, 0);
                o_acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_h[dt], ph, o_acc[dt], 0, 0, 0);
            }
            l_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pl, l_acc, 0, 0, 0);
            l_acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ph, l_acc, 0, 0, 0);
            if (it + 1 < total) store_tile(buf ^ 1, rk, rv);    // that buffer was last read before the previous barrier
            __syncthreads();
        }
    }
    // merge the two key halves of a query group (same lane, wavefronts qg and qg + 4): LDS is free after the last barrier
    __shared__ float merge[4][10][64];
    if (kh == 1) {
        merge[qg][0][lane] = m_run;
        merge[qg][1][lane] = l_acc[0];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) merge[qg][2 + 4 * dt + r][lane] = o_acc[dt][r];
    }
    __syncthreads();
    if (kh == 0) {
        const float m2 = merge[qg][0][lane], l2 = merge[qg][1][lane];
        const float m = fmaxf(m_run, m2);
        const float a1 = __builtin_amdgcn_exp2f(m_run - m), a2 = __builtin_amdgcn_exp2f(m2 - m);
        const float inv = 1.f / (l_acc[0] * a1 + l2 * a2);
        float o[8];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[4 * dt + r] = (o_acc[dt][r] * a1 + merge[qg][2 + 4 * dt + r][lane] * a2) * inv;
        // lane (query lq, g) holds channels 8 g + 4 dt + r (row order of the V^T tiles)
        float* op = out + ((size_t)b * Q + q0 + qg * 16 + lq) * HD + head * D + 8 * g;
        *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

int launch_cross_attention_split(const float* q, const float* k, const float* v, float* out, int b, int n_cam, int Q, int K,
                                 int heads, int dim_head, hipStream_t st) {
    HMVIT_CHECK_ARG(dim_head == 32 && Q % 64 == 0 && K % 64 == 0, "cross_attention (split): dim_head=%d (32), Q=%d, K=%d (multiples of 64)",
                    dim_head, Q, K);
    if (b <= 0 || Q <= 0) return HMVIT_OK;
    const float qscale = 1.44269504088896341f / sqrtf((float)dim_head);
    hipLaunchKernelGGL(k_cross_attention_split, dim3(Q / 64, heads, b), dim3(512), 0, st, q, k, v, out, n_cam, Q, K, heads, qscale);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

int launch_cross_attention_f16(const half_t* q, const half_t* k, const half_t* v, float* out, int b, int n_cam, int Q, int K,
                               int heads, int dim_head, hipStream_t st) {
    HMVIT_CHECK_ARG(dim_head == 32 && Q % 64 == 0 && K % 64 == 0, "cross_attention (f16): dim_head=%d (32), Q=%d, K=%d (multiples of 64)",
                    dim_head, Q, K);
    if (b <= 0 || Q <= 0) return HMVIT_OK;
    const float qscale = 1.44269504088896341f / sqrtf((float)dim_head);
    hipLaunchKernelGGL(k_cross_attention_mfma, dim3(Q / 64, heads, b), dim3(256), 0, st, q, k, v, out, n_cam, Q, K, heads, qscale);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

}  // namespace hmvit
