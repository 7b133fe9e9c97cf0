// Internal launch interfaces between the translation units of libhmvit.
#pragma once
#include "common.hpp"

namespace hmvit {

constexpr int kMaxSlots = 64;  // agent slots (B * L) handled by one launch

struct AgentTypes {
    int8_t t[kMaxSlots];
};

// ---- capi.hip ----
int check_desc(const HmvitFusionDesc* d);   // argument validation shared by the inference and training entry points

// ---- tok.hip ----
int launch_transpose(const float* x, float* y, int n, int R, int S, hipStream_t st);
int launch_layernorm(const float* x, void* y, const float* gamma, const float* beta,
                     const AgentTypes& types, int n_agents, int P, int C, int precision,
                     hipStream_t st);
int launch_pair_affines(const float* t, float* ainv, int n, int H, int W, float discrete_ratio,
                        float downsample_rate, hipStream_t st);
int launch_warp(const float* src, const float* ainv, float* dst, float* roi, int n, int H, int W,
                int C, hipStream_t st);

// ---- gemm.hip ----
// y[plane(n)][m][n % n_per_plane] = act(sum_k a[m][k] w[n][k] + bias[n]) + residual[m][n]
struct GemmJob {
    const void* a;          // (M, K) element type T, or f32 when a_f32
    const void* w;          // (N, K) element type T
    const float* bias;      // (N) or null
    const float* residual;  // (M, N) f32 or null (may alias y when out_f32)
    void* y;
    long long plane_stride;  // elements between output planes
    int M, N, K;
    int n_per_plane;        // columns per output plane (N when there is a single plane)
};
constexpr int kMaxJobs = 24;
struct GemmJobs {
    GemmJob j[kMaxJobs];
    int n;
};
// all jobs of one launch share a_f32 / gelu / out_f32
int launch_gemm(const GemmJobs& jobs, bool a_f32, bool gelu, bool out_f32, int precision,
                hipStream_t st);

// ---- chain.hip: Linear(256 -> 256 n_mat) on f32 rows with split-f16 products ("x16" tiles: every row of `a` is read once, all output
// columns are formed in the same workgroup; the training path's skinny GEMMs).  Weights come as x16 split images built on the device
// by launch_weight_images16 (per-matrix power-of-two scale, its inverse in w_inv); activations are scaled per token inside.
constexpr int kMaxLinMats = 3, kMaxLinJobs = 16;
// counter-based dropout: keep(seed, salt, i) is a pure function, so the backward pass regenerates the forward's mask
struct DropCfg {
    unsigned long long seed;
    unsigned salt;
    float p;                // drop probability, 0 = identity
};
// The mask: 16 uniform bits per element, two elements per 32-bit mix of (seed, salt, index of the PAIR): an odd multiply of the index, the
// launch's key added, murmur3's finaliser; element 2 k takes the low half, 2 k + 1 the high half (callers walk consecutive elements, the
// pair's mix is computed once).  p is honoured to 2^-16.  (Rounds 3-4 ran a splitmix64 finaliser per element - three 64-bit multiplies,
// ~25 instructions with six quarter-rate ones; that was affordable in the element-wise passes, which wait for memory, not in the
// epilogue of the Linear kernels, where the residual + dropout of the training forward now happens: k_linear16.)
__device__ __forceinline__ unsigned drop_key(const DropCfg& d) {
    return (unsigned)d.seed ^ ((unsigned)(d.seed >> 32) * 0x9E3779B1u) ^ (d.salt * 0x85EBCA77u + 0x27D4EB2Fu);
}
__device__ __forceinline__ float drop_scale(const DropCfg& d, unsigned key, unsigned long long idx, float inv_keep) {
    const unsigned long long pair = idx >> 1;
    unsigned h = (unsigned)pair * 0x9E3779B1u + (unsigned)(pair >> 32) * 0xC2B2AE3Du + key;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    const unsigned u = (idx & 1ull) ? h >> 16 : h & 0xffffu;
    return u < (unsigned)(d.p * 65536.f + 0.5f) ? 0.f : inv_keep;
}

struct LinJob {
    const float* a;                      // (M, 256) f32 rows
    const half_t* wimg[kMaxLinMats];     // images of the (256, 256) matrices W: y_m = a W_m^T
    const float* w_inv[kMaxLinMats];     // device scalars written by launch_weight_images16
    const float* bias[kMaxLinMats];      // (256) or null
    float* y[kMaxLinMats];               // (M, ldy) f32
    const float* residual;               // (M, ldy) f32 or null: added to y[0] (may alias it)
    const float* ln_gamma;               // (256) or null: the rows are LayerNorm-ed (eps 1e-5) on their way into the products,
    const float* ln_beta;                //   i.e. y = LN(a) W^T: the normalised rows are never written
    int M, n_mat, ldy;
    const float* a_more[kMaxLinMats - 1];// sum_inputs: the rows of inputs 1 .. n_mat - 1 (input 0 = a)
    int sum_inputs;                      // 1: ONE output, y[0] = sum_m a_m W_m^T + bias[0] (+ residual): the sum stays in registers
                                         //   (the backward's d(xn) = dq W_q + dK' W_k + dV' W_v was three passes over y)
    DropCfg drop;                        // p > 0 (with a residual, one matrix): y = residual + Dropout(a W^T + bias), element index
                                         //   row * ldy + column in the mask's stream (the training forward's two residual adds)
};
struct LinJobs {
    LinJob j[kMaxLinJobs];
    int n;
};
int launch_linear16(const LinJobs& jobs, hipStream_t st);
// n_mat row-major (256, 256) f32 matrices, contiguous -> n_mat images (65536 (hi, lo) pairs each, same byte offsets as the
// matrices) + n_mat inverse scales
int launch_weight_images16(const float* w, half_t* img, float* w_inv, int n_mat, hipStream_t st);

// ---- chain.hip (f16 mode: register-resident token chains) ----
constexpr int kMaxChainJobs = 16;
struct QkvJob {
    const float* x;          // (P, C) token-major, or (C, P) when in_nchw
    float* xs_out;           // token-major copy written when in_nchw (nullptr: not needed)
    const half_t* w[5];      // weight images (NT chunks each): [Q] K'(e0) V'(e0) [K'(e1) V'(e1)]
    void* y[5];              // output planes (P, C): f16, or f32 in split mode
    float c[5];              // split modes: plane = accumulator * c (HmvitStageScales c_q / c_k / c_v; 1 when not in use)
    int n_mat;
    int type;
};
struct QkvParams {
    QkvJob job[kMaxChainJobs];
    const float* gamma;      // (T, C)
    const float* beta;
    int P;
    int in_nchw;
    int* pull;               // split modes, C = 256, optional: one zeroed int - k_ln_qkv16 then runs as pulled tiles (FfnParams::pull)
};
// split: 0 = f16 operands, 1 = split (hi + lo) operands with f32 planes, 2 = split operands with f16 Q / K' / V' / O planes ("mixed")
int launch_ln_qkv(const QkvParams& p, int n_jobs, int C, int split, hipStream_t st);

struct FfnJob {
    const void* o;           // (P, C) attention output (FFN_FULL): f16, or f32 in split mode
    const float* x;          // (P, C) residual stream in
    float* out;              // (P, C) token-major (may alias x) or (C, P) for FFN_HEAD_NCHW
    int type;
    int pad;
    int cls;                 // pulled tiles: job class, set by the launcher (x16_job_classes)
    int x_nchw;              // k_out_ffn_qkv only: x is a (C, P) map (the module input) instead of the token-major stream
    const unsigned char* need;   // optional (H/8, W/8): windows of this agent that a later stage can reach (k_window_need);
                                 // workgroups whose 128 tokens lie in unreachable windows return at once
};
struct FfnParams {
    FfnJob job[kMaxChainJobs];
    const half_t* w_o;       // (T, NT chunks)
    const float* b_o;        // (T, C)
    const float* ln_g;       // (T, C)
    const float* ln_b;
    const half_t* w_ffn;     // (T, 2 NH chunks): W_1 tile 0, W_2 slice 0, W_1 tile 1, ...
    const float* b_1;        // (T, C)
    const float* b_2;        // (T, C)
    int P;
    int W;                   // map width (FfnJob::need only)
    const half_t* w_head;    // k_out_ffn_head only: mlp_head image (T, 2 NH chunks) and biases (T, C)
    const float* hb_1;
    const float* hb_2;
    // split modes: power-of-two range normalisation (HmvitStageScales / HmvitHeadScales, include/hmvit.h); all 1 / dyn_head = 0
    // when the tensors are at their true scale.  b_o arrives pre-divided by c_o and b_2 pre-multiplied by k_2.
    float c_o[HMVIT_NUM_TYPES], c_1[HMVIT_NUM_TYPES], s_g[HMVIT_NUM_TYPES], k_2[HMVIT_NUM_TYPES];
    HmvitHeadScales head;    // mlp_head (k_out_ffn_head, FFN_HEAD_NCHW): operand scaled per token when dyn_head
    int dyn_head;
    // split modes, C = 256, optional: ONE zeroed int per launch - the x16 tails then run one workgroup per CU that pulls (job, tile)
    // tickets from it (chain.hip tail16_pull) instead of one workgroup per tile; n_jobs is set by the launcher
    int* pull;
    int n_jobs;
};
// FfnParams scale fields <- a stage's scales (null: all 1)
inline void set_ffn_scales(FfnParams& p, const HmvitStageScales* sc, const HmvitHeadScales* hs) {
    for (int t = 0; t < HMVIT_NUM_TYPES; ++t) {
        p.c_o[t] = sc ? sc->c_o[t] : 1.f;
        p.c_1[t] = sc ? sc->c_1[t] : 1.f;
        p.s_g[t] = sc ? sc->s_g[t] : 1.f;
        p.k_2[t] = sc ? sc->k_2[t] : 1.f;
    }
    p.dyn_head = hs ? 1 : 0;
    if (hs) p.head = *hs;
    else for (int t = 0; t < HMVIT_NUM_TYPES; ++t) p.head.w1[t] = p.head.w2[t] = 1.f, p.head.l1[t] = p.head.b1max[t] = 0.f;
}
enum { FFN_FULL = 0, FFN_NO_ATTN = 1, FFN_HEAD_NCHW = 2 };
int launch_out_ffn(const FfnParams& p, int n_jobs, int C, int variant, int split, hipStream_t st);
// k_out_ffn (FFN_FULL) of a stage fused with k_ln_qkv of the next one; job j of both lists = the same agent;
// FfnJob::pad = 1 suppresses the store of the updated residual row
int launch_out_ffn_qkv(const FfnParams& p, const QkvParams& q, int n_jobs, int C, int split, hipStream_t st);
// k_out_ffn (FFN_FULL) of the last stage with mlp_head appended; FfnJob::out = (C, P) output map
int launch_out_ffn_head(const FfnParams& p, int n_jobs, int C, int split, hipStream_t st);

// ---- enc.hip (PointPillar branch) ----
struct PfnParams {
    const float* voxels;      // (Nv, 32, 4)
    const int* coords;        // (Nv, 4) [agent, z, y, x]
    const int* num_points;    // (Nv)
    const float* w;           // (64, 10) Linear weight with the BatchNorm scale folded in
    const float* shift;       // (64) BatchNorm shift
    void* canvas;             // (n_agents, ny, nx, 64) NHWC, zero-filled by the caller; may be null
    float* pillar_out;        // (Nv, 64) f32 or null
    int n_pillars, nx, ny;
    int n_agents;             // canvas planes: pillars whose agent / y / x index is out of range are dropped
    int* oob_count;           // device counter of dropped pillars, may be null
    unsigned* canvas_absmax;  // optional, device, zeroed by the caller: atomicMax of the values scattered into the canvas (f32 bits)
    float vx, vy, vz, x_off, y_off, z_off;
};
int launch_pfn_scatter(const PfnParams& p, int precision, hipStream_t st);

struct ConvParams {
    const void* x;            // (N, H, W, Cin) NHWC
    const void* w;            // (Ncols, KH*KW*Cin): k = (ky*KW + kx)*Cin + ci
    const float* bias;        // (Cout) or null
    void* y;                  // NHWC, channel stride y_ctot, first channel y_coff
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, relu;
    int y_ctot, y_coff;
    int deconv_s;             // > 0: ConvTranspose2d with kernel = stride = s (Ncols = s*s*Cout, Ho = H, Wo = W)
    int out_f32;
    const void* res;          // optional residual (N, Ho, Wo, Cout) in the precision's element type, added before the ReLU
    int no_patch;             // 1: never take the patch-in-LDS 3 x 3 kernel (A/B checks of the two kernels against each other)
    int force_patch;          // 1: take the stride-2 ring kernel wherever it applies, also where the generic kernel is as fast (tests)
    int up2;                  // 1: the input is the nearest-neighbour x2 upsampling of x (N, H/2, W/2, Cin); H, W are the upsampled sizes
    // split mode: both operands are brought to [2^13, 2^14) by a power of two on their way into LDS and the product is scaled back.
    const unsigned* x_absmax; // device, f32 bit pattern of max |x| (an upper bound is as good): written by the producer of x
                              // (ConvParams::y_absmax of the convolution before, hmvit_absmax) or, when null, by a pass that
                              // launch_conv runs itself
    float w_absmax;           // > 0: max |w| (host value); < 0: weights pre-multiplied by the power of two -w_absmax; 0: measured
    unsigned* y_absmax;       // optional, device, zeroed by the caller: atomicMax of |y| over everything this launch stores
    const void* w_image;      // optional: the weights as an LDS ring image: kind 0 for k_conv3r (launch_conv3_pack: 3 x 3 / stride 1),
    int w_image_kind;         // kind 1 in the GEMM's k order for the generic kernel (launch_conv_gemm_pack); a kind the chosen kernel does not read is ignored
    int rowpack;              // 1: few-channel stem.  x is a physically zero-padded (N, H, W, 4) map, output pixel (oy, ox) reads
                              // rows oy*stride .. + KH - 1 and pixels ox*stride .. + 7 of it; w is (Ncols, KH * 32) with
                              // k = ky * 32 + px * 4 + ci; Ho, Wo are given, pad / KW / Cin are not used
};
// tok.hip: mode / record_len / mask (any of six dtypes) + the identity check of pairwise[b, l, l] -> int64 words (k_pack_small)
struct SmallPack {
    const void* src[3];
    int dtype[3], n[3];
    const void* pairwise;     // (B, L, L, 4, 4) f32 (pw_dtype 0) or f64 (1), or null: no flag word
    int pw_dtype, B, L;
    long long* out;           // n[0] + n[1] + n[2] (+ 1) words
};
int launch_pack_small(const SmallPack& a, hipStream_t st);
int launch_conv(const ConvParams& p, int precision, hipStream_t st);
// ring image of a 3 x 3 convolution's weights (w: (Cout, 9 Cin) in the precision's element type; split: pre-scaled f32)
size_t conv3_image_size(int Cout, int Cin, int precision);
int launch_conv3_pack(const void* w, int Cout, int Cin, int precision, void* image, hipStream_t st);
size_t conv_gemm_image_size(int Ncols, int Ktot);
int launch_conv_gemm_pack(const float* w, int Ncols, int Ktot, void* image, hipStream_t st);
// atomicMax of max |x| (f32 bit pattern) into slot[0]; the caller zeroes the slot
int launch_absmax(const float* x, size_t n, unsigned* slot, hipStream_t st);
// max pooling on NHWC maps (the 3x3 / stride 2 / pad 1 stage of a ResNet stem)
int launch_maxpool(const void* x, void* y, int N, int H, int W, int C, int ksize, int stride, int pad, int precision, hipStream_t st);
int launch_maxpool_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int ksize, int stride, int pad, hipStream_t st);

// ---- split.hip (architect_mode == 'parallel') ----
struct SplitSlots {
    int8_t s[kMaxSlots];     // agent slots to merge
};
struct SplitWeights {
    const float* fc1;        // (C, C)
    const float* ln_g;       // (C)
    const float* ln_b;
    const float* fc2;        // (2C, C)
};
// partial: (n_slots, ceil(P / 256), C) f32 scratch, w: (n_slots, 2, C) f32 scratch; out may alias a or b
int launch_split_attn(const float* a, const float* b, float* out, const SplitSlots& slots, int n_slots,
                      const SplitWeights& sw, float* partial, float* w, int P, int C, hipStream_t st);

// ---- attn.hip ----
struct AttnParams {
    const void* q;            // (B, L, P, C)
    const void* kv;           // (B, L, E, 2, P, C)
    const float* b_q;         // (T, C)
    const float* b_kv;        // (T_ego, T_src, 2C)
    const float* bias_frag;   // (heads, NB, 64, 4); generic shapes (window not 4 / 8, or dim_head != 32): dense (heads, N, N) bias, N = window^2
    const float* ainv;        // (B, L_src, L_ego, 8): sampling map of pairwise_t[b, src, ego]
    void* out;                // (B, L, P, C)
    int B, L, n_ego, n_src, E, C, H, W, window, partition, skip_masked;
    unsigned long long* trace; // optional s_memtime trace buffer (debug probe), else null
    int variant;              // 0: default kernel choice, 1: force the one-window-per-workgroup kernel
    const unsigned* vis_mask;        // optional (B * n_ego * H/8 * W/8): visible-chunk bits per window (launch_tile_vis)
    int prune;                       // vis_mask bit 31 marks items no later stage can reach: skipped by the schedule
    int self_identity;               // caller's guarantee that every self transform pairwise_t[b, i, i] is the identity
                                     // (needed by the split-precision persistent kernel, which has no general-self loader)
    int rigid_patch;                 // caller's guarantee that every pair transform is a rotation to 2 % (HmvitFusionDesc::rigid_patch):
                                     // the local stages of the split mode may take k_attention_patch
    const void* patch_tab;           // rigid_patch == 2: the items' tables (launch_patch16_tables), B n_ego (H/8) (W/8) blocks of 9 KB
    float* lse;               // optional (B, L, P, heads) f32: log-sum-exp of every query row (f32 kernel; kept for the backward pass)
    const int* sched;         // optional world-ordered item list of the persistent kernels (launch_attn_schedule), n_sched items
    int n_sched, sched_sub;   // sched_sub: steps per list segment (pc_fetch_sched)
    int* tail_pull;           // optional, 48 ints zeroed by k_tile_vis: the pull counters of the tail launches that follow (FfnParams::pull)
    int* queue;               // optional, 16 zeroed ints: per-(XCD, head group) pull counters of k_attention_pcs2's dynamic item
                              // assignment (the workgroups of an XCD pull the XCD's item sequence instead of walking fixed shares)
    float k_logit;            // f32-plane kernels: logits formed from the planes * k_logit = natural units (HmvitStageScales;
                              // 0 is read as 1: descriptors that never heard of it)
    int dim_head;             // channels per head (0 is read as 32); anything but 32, or a window other than 4 / 8, takes the generic
                              // exact-f32 kernel k_attention_any (f32 planes only)
    int8_t mode[kMaxSlots];   // (B, L)
    int8_t cav[kMaxSlots];    // (B, L)
    int8_t ego_e[kMaxSlots];  // (B, L): K/V variant used by ego (b, i)
};
int launch_attention(const AttnParams& p, int precision, hipStream_t st);
int launch_tile_vis(const AttnParams& p, unsigned* vis_mask, const unsigned char* need, hipStream_t st);
int launch_count_live(const unsigned* vis_mask, int n, int* out, hipStream_t st);   // += items with bit 31 clear
// world-ordered work list of a local-partition stage with p.n_ego egos (p.ainv, B, L, H, W are read); ws: attn_schedule_bytes
size_t attn_schedule_bytes(int B, int n_ego, int H, int W);
int launch_attn_schedule(const AttnParams& p, int* ws, hipStream_t st);
// to[(b * n_ego + k) * (H/8) * (W/8) + window] = 1 when an ego reads a key / value inside that 8 x 8 window of agent k's
// map: ego 0 over its whole map (from = nullptr: the reads of the pruned last stage), or every ego j over the windows
// marked in `from` (plus those windows themselves).  `to` must be zeroed by the caller.
int launch_window_need(const AttnParams& p, const unsigned char* from, unsigned char* to, hipStream_t st);
int launch_debug_tr16(uint16_t* out, hipStream_t st);
// k_attention_patch16's per-item tables (they depend on the pair transforms only: once per forward); ws: patch16_tables_bytes
size_t patch16_tables_bytes(int B, int n_ego, int H, int W);
int launch_patch16_tables(const AttnParams& p, void* ws, hipStream_t st);

// ---- train.hip (backward pass of the fusion, exact-f32 MFMA; SURVEY 8b "autograd must flow") ----
// dw[n][k] += sum_m dy[m][n] a[m][k]  (weight gradient of y = a w^T), dbias[n] += sum_m dy[m][n]; f32 atomics
struct GemmTnJob {
    const float* dy;        // (M, N) row stride ld_dy
    const float* a;         // (M, K) row stride ld_a
    float* dw;              // (N, K) row-major, accumulated into
    float* dbias;           // (N) accumulated into, or null
    int M, N, K, ld_dy, ld_a;
};
struct GemmTnJobs {
    GemmTnJob j[kMaxJobs];
    int n;
};
int launch_gemm_tn(const GemmTnJobs& jobs, hipStream_t st);
// dx = dres + LayerNorm-backward(dy; x, gamma[type]); dgamma / dbeta (T, C) accumulated with atomics
int launch_layernorm_bwd(const float* x, const float* dy, const float* gamma, const AgentTypes& types, int n_agents,
                         const float* dres, float* dx, float* dgamma, float* dbeta, int P, int C, hipStream_t st);
int launch_add_drop(const float* x, const float* a, float* y, size_t n, DropCfg d, hipStream_t st);      // y = x + drop(a); x may be null
int launch_gelu_drop(const float* pre, float* h, size_t n, DropCfg d, hipStream_t st);                   // h = drop(gelu(pre))
int launch_gelu_bwd(const float* pre, const float* dh, float* dpre, size_t n, DropCfg d, hipStream_t st); // dpre = drop'(dh) gelu'(pre)
int launch_dropout_mask(float* mask, size_t n, DropCfg d, hipStream_t st);                               // mask[i] = 0 or 1 / (1 - p)
struct AttnBwdParams {
    AttnParams f;             // the forward launch (q, kv, biases, bias_frag, ainv, out = O, lse, geometry)
    const float* bias_frag_neg;   // bias fragments of the negated offset table: the S (not S^T) tiles' bias
    const float* d_out;       // (B, L, P, C) gradient of O
    float* dq;                // (B, L, P, C) gradient of the (un-biased) q planes, ego slots
    float* dkg;               // (B, n_ego, n_src, 2, P, C): gradient of the gathered K / V keys of every (ego, source) pair,
                              // indexed by EGO pixel; every key row of every pair is written (zeros where nothing is visible)
    float* d_bias_frag;       // (heads, NB, 64, 4) accumulated into
    const float* v_bound;     // device scalar: a-priori bound on the 2-norm of a head's slice of a V' / O row (launch_v_bound), or null
    int probe;                // probe builds (HMVIT_BWD_PROBE): 1 = skip the products, 3 = skip the prologue's dot products
};
int launch_attention_bwd(const AttnBwdParams& p, hipStream_t st);
// out[0] = max over the V' matrices (w_kv[(pair, 1)], pair < n_pairs) and heads of ||W_head||_F ||xn||_2 + ||b_head||_2 with
// ||xn||_2 <= sqrt(C) (max|gamma| + max|beta|): an a-priori bound - weights only - on the 2-norm of a head's 32-channel slice of
// every V' row and, the attention output being a convex combination of V' rows, of every O row
int launch_v_bound(const float* w_kv, const float* b_kv, const float* ln_gamma, const float* ln_beta, int n_pairs, int n_types, int C,
                   float* out, hipStream_t st);
// adjoint of the bilinear key gather: dkv[(b, src), e, plane, s, :] = sum over egos of variant e and ego pixels u whose taps
// touch source pixel s of weight(u -> s) dkg[(b, ego, src), plane, u, :]   (gather form, no atomics)
struct WarpAdjParams {
    const float* dkg;         // as AttnBwdParams::dkg
    const float* ainv;        // (B, L_src, L_ego, 8)
    float* dkv;               // (B, L, E, 2, P, C)
    int B, L, n_ego, n_src, E, C, H, W;
    int8_t ego_e[kMaxSlots];
    // column sums of dkg on the way (the K' / V' bias gradients); null: not wanted
    float* db_kv;             // (T, T, 2, C): [type of the ego][type of the source][plane]
    float* db_rep;            // scratch, warp_adjoint_replica_floats(T, C) floats: the workgroups' partial sums before they are folded
    int T;
    int8_t mode[kMaxSlots];   // (B, L) agent types
};
int launch_warp_adjoint(const WarpAdjParams& p, hipStream_t st);
size_t warp_adjoint_replica_floats(int T, int C);
// training-mode BatchNorm (+ ReLU) on (M, C) maps: csrc/train.hip k_bn_reduce / k_bn_apply
struct BnArgs {
    const float* x;        // (M, C) pre-normalisation values
    const float* y;        // (M, C) forward output (backward: the ReLU mask), or null without ReLU
    const float* dy;       // (M, C) backward only
    const float* mean;     // (C)
    const float* rstd;     // (C)
    const float* gamma;    // (C)
    const float* beta;     // (C) forward only
    const float* sums;     // (2 C) backward apply: sum g, sum g xhat
    float* out;            // reduce: (2 C) accumulated;  apply: (M, C)
    int M, C, relu;
};
int launch_bn(const BnArgs& a, int bwd, int apply, hipStream_t st);

// ---- post.hip (detection post-processing) ----
struct BoxDecodeParams {
    const float* psm;       // (A, H, W) logits
    const float* rm;        // (7A, H, W) anchor deltas
    const float* anchors;   // (H, W, A, 7)
    const float* T;         // (4, 4) row-major projection into the ego frame, or null
    int H, W, A;
    float thresh;
    int order_hwl;
    float* corners;         // (capacity, 8, 3)
    float* scores;          // (capacity)
    int* index;             // (capacity) anchor index of the candidate
    int* count;             // device counter, zeroed by the launcher's caller
    int capacity;
};
int launch_box_decode(const BoxDecodeParams& p, hipStream_t st);
int launch_quad_iou(const float* a, const float* b, int na, int nb, int stride_box, int stride_pt, float* iou, hipStream_t st);
int launch_nms_rotated(const float* corners, const float* scores, const int* index, int n, float thresh, const float* range4,
                       int* rank, int* sorted, float* iou, int* keep, int* n_keep, hipStream_t st);

// ---- vox.hip (pillariser) ----
struct VoxParams {
    const float* points;    // (n_points, 4)
    int n_points;
    float rmin[3], vsize[3];
    int nx, ny, nz, max_points, max_voxels;
    // workspace
    int* cell;              // (n_points) linear cell or -1
    int* placed;            // (n_points)
    int* scan;              // (n_points)
    int* first;             // (cells) first point of the cell
    int* count;             // (cells)
    int* cmin;              // (cells) bidding array
    int* vox_id;            // (cells)
    // outputs
    float* voxels;          // (max_voxels, max_points, 4), zero padded
    int* coords;            // (max_voxels, 3) z, y, x
    int* num_points;        // (max_voxels)
    int* n_voxels;          // device scalar
};
int launch_voxelize(const VoxParams& p, hipStream_t st);

// ---- cvt.hip (camera -> BEV lift) ----
struct CvtEmbedParams {
    int mode;               // 0: image-ray embedding of feature pixels, 1: BEV query embedding (+ x)
    int bn, n_cam, P, H, W, dim;   // bn = agents * cameras; P = H * W tokens per map
    float img_w, img_h;
    const float* I_inv;     // (bn, 3, 3), mode 0
    const float* E_inv;     // (bn, 4, 4)
    const float* grid;      // (2+, P) BEV cell coordinates, mode 1
    const float* w_in;      // (dim, 4) img_embed or (dim, 2) bev_embed
    const float* w_bias;    // (dim) bev_embed bias or null
    const float* w_cam;     // (dim, 4) cam_embed
    const float* x;         // (agents, dim, P) added to the query embedding, or null
    float* out;             // (bn, P, dim)
};
int launch_cvt_embed(const CvtEmbedParams& p, hipStream_t st);
int launch_bn_relu_tokens(const float* x, const float* scale, const float* shift, float* y, int n, int C, int P, hipStream_t st);
int launch_cross_attention(const float* q, const float* k, const float* v, float* out, int b, int n_cam, int Q, int K, int heads,
                           int dim_head, const float* bias, hipStream_t st, float* lse = nullptr);
// backward of the joint-softmax cross attention: dq (b, n_cam, Q, HD), dk (b, n_cam, K, HD), dv (b, n_cam K, HD)
int launch_cross_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* d_out,
                               float* dq, float* dk, float* dv, int b, int n_cam, int Q, int K, int heads, int dim_head, hipStream_t st,
                               const float* bias = nullptr, float* d_bias = nullptr);   // bias (heads, Q, K), n_cam = 1: d_bias written
int launch_cross_attention_f16(const half_t* q, const half_t* k, const half_t* v, float* out, int b, int n_cam, int Q, int K,
                               int heads, int dim_head, hipStream_t st);
int launch_cross_attention_split(const float* q, const float* k, const float* v, float* out, int b, int n_cam, int Q, int K,
                                 int heads, int dim_head, hipStream_t st);

}  // namespace hmvit
