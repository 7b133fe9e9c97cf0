/* hmvit.h -- C ABI of the MI355X-native HM-ViT fusion hot path (libhmvit.so).
 *
 * The reference has no FFI for this path: it is pure Python/PyTorch, and its GPU arithmetic
 * comes from torch (cuBLAS/cuDNN).  The entry points below are what a binding for the path
 * would call; each one names the reference code it replaces (paths relative to the reference
 * repository root).  Conventions:
 *   - plain pointers and sizes only, no torch / C++ types;
 *   - every pointer marked "device" is HBM memory of the current HIP device;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*) and is stream-ordered;
 *     nothing here synchronises or allocates device memory: scratch comes from the caller
 *     (`workspace`, size from hmvit_fusion_workspace_bytes);
 *   - return 0 on success, a negative HMVIT_E* code otherwise; hmvit_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef HMVIT_H
#define HMVIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMVIT_ABI_VERSION 12

#define HMVIT_OK 0
#define HMVIT_EINVAL (-22)   /* bad argument / unsupported shape */
#define HMVIT_ENOMEM (-12)   /* workspace too small */
#define HMVIT_EHIP (-5)      /* a HIP call failed */

#define HMVIT_MAX_AGENTS 8   /* L (max_cav) upper bound */
#define HMVIT_NUM_TYPES 2    /* 0 = camera, 1 = lidar (base_camera_lidar_dataset.py:136,178) */

/* arithmetic modes */
#define HMVIT_PREC_F32 0     /* f32 MFMA everywhere: strict parity mode */
#define HMVIT_PREC_F16 1     /* f16 operands, f32 accumulate, f32 LN/softmax/residual */
#define HMVIT_PREC_SPLIT 2   /* fp32-class products on the f16 matrix pipes: every operand x = hi + lo (two f16 halves),
                                three MFMAs per product, f32 accumulate; Q / K' / V' / O planes are f32.  Reads the
                                fragment images of HMVIT_PREC_F16 with, per k-step, the hi fragment followed by the lo
                                fragment (twice the size; weights.py weight_image(split=True)); w_q / b_q / bias_frag carry
                                no log2(e) factor.  Held to the f32 tolerance (1e-4) by the parity tests.
                                In the single operators: hmvit_conv2d / _ex / _rowpack take f32 maps and f32 weights (exactly the
                                HMVIT_PREC_F32 layouts) and split each K slab on the way into LDS; hmvit_linear needs f32 a / w / y
                                (out_f32 = 1); hmvit_pfn_scatter / hmvit_maxpool2d treat it as HMVIT_PREC_F32 (f32 maps). */
#define HMVIT_PREC_MIXED 3   /* HMVIT_PREC_SPLIT arithmetic in every Linear / FFN / LayerNorm / residual (the token chains), but the
                                attention operands Q / K' / V' / O are stored as f16 planes and the f16 attention kernels run on
                                them (f32 accumulate and softmax): half the attention's HBM traffic.  C = 256 only (other widths
                                behave as HMVIT_PREC_SPLIT); w_q / b_q / bias_frag carry log2(e) as in HMVIT_PREC_F16.  Held to the
                                f32 tolerance (1e-4) by the parity tests on the reference's goldens; the attention-operand rounding
                                contributes < 1e-5 there. */

/* partition of one attention stage (hetero_fusion.py:387-389 vs :430-431) */
#define HMVIT_PART_WINDOW 0  /* 'b m d (x w1) (y w2)': contiguous w x w windows */
#define HMVIT_PART_GRID 1    /* 'b m d (w1 x) (w2 y)': dilated grid */

/* Weights of one attention stage, already folded by the host
 * (hm-vit_amd/weights.py; identities (i)-(ii) of SURVEY.md 8a).  Element type is f32 or
 * f16 according to the precision mode, except where noted.  All device pointers.
 *   ln_*      (T, C)            f32   HeteroLayerNorm affine, base_transformer.py:172-177
 *   w_q       (T, C, C)               q_linears[t].weight * dim_head^-0.5, hetero_fusion.py:111-140,217
 *                                     (f16 mode: w_q, b_q and bias_frag additionally carry log2(e): the
 *                                     f16 kernels evaluate the softmax with exp2)
 *   b_q       (T, C)            f32   q_linears[t].bias   * dim_head^-0.5
 *   w_kv      (T_ego, T_src, 2C, C)   rows [0,C): blockdiag_h(relation_att[e,h]) k_linears[ts].weight
 *                                     rows [C,2C): blockdiag_h(relation_msg[e,h]^T) v_linears[ts].weight,
 *                                     e = t_ego*2 + t_src  (hetero_fusion.py:154-185,221-223,263-264)
 *   b_kv      (T_ego, T_src, 2C) f32  the same maps applied to the k / v biases
 *   bias_frag (heads, NB, 64, 4) f32  relative-position bias in MFMA accumulator order,
 *                                     NB = 7 (window 8) or 1 (window 4), hetero_fusion.py:82-109,227-233.
 *                                     Generic shapes (window not 4 / 8, up to 16; or dim_head != 32, up to 64; HMVIT_PREC_F32,
 *                                     inference only): the dense bias (heads, N, N), N = window^2, [h][query][key]
 *   w_o, b_o  (T, C, C), (T, C)       a_linears[t][0], hetero_fusion.py:142-152
 *   ffn_ln_*  (T, C)            f32   {window,grid}_ffd.norm, base_transformer.py:129-136
 *   w_1,b_1   (T, mlp, C),(T, mlp)    {window,grid}_ffd.fn.net[t][0], base_transformer.py:180-192
 *   w_2,b_2   (T, C, mlp),(T, C)      {window,grid}_ffd.fn.net[t][3]
 * Biases are always f32.
 *
 * HMVIT_PREC_F32 reads the plain row-major matrices w_q, w_kv, w_o, w_1, w_2 (f32).
 * HMVIT_PREC_F16 reads "fragment images" instead (the plain pointers may then be NULL): an
 * (N, K) matrix is stored as (N/32, K/16, 64, 8) f16 with
 *     img[t][kk][lane][4 jj + i] = W[32 t + (lane & 31)][16 kk + 8 jj + 4 (lane >> 5) + i],
 * i.e. one 32-row tile after another, inside a tile one v_mfma_f32_32x32x16_f16 A-operand
 * fragment after another, so that weight staging is a linear copy (hm-vit_amd/weights.py:
 * weight_image).  img_q (T, ...), img_kv (T_ego, T_src, ...) of the (2C, C) matrix, img_o (T, ...);
 * img_q and img_kv (results stored to memory as f16) take row(r) = 16 (j >> 1) + 8 hi + 4 (j & 1) + i
 * for r = 8 j + 4 hi + i in place of (lane & 31), which makes the 16 results of an accumulator lane two
 * runs of 8 consecutive channels = two 16-byte stores (weights.py: store_row_order);
 * img_o (operand read from memory) takes the K index 16 kk + 8 (lane >> 5) + 4 jj + i instead;
 * img_ffn (T, C/32, 2, C/16, 64, 8) interleaves, for every hidden tile hc, the image of
 * W_1 rows [32 hc, 32 hc + 32) with the fragments (t, 2 hc + s), t < C/32, s < 2, of the image of
 * W_2 (requires mlp_dim == C). */
/* Range normalisation of the split-operand modes (HMVIT_PREC_SPLIT / HMVIT_PREC_MIXED).  A split product forms x = hi + lo
 * from two f16 halves, and f16 has 5 exponent bits: operands above 65504 overflow, operands whose lo half falls below 2^-14
 * lose it.  So every operand of those modes is carried at a power-of-two multiple that puts its STATIC bound (LayerNorm
 * output: sqrt(C) max|gamma| + max|beta|; a Linear of a LayerNorm output: Cauchy-Schwarz over the weight rows; GELU(h): |h|)
 * just below 2^14, and the products are brought back by the factors below - all exact powers of two, computed once per
 * parameter version by hm-vit_amd/weights.py:fold_stage, which also pre-scales the tensors it applies to (ln_*, b_q, b_kv,
 * bias_frag, b_o, b_1, b_2 and the weight images).  The only data-dependent operand, the un-normalised residual row that
 * mlp_head multiplies (bevformer_point_pillar_hetero.py:48), is scaled per token inside the kernel (HmvitHeadScales).
 * A null `scales` pointer means "all 1": tensors at their true scale (HMVIT_PREC_F32 / HMVIT_PREC_F16, training). */
typedef struct HmvitStageScales {
    float c_q[HMVIT_NUM_TYPES];                     /* Q plane  = accumulator * c_q[t], t = agent type                      */
    float c_k[HMVIT_NUM_TYPES][HMVIT_NUM_TYPES];    /* K' plane = accumulator * c_k[t_ego][t_src]                           */
    float c_v[HMVIT_NUM_TYPES][HMVIT_NUM_TYPES];
    float k_logit;                                  /* logits formed from the planes * k_logit = natural units (one value
                                                       per stage: bias_frag is pre-multiplied by 1 / k_logit)              */
    float c_o[HMVIT_NUM_TYPES];                     /* x' = x + accumulator * c_o[t]  (b_o pre-divided by c_o)              */
    float c_1[HMVIT_NUM_TYPES];                     /* h  = accumulator * c_1[t]      (b_1 pre-divided by c_1)              */
    float s_g[HMVIT_NUM_TYPES];                     /* GELU(h) * s_g[t] = operand of the second Linear                      */
    float k_2[HMVIT_NUM_TYPES];                     /* the residual row is carried as x * k_2[t] while W_2's products
                                                       accumulate into it (b_2 pre-multiplied by k_2)                      */
} HmvitStageScales;
typedef struct HmvitHeadScales {                    /* mlp_head: operand scaled per token, 2^e with |row| 2^e < 2^14        */
    float w1[HMVIT_NUM_TYPES], w2[HMVIT_NUM_TYPES]; /* power-of-two multiples the two weight images are stored at           */
    float l1[HMVIT_NUM_TYPES];                      /* max row L1 norm of W_1 (true scale): |h| <= l1 max|x| + b1max        */
    float b1max[HMVIT_NUM_TYPES];
} HmvitHeadScales;

typedef struct HmvitStageWeights {
    const float* ln_gamma;
    const float* ln_beta;
    const void* w_q;
    const float* b_q;
    const void* w_kv;
    const float* b_kv;
    const float* bias_frag;
    const void* w_o;
    const float* b_o;
    const float* ffn_ln_gamma;
    const float* ffn_ln_beta;
    const void* w_1;
    const float* b_1;
    const void* w_2;
    const float* b_2;
    const void* img_q;
    const void* img_kv;
    const void* img_o;
    const void* img_ffn;
    const HmvitStageScales* scales;   /* HOST pointer (read during the call), or NULL = all 1 */
} HmvitStageWeights;

/* One HeteroFusion / HeteroFusionBlock forward.
 * Replaces HeteroFusion.forward (opencood/models/bevformer_point_pillar_hetero.py:39-49) when
 * apply_head = 1 and HeteroFusionBlock.forward (sequential or parallel mode,
 * opencood/models/sub_modules/hetero_fusion.py:446-474) iterated num_iters times when
 * apply_head = 0. */
typedef struct HmvitFusionDesc {
    int32_t B, L, C, H, W;        /* x is (B, L, C, H, W) f32, NCHW per agent            */
    int32_t heads, dim_head;      /* C = heads * dim_head; dim_head must be 32            */
    int32_t window;               /* 4 or 8; H and W must be divisible by it              */
    int32_t mlp_dim;              /* FFN hidden width                                     */
    int32_t num_iters;            /* the same block weights are applied num_iters times   */
    int32_t precision;            /* HMVIT_PREC_*                                         */
    int32_t apply_head;           /* 1: ego slice + mlp_head -> out (B, C, H, W)
                                     0: out (B, L, C, H, W), all agents                   */
    int32_t skip_masked;          /* 0: everything computed, masked keys at -inf; 1 (default): key tiles whose 64 keys are all
                                     masked and windows no later stage can reach are skipped (both exact: identical output);
                                     2: masked key tiles only */
    float discrete_ratio;         /* spatial_transform.voxel_size[0]                      */
    float downsample_rate;        /* spatial_transform.downsample_rate                    */
    /* host arrays (read during the call, not retained) */
    const int32_t* mode;          /* (B, L) agent types; padding = 0                      */
    const int32_t* record_len;    /* (B)                                                  */
    const int32_t* cav_mask;      /* (B, L) 1 = real agent                                */
    /* device arrays */
    const float* x;               /* (B, L, C, H, W)                                      */
    const float* pairwise_t;      /* (B, L, L, 4, 4), [b,i,j] maps agent i into agent j   */
    float* out;
    HmvitStageWeights stage[2];   /* [0] window (local), [1] grid (global)                */
    const void* head_w1;          /* mlp_head.net[t][0]: (T, C, C)                        */
    const float* head_b1;         /* (T, C)                                               */
    const void* head_w2;          /* mlp_head.net[t][3]: (T, C, C)                        */
    const float* head_b2;
    const void* head_img_ffn;     /* f16 mode: (T, ...) image of mlp_head as img_ffn above */
    void* workspace;              /* device scratch                                       */
    size_t workspace_bytes;
    /* architect_mode == 'parallel' (hetero_fusion.py:459-470): local and global stage both start
     * from the block input and are merged by SplitAttn (fusion_modules/split_attn.py:32-67).
     * All f32 device arrays: fc1 (C, C), LayerNorm affine (C), fc2 (2C, C); no biases. */
    int32_t parallel;             /* 0 = sequential (shipped yaml), 1 = parallel          */
    const float* split_fc1;
    const float* split_ln_g;
    const float* split_ln_b;
    const float* split_fc2;
    const HmvitHeadScales* head_scales;   /* HOST pointer, or NULL = head_img_ffn at its true scale and no per-token scaling */
    int32_t self_identity;        /* 1: the caller guarantees pairwise_t[b, i, i] = I for every agent (what the reference's
                                     datasets produce, mixed/intermediate_fusion_dataset.py:163-202); lets HMVIT_PREC_SPLIT use
                                     its persistent attention kernel.  0: unknown (always correct, slower in split mode) */
    int32_t rigid_patch;          /* 1 / 2: the caller guarantees that the upper-left 2 x 2 block M of every pairwise_t[b, i, j] is a
                                     rotation to 2 % (|M^T M - I| <= 0.02 elementwise; the reference's poses are rigid: x / y / yaw and
                                     small roll / pitch).  Then the 32 keys of a half window never touch more than 64 source pixels (a
                                     16-key block: 40) and the local stages of HMVIT_PREC_SPLIT run a de-duplicated patch kernel:
                                     1 = k_attention_patch (8 wavefronts per window), 2 = k_attention_patch16 (16; its per-item tables
                                     come from a pre-pass once per forward).  Same results as the gather kernel to fp32 round-off; both
                                     measured ~10 % slower than it at cfg2, hence opt-in.
                                     0: unknown / not wanted (always correct: the gather kernel).  hmvit_pack_small reports both
                                     guarantees. */
} HmvitFusionDesc;

int hmvit_abi_version(void);
const char* hmvit_last_error(void);

/* bytes of device scratch hmvit_fusion_forward needs for this descriptor (0 on bad input) */
size_t hmvit_fusion_workspace_bytes(const HmvitFusionDesc* desc);

/* bevformer_point_pillar_hetero.py:39-49 / hetero_fusion.py:446-458 (see HmvitFusionDesc) */
int hmvit_fusion_forward(const HmvitFusionDesc* desc, void* stream);

/* Same forward with one HIP event recorded (on `stream`) after each phase; synchronises the
 * stream before returning (the only entry point that does).  phase_ms[p] = milliseconds spent
 * in phase p summed over its occurrences, phase_launches[p] = number of occurrences (e.g. 4
 * attention launches for num_iters = 2).  Both arrays have HMVIT_NUM_PHASES entries. */
#define HMVIT_PHASE_LAYOUT_IN 0   /* NCHW -> token-major + pair affines                    */
#define HMVIT_PHASE_LN_ATTN 1     /* HeteroLayerNorm before attention                       */
#define HMVIT_PHASE_QKV 2         /* Q / folded K,V projection GEMMs                        */
#define HMVIT_PHASE_ATTENTION 3   /* fused warp + partition + attention                     */
#define HMVIT_PHASE_OUT_PROJ 4    /* a_linears GEMM + residual                              */
#define HMVIT_PHASE_LN_FFN 5      /* HeteroLayerNorm of the FFN                             */
#define HMVIT_PHASE_FFN1 6        /* Linear + GELU                                          */
#define HMVIT_PHASE_FFN2 7        /* Linear + residual                                      */
#define HMVIT_PHASE_HEAD 8        /* mlp_head (two GEMMs); fused modes at C = 256: the LAST stage's
                                     tail launch with mlp_head appended (k_out_ffn_head)      */
#define HMVIT_PHASE_LAYOUT_OUT 9  /* token-major -> NCHW                                    */
#define HMVIT_NUM_PHASES 10
int hmvit_fusion_profile(const HmvitFusionDesc* desc, void* stream, float* phase_ms,
                         int32_t* phase_launches);
/* (ego, window) attention items of every stage of this thread's last hmvit_fusion_profile, in launch order: live[i] = items
 * the launch ran (what the reachability pruning of the last two stages left: DESIGN 5.1), total[i] = egos x windows of the
 * stage.  Returns the number of stages written (<= capacity).  bench.py prices the attention kernel's roofline on these. */
int hmvit_fusion_profile_items(int32_t* live, int32_t* total, int capacity);

/* ---- training: forward that keeps its activations + backward ("autograd must flow to x and all used
 * parameters").  The reference trains this path through torch.autograd (opencood/tools/train_camera.py:163-199, Dropout 0.1
 * after the out-projection and inside the FFN, hetero_fusion.py:65-66, base_transformer.py:186-192).  Exact-f32 arithmetic,
 * sequential block, apply_head = 1; gradients are with respect to the FOLDED weights of HmvitStageWeights (the host maps them
 * back to the reference's parameters through the fold, hm-vit_amd/weights.py + autograd).  Every gradient buffer is
 * ACCUMULATED into (the two iterations share the block weights) and must be zero-filled by the caller, f32, same shapes as
 * the corresponding HmvitStageWeights field in HMVIT_PREC_F32 layout.
 *
 * Dropout: element i of the activation of agent slot s = b * L + l in stage k (k = 2 * iteration + {0 window, 1 grid}) is
 * kept iff u(seed + 0x51ED270B1 * (s + 1), salt = 4 k + which, i) >= drop_p, which = 0 out-projection, 1 FFN hidden,
 * 2 FFN output; hmvit_dropout_mask writes exactly that mask (0 or 1 / (1 - p)) so a checker can replay it. */
typedef struct HmvitStageGrads {
    float* ln_gamma;
    float* ln_beta;
    float* w_q;
    float* b_q;
    float* w_kv;
    float* b_kv;
    float* bias_frag;
    float* w_o;
    float* b_o;
    float* ffn_ln_gamma;
    float* ffn_ln_beta;
    float* w_1;
    float* b_1;
    float* w_2;
    float* b_2;
} HmvitStageGrads;

typedef struct HmvitFusionTrainDesc {
    HmvitFusionDesc fwd;             /* precision = HMVIT_PREC_F32, apply_head = 1, parallel = 0; fwd.workspace: scratch of
                                        max(B L P C, B L P mlp) floats; pairwise_t[b, i, i] must be the identity            */
    float drop_p;                    /* hetero_fusion_block.drop_out in training, 0 in eval                                */
    uint64_t seed;                   /* dropout stream of this step                                                        */
    void* saved;                     /* device: activations kept for the backward pass                                     */
    size_t saved_bytes;              /* >= hmvit_fusion_train_saved_bytes                                                  */
    const void* bias_frag_neg[2];    /* per stage, device (heads, NB, 64, 4) f32: bias_frag of the table with negated
                                        offsets (table flipped along its first axis); read by the backward pass only       */
    int32_t only_stage;              /* 0: HeteroFusion (fwd.apply_head = 1).  1 / 2: ONE stage of the block on its own - the window
                                        (local) or the grid (global) stage - with fwd.apply_head = 0: out and d_out are
                                        (B, L, C, H, W), every agent runs as an ego, no mlp_head (its gradient pointers may be
                                        NULL).  The branches of architect_mode 'parallel' (hetero_fusion.py:459-470) train
                                        through this form; their SplitAttn merge is host-side autograd (hm-vit_amd/train.py)  */
    int32_t recompute;               /* memory for time.  Bit 0: the FFN pre-activations W_1 LN(x') + b_1 of every stage are not kept for the
                                        backward pass; bit 1: neither are the queries LN(x) W_q.  The backward recomputes them with the
                                        forward's own kernels and weight images (the same rows bit for bit).  Each bit takes B L P C
                                        (mlp) floats per stage off hmvit_fusion_train_saved_bytes (cfg2: 19.8 -> 17.1 -> 14.5 GiB; peak
                                        of a step 32.8 -> 30.1 -> 27.4 GiB) and adds one Linear per stage to the backward (cfg2, whole
                                        step, cumulative: 92.8 -> 94.4 -> 96.8 ms, i.e. + 1.6 ms with bit 0, + 4.0 ms with both; HISTORY.md
                                        12.8).  With any bit set fwd.workspace must hold 2 x max(B L P C, B L P mlp) floats.  Values
                                        other than 0 .. 3 are rejected; the backward checks saved_bytes against the plan of ITS
                                        descriptor, so the bits must be those of the forward                                   */
} HmvitFusionTrainDesc;

size_t hmvit_fusion_train_saved_bytes(const HmvitFusionTrainDesc* desc);
size_t hmvit_fusion_backward_workspace_bytes(const HmvitFusionTrainDesc* desc);
/* HeteroFusion.forward in training mode: fwd.out (B, C, H, W) */
int hmvit_fusion_train_forward(const HmvitFusionTrainDesc* desc, void* stream);
/* d_out (B, C, H, W) -> d_x (B, L, C, H, W) (padded agents: 0), grads[2] (window, grid), mlp_head gradients (T, C, C) / (T, C).
 * `desc` must be the descriptor of the matching hmvit_fusion_train_forward call (same saved area, seed, weights).
 * Range (ABI-compatible behaviour change in round 5): d_out may have any magnitude in [2^-40, 2^40] times that of the forward's
 * activations (the per-slab operand scale is a power of two clamped to 2^+-40, and the combined accumulator rescale of the
 * weight-gradient products stays finite inside that window; the Python binding brings max |d_out| to 2^9 first, exactly, and scales
 * the results back).  The products of the pass run on split-f16
 * operands, and every kernel scales its own operands by exact powers of two taken from the data (per token row, per 32-token
 * slab, per workgroup against a weights-only bound on |V'|): one pass, no non-finite-detect-and-repeat, nothing read back by the
 * host.  Preconditions are the forward's: the stage's Q, K', V' themselves must be f16-representable (|.| < 65504).
 * window_size 4 / 8 with dim_head 32 take the tuned kernels; every other (window <= 16, dim_head <= 64 dividing C) trains through
 * the generic exact-f32 attention kernels, with bias_frag / grads[.].bias_frag in the dense (heads, N, N) layout and bias_frag_neg
 * unused (may be NULL). */
int hmvit_fusion_backward(const HmvitFusionTrainDesc* desc, const float* d_out, float* d_x, const HmvitStageGrads* grads,
                          float* d_head_w1, float* d_head_b1, float* d_head_w2, float* d_head_b2, void* workspace,
                          size_t workspace_bytes, void* stream);
int hmvit_dropout_mask(float* mask, size_t n, uint64_t seed, uint32_t salt, float p, void* stream);

/* ---- training-mode operators of the detection tail (hetero_decoder.py:7-89 / naive_decoder.py:45-54 under train()) ---- */

/* Weight / bias gradient of a Linear or of one tap of a convolution: dw (N, K) += dy^T a, dbias (N) += column sums of dy
 * (dbias may be NULL).  dy (M, N) with row stride ld_dy, a (M, K) with row stride ld_a, f32; N, K, ld multiples of 4; the
 * products run on split-f16 operands (fp32-class accuracy).  dw / dbias are ACCUMULATED into (atomics). */
int hmvit_gemm_tn(const float* dy, const float* a, float* dw, float* dbias, int M, int N, int K, int ld_dy, int ld_a, void* stream);

/* y (M, 256 n_mat) = a (M, 256) W^T (+ bias) (+ residual), W (256 n_mat, 256) row-major f32, n_mat <= 3: the training path's
 * skinny Linear (torch.nn.Linear forward, hetero_fusion.py:142-152 / base_transformer.py:129-192 under train()), products on
 * split-f16 operands with per-token and per-matrix power-of-two scaling (exact; any magnitude of a / W).  `image_ws` takes the
 * weights' device-built operand images: 65536 n_mat + 64 floats.  residual (M, 256), n_mat == 1 only, may alias y. */
int hmvit_linear16(const float* a, const float* w, const float* bias, const float* residual, float* y, int M, int n_mat,
                   float* image_ws, void* stream);

/* nn.BatchNorm2d on batch statistics + ReLU over NHWC maps viewed as (M = N H W, C), C % 4 == 0, f32:
 *   stats:    sums[c] += sum_m x[m][c], sums[C + c] += sum_m x[m][c]^2   (zero-fill sums first; the caller forms mean / rstd)
 *   apply:    y = relu?(gamma (x - mean) rstd + beta)
 *   backward: g = dy [y > 0] (relu) or dy; sums[c] += sum g = dbeta, sums[C + c] += sum g xhat = dgamma (zero-fill first);
 *             dx = gamma rstd (g - dbeta / M - xhat dgamma / M) */
int hmvit_bn_train_stats(const float* x, float* sums, int M, int C, void* stream);
/* the same sums of (x - pivot[c]): with pivot = the batch mean of a first pass, sums[C + c] / M - (sums[c] / M)^2 is the variance without
 * the cancellation of E[x^2] - mean^2 (channels whose |mean| dwarfs their spread, e.g. behind a convolution bias) */
int hmvit_bn_train_stats_centered(const float* x, const float* pivot, float* sums, int M, int C, void* stream);
int hmvit_bn_train_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                         int M, int C, int relu, void* stream);
int hmvit_bn_train_backward(const float* x, const float* y, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            float* sums, float* dx, int M, int C, int relu, void* stream);

/* ---- single operators (used by the parity tests; same kernels as the fused forward) ---- */

/* The small integer inputs of HeteroFusion.forward - mode (B, L), record_len (B), mask (B, L), in whatever dtype the caller holds
 * them on the device (codes: 0 f32, 1 f64, 2 i32, 3 i64, 4 u8 / bool, 5 f16) - as int64 words out[0 .. n_mode + n_rl + n_mask), so
 * that the host reads them with one copy (the launch plan needs them: hetero_fusion.py:127-131 reads them back element by element).
 * pairwise (B, L, L, 4, 4) f32 (pw_dtype 0) / f64 (1) or NULL: one more word of flags - bit 0: every pairwise[b, l, l] equals the
 * identity (HmvitFusionDesc::self_identity); bit 1: the upper-left 2 x 2 block M of EVERY pairwise[b, i, j] satisfies
 * |M^T M - I| <= 0.02 elementwise (HmvitFusionDesc::rigid_patch). */
int hmvit_pack_small(const void* mode, int mode_dtype, int n_mode, const void* record_len, int rl_dtype, int n_rl, const void* mask,
                     int mask_dtype, int n_mask, const void* pairwise, int pw_dtype, int B, int L, int64_t* out, void* stream);

/* (n_agents, C, P) f32 -> (n_agents, P, C) f32 and back: the layout change between the
 * reference's NCHW maps and the token-major residual stream used internally. */
int hmvit_nchw_to_tokens(const float* x, float* y, int n_agents, int C, int P, void* stream);
int hmvit_tokens_to_nchw(const float* x, float* y, int n_agents, int C, int P, void* stream);

/* HeteroLayerNorm (base_transformer.py:138-177) on token-major maps.
 * x (n_agents, P, C) f32; types: host array (n_agents); gamma/beta (T, C) f32 device;
 * y (n_agents, P, C) f32 or f16 by `precision`. */
int hmvit_layernorm(const float* x, void* y, const int32_t* types, const float* gamma,
                    const float* beta, int n_agents, int P, int C, int precision, void* stream);

/* y = act(a @ w^T + bias) (+ residual): nn.Linear as used by to_qkv / to_out /
 * HeteroFeedForward (hetero_fusion.py:111-152, base_transformer.py:180-192).
 * a (M, K), w (N, K) in the precision's element type; bias (N) f32 or NULL;
 * residual (M, N) f32 or NULL; gelu: 0/1 (exact erf form);
 * out_f32: 1 -> y is f32, 0 -> y has the precision's element type. */
int hmvit_linear(const void* a, const void* w, const float* bias, const float* residual,
                 void* y, int M, int N, int K, int gelu, int out_f32, int precision,
                 void* stream);

/* get_discretized_transformation_matrix + get_transformation_matrix + inverse
 * (torch_transformation_utils.py:108-134, 254-297, 349): pairwise_t (n, 4, 4) f32 device ->
 * ainv (n, 8) f32 device: [a00 a01 a02 a10 a11 a12 is_identity 0], the pixel-space map
 * src = Ainv [u, v, 1] used for sampling. */
int hmvit_pair_affines(const float* pairwise_t, float* ainv, int n, int H, int W,
                       float discrete_ratio, float downsample_rate, void* stream);

/* warp_affine (bilinear, zeros, align_corners=True) + get_roi_and_cav_mask (nearest)
 * (torch_transformation_utils.py:11-105, 317-355) of token-major maps, with the sampling code
 * the attention kernel uses.  src (n, P, C) f32, ainv (n, 8) -> dst (n, P, C) f32,
 * roi (n, P) f32. */
int hmvit_warp_affine(const float* src, const float* ainv, float* dst, float* roi, int n,
                      int H, int W, int C, void* stream);

/* One call of HeteroAttention.forward for every ego (hetero_fusion.py:187-277) minus the
 * output projection, fused with the L^2 warps of warp_features (:338-361) and the window /
 * grid partition (:384-394, 427-434).
 *   q      (B, L, P, C)          projected queries (scale folded), no bias
 *   kv     (B, L, E, 2, P, C)    projected, relation-folded keys / values, no bias
 *   ego_e  host (B, L)           which of the E variants ego (b, i) uses
 *   b_q    (T, C) f32, b_kv (T_ego, T_src, 2C) f32, mode host (B, L)
 *   ainv   (B, L_src, L_ego, 8)  hmvit_pair_affines(pairwise_t): record [b, j, i] samples source j
 *                                 in ego i's frame (pairwise_t[:, :, i] of hetero_fusion.py:345)
 *   out    (B, L, P, C)          attention output of ego i (rows >= n_ego untouched) */
int hmvit_window_attention(const void* q, const void* kv, const float* b_q, const float* b_kv,
                           const float* bias_frag, const float* ainv, const int32_t* mode,
                           const int32_t* cav_mask, const int32_t* ego_e, void* out, int B, int L,
                           int n_ego, int n_src, int E, int C, int H, int W, int window,
                           int partition, int precision, int skip_masked, void* stream);

/* ---- LiDAR BEV encoder (PointPillar branch) ---- */

/* PillarVFE with one PFN layer (use_norm, use_absolute_xyz, no distance) fused with
 * PointPillarScatter (sub_modules/pillar_vfe.py:31-53,105-146, point_pillar_scatter.py:14-47).
 *   voxels (Nv, 32, 4) f32, coords (Nv, 4) int32 [agent, z, y, x], num_points (Nv) int32;
 *   w (64, 10) f32 = linear.weight * bn_scale[:, None], shift (64) f32 = bn_bias - bn_mean * bn_scale
 *   (eval-mode BatchNorm1d, eps 1e-3);
 *   canvas (n_agents, ny, nx, 64) NHWC in the precision's element type, ZERO-FILLED by the caller
 *   (index z + y * nx + x), may be NULL; pillar_out (Nv, 64) f32, may be NULL;
 *   voxel_size / lidar_range: host arrays of 3 / 6 floats.  The voxel layout is fixed at 32 points x 4 features
 *   (max_points_per_voxel of the shipped yaml).  A pillar whose agent index is not in [0, n_agents), whose y / x lies
 *   outside the grid or whose z is not 0 is dropped (the reference's indexed scatter raises there) and counted in the
 *   device counter oob_count (may be NULL).
 *   A y_absmax slot announced with hmvit_conv_range(NULL, 0, slot) before this call receives atomicMax(|v|) over the values
 *   scattered into the canvas (and is consumed): the range the first HMVIT_PREC_SPLIT convolution needs, without a pass over
 *   the canvas. */
int hmvit_pfn_scatter(const float* voxels, const int32_t* coords, const int32_t* num_points, const float* w,
                      const float* shift, void* canvas, float* pillar_out, int n_pillars, int nx, int ny, int n_agents,
                      int32_t* oob_count, const float* voxel_size, const float* lidar_range, int precision, void* stream);

/* nn.Conv2d (square kernel) or nn.ConvTranspose2d (kernel = stride = deconv_stride) + bias + ReLU on
 * NHWC maps as an MFMA implicit GEMM: the layers of BaseBEVBackbone with eval-mode BatchNorm2d
 * folded into weight and bias (backbones/base_bev_backbone.py:36-87) and of DownsampleConv
 * (sub_modules/downsample_conv.py:20-51).
 *   x (N, H, W, Cin); w: conv (Cout, k*k*Cin) with k-index (ky*k + kx)*Cin + ci,
 *   deconv (s*s*Cout, Cin) with row (dy*s + dx)*Cout + co; bias (Cout) f32 or NULL;
 *   y NHWC with y_ctot channels per pixel, this layer writes channels [y_coff, y_coff + Cout)
 *   (the backbone's torch.cat is free); out_f32 = 1 stores f32 regardless of the precision.
 *   Cin must be a multiple of 64 (f16 mode) / 32 (f32 mode).  For deconv pass ksize 1, stride 1, pad 0. */
int hmvit_conv2d(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                 int ksize, int stride, int pad, int relu, int y_ctot, int y_coff, int deconv_stride, int out_f32,
                 int precision, void* stream);

/* Range information for the NEXT hmvit_conv2d / _ex / _rowpack call of the calling thread in HMVIT_PREC_SPLIT (consumed by it).
 * The split convolutions scale both operands into f16's comfortable range by powers of two taken from max |x| and max |w|:
 *   x_absmax  device pointer to the f32 bit pattern of max |x| (any upper bound works), e.g. the y_absmax slot of the
 *             convolution that produced x, or a slot filled by hmvit_absmax; NULL: the library measures it (an extra pass);
 *   w_absmax  > 0: max |w| as a host value (known when the weights are prepared); < 0: the weights were pre-multiplied by the
 *             power of two -w_absmax when they were prepared (nothing left to do while staging them); 0: measured by the library;
 *   y_absmax  device slot, zeroed by the caller, that receives atomicMax(|y|) over everything the call stores, or NULL.
 * hmvit_absmax: atomicMax(max |x|) into a zeroed slot. */
int hmvit_conv_range(const void* x_absmax, float w_absmax, void* y_absmax);
int hmvit_absmax(const float* x, size_t n, void* slot, void* stream);

/* The weights of a 3 x 3 / pad 1 convolution (stride 1 or 2) as the image its LDS ring holds (HMVIT_PREC_SPLIT: of the PRE-SCALED f32
 * weights, i.e. the ones passed with w_absmax < 0; HMVIT_PREC_F16: of the f16 weights).  The convolution kernels then copy weight
 * slabs global -> LDS by DMA, two taps ahead, instead of staging (and, in split mode, splitting) them through registers in every
 * workgroup: the Conv2d layers of base_bev_backbone.py:6-122, downsample_conv.py:32-51, resnet_ms.py / torchvision BasicBlock.
 *   hmvit_conv3x3_image_bytes  size of the image (0: shape / precision without one: Cin % 32 (split) or % 64 (f16) != 0);
 *   hmvit_conv3x3_image        w (Cout, 9 Cin) in hmvit_conv2d's layout -> image (device, that many bytes), once per weight version;
 *   hmvit_conv_gemm_image(_bytes)  the same for every other geometry in HMVIT_PREC_SPLIT (strided, 1 x 1, transposed): w is the
 *                              (Ncols, Ktot) pre-scaled f32 matrix hmvit_conv2d takes (Ncols = Cout, or stride^2 Cout for a transposed
 *                              convolution; Ktot = k k Cin, a multiple of 32), the slabs follow its own column order;
 *   hmvit_conv_weight_image    hands an image to the NEXT hmvit_conv2d / _ex call of the calling thread (consumed by it, like
 *                              hmvit_conv_range); kind 0: hmvit_conv3x3_image, 1: hmvit_conv_gemm_image.  `w` is still passed and is
 *                              what the call reads when the kernel it selects does not use that kind of image.
 * Results are bit-identical with and without an image. */
size_t hmvit_conv3x3_image_bytes(int Cout, int Cin, int precision);
int hmvit_conv3x3_image(const void* w, int Cout, int Cin, int precision, void* image, void* stream);
size_t hmvit_conv_gemm_image_bytes(int Ncols, int Ktot);
int hmvit_conv_gemm_image(const float* w, int Ncols, int Ktot, void* image, void* stream);
int hmvit_conv_weight_image(const void* image, int kind);

/* hmvit_conv2d with a residual operand and an up-sampled input (camera branch):
 *   residual (N, Ho, Wo, Cout) in the precision's element type or NULL: y = act(conv(x) + bias + residual), the tail of a
 *   torchvision BasicBlock / Bottleneck (`out += identity; out = relu(out)`, used by resnet_ms.py:27-38 and
 *   cvt_modules.py:13,303-305);
 *   upsample2 = 1: x is (N, H/2, W/2, Cin) and stands for its nearest-neighbour x2 upsampling (N, H, W, Cin)
 *   (NaiveDecoder.upsample, naive_decoder.py:56-61), which is never materialised.  Bit 1 of `upsample2` (value 2) selects the
 *   generic implicit-GEMM kernel where the library would take the patch-in-LDS 3 x 3 kernel (for A/B checks); bit 2 (value 4)
 *   takes the stride-2 ring kernel wherever it applies (by default only for Cin >= 256, where it is the faster one). */
int hmvit_conv2d_ex(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W, int Cin,
                    int Cout, int ksize, int stride, int pad, int relu, int upsample2, int out_f32, int precision, void* stream);

/* Few-channel stem convolution (the 7x7 / stride-2 / 3-channel conv1 of a ResNet, resnet_ms.py:27-38 through torchvision):
 * x is a zero-padded (N, Hp, Wp, 4) NHWC map in the precision's element type whose padding already contains the
 * convolution's border (output pixel (oy, ox) reads rows oy*stride .. oy*stride + krows - 1 and pixels ox*stride ..
 * ox*stride + 7); w is (Cout, krows * 32) with k = ky * 32 + px * 4 + ci (zero where px, ky or ci exceed the kernel);
 * y (N, Ho, Wo, Cout) in the precision's element type.  One 64-byte run per kernel row instead of one zero-padded K slab
 * per tap. */
int hmvit_conv2d_rowpack(const void* x, const void* w, const float* bias, void* y, int N, int Hp, int Wp, int Ho, int Wo, int Cout,
                         int krows, int stride, int relu, int precision, void* stream);

/* nn.MaxPool2d on NHWC maps (C a multiple of 8): the 3x3 / stride 2 / pad 1 pooling of the ResNet stem. */
int hmvit_maxpool2d(const void* x, void* y, int N, int H, int W, int C, int ksize, int stride, int pad, int precision, void* stream);

/* ---- detection post-processing ---- */

/* VoxelPostprocessor.post_process up to the candidate list, for one agent's head outputs
 * (opencood/data_utils/post_processor/voxel_postprocessor.py:232-330, delta_to_boxes3d :355-396;
 * box_utils.py boxes_to_corners_3d :139-184, project_box3d :258-296, remove_large_pred_bbx :722-751,
 * remove_bbx_abnormal_z :754-772).
 *   psm (A, H, W) f32 logits, rm (7A, H, W) f32, anchors (H, W, A, 7) f32 [x y z h w l yaw] (order_hwl = 1) or
 *   [x y z l h w yaw]; transform: (4, 4) row-major DEVICE matrix into the ego frame, or NULL (no_post_projection).
 *   Anchors with sigmoid(psm) > score_threshold that pass both sanity filters are appended (unordered) to
 *   corners (capacity, 8, 3), scores (capacity), index (capacity: anchor index (h W + w) A + a); *count (device)
 *   receives the number found (may exceed capacity: only `capacity` are stored). */
int hmvit_box_decode(const float* psm, const float* rm, const float* anchors, const float* transform, int H, int W, int A,
                     float score_threshold, int order_hwl, float* corners, float* scores, int32_t* index, int32_t* count,
                     int capacity, void* stream);

/* box_utils.nms_rotated (:575-620) + get_mask_for_boxes_within_range_torch (:326-357): the candidates are ranked by
 * (score descending, index ascending), the top 1000 go through greedy suppression at polygon IoU > iou_threshold
 * (IoU of the convex quadrilaterals of corners 0..3, x / y), and the survivors whose 8 corners all lie in
 * range_xy = host {x_lo, y_lo, x_hi, y_hi} are written to keep[] in pick order; *n_keep (device) = their number.
 * index may be NULL (ties then break by position). */
size_t hmvit_nms_workspace_bytes(int n);
int hmvit_nms_rotated(const float* corners, const float* scores, const int32_t* index, int n, float iou_threshold,
                      const float* range_xy, void* workspace, size_t workspace_bytes, int32_t* keep, int32_t* n_keep,
                      void* stream);

/* iou (na, nb) f32 of convex quadrilaterals: corner k of box i at a[i * stride_box + k * stride_pt + {0, 1}]
 * (common_utils.compute_iou :120-139 with shapely replaced by Sutherland-Hodgman clipping; eval_utils.py:144-196). */
int hmvit_quad_iou(const float* a, const float* b, int na, int nb, int stride_box, int stride_pt, float* iou, void* stream);

/* ---- pillariser ---- */

/* spconv.utils.Point2VoxelCPU3d.point_to_voxel as called by SpVoxelPreprocessor.preprocess
 * (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:34-57; spconv-cu113, third party, version unpinned), same
 * deterministic result as its sequential algorithm: voxels in order of first appearance, points in input order, at most
 * max_points per voxel and max_voxels voxels.
 *   points (n_points, 4) f32 device; voxel_size / lidar_range: host arrays of 3 / 6 floats (grid = round(extent / size));
 *   voxels (max_voxels, max_points, 4) f32 zero padded, coords (max_voxels, 3) int32 [z, y, x], num_points (max_voxels)
 *   int32, *n_voxels device int32 = voxels produced. */
size_t hmvit_voxelize_workspace_bytes(int n_points, int nx, int ny, int nz);
int hmvit_voxelize(const float* points, int n_points, const float* voxel_size, const float* lidar_range, int max_points,
                   int max_voxels, void* workspace, size_t workspace_bytes, float* voxels, int32_t* coords, int32_t* num_points,
                   int32_t* n_voxels, void* stream);

/* ---- camera -> BEV lift: the non-GEMM parts of CrossViewAttention (cvt_modules.py:95-280) ---- */

/* Camera-aware positional embeddings (cvt_modules.py:229-269), all f32, outputs token-major (n_agents * n_cam, H * W, dim):
 *   mode 0: normalize(img_embed(E_inv [I_inv [px, py, 1]; 1]) - cam_embed(E_inv[:, 3])) for every pixel of an H x W feature
 *           map (pixel plane of generate_grid scaled by image_w / image_h; square maps), w_in = img_embed.weight (dim, 4);
 *   mode 1: normalize(bev_embed(grid xy) - cam_embed(E_inv[:, 3])) + x, grid (>= 2, H * W) = BEVEmbedding.grid, w_in =
 *           bev_embed.weight (dim, 2), w_bias its bias, x (n_agents, dim, H * W) NCHW or NULL.
 * I_inv (n_agents * n_cam, 3, 3), E_inv (n_agents * n_cam, 4, 4), w_cam = cam_embed.weight (dim, 4). */
int hmvit_cvt_embed(int mode, const float* I_inv, const float* E_inv, const float* grid, const float* w_in, const float* w_bias,
                    const float* w_cam, const float* x, float* out, int n_agents, int n_cam, int H, int W, int dim,
                    float image_w, float image_h, void* stream);

/* y (n, P, C) = relu(x (n, C, P) * scale[c] + shift[c]): eval-mode BatchNorm2d + ReLU in front of the 1x1 convolutions of
 * feature_linear / feature_proj (cvt_modules.py:197-207), with the NCHW -> token-major transpose. */
int hmvit_bn_relu_tokens(const float* x, const float* scale, const float* shift, float* y, int n, int C, int P, void* stream);

/* CrossAttention core (cvt_modules.py:148-158): q (n_agents, n_cam, Q, heads * 32), k (n_agents, n_cam, K, heads * 32),
 * v (n_agents, n_cam * K, heads * 32) in the precision's element type -> out (n_agents, Q, heads * 32) f32; logits
 * q_cam . k_cam / sqrt(32), one softmax over the keys of all cameras.  HMVIT_PREC_F16 runs on the matrix cores and needs
 * Q and K to be multiples of 64.
 * PRECONDITION of HMVIT_PREC_SPLIT (ABI >= 11): the f32 operands are taken at their own scale and split into f16 (hi, lo)
 * halves, so |q| / sqrt(dim_head), |k| and |v| must stay below 65504 (f16's largest finite value; comfortably below for
 * full accuracy of the low halves: 1e-3 < typical magnitude < 1e4).  The library cannot see the operands' range without a
 * pass over them: callers that cannot bound it (hm-vit_amd/cvt.py bounds it from the LayerNorm output and the projection's
 * largest row L1 norm) must pass HMVIT_PREC_F32, which is exact at any scale. */
int hmvit_cross_attention(const void* q, const void* k, const void* v, float* out, int n_agents, int n_cam, int Q, int K,
                          int heads, int dim_head, int precision, void* stream);

/* ---- training-mode operators of the camera lift (cvt_modules.py:95-165 and resnet_ms.py under autograd) ----
 * hmvit_cross_attention_train: hmvit_cross_attention in exact f32 that also returns lse (n_agents, heads, Q), the log-sum-exp of
 *   every query row over the keys of all cameras (logits q . k / sqrt(dim_head)).
 * hmvit_cross_attention_backward: d_out (n_agents, Q, HD) -> dq (n_agents, n_cam, Q, HD), dk (n_agents, n_cam, K, HD),
 *   dv (n_agents, n_cam K, HD); the probabilities are rebuilt from lse; every output element is written (no accumulation).
 * hmvit_layernorm_backward: nn.LayerNorm(C), C in {64, 128, 256}: dx (M, C) written, dgamma / dbeta (C) ACCUMULATED (zero-fill).
 * hmvit_gelu / hmvit_gelu_backward: erf GELU and dx = dy gelu'(pre), elementwise f32. */
int hmvit_cross_attention_train(const float* q, const float* k, const float* v, float* out, float* lse, int n_agents, int n_cam, int Q,
                                int K, int heads, int dim_head, void* stream);
int hmvit_cross_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* d_out,
                                   float* dq, float* dk, float* dv, int n_agents, int n_cam, int Q, int K, int heads, int dim_head,
                                   void* stream);
int hmvit_layernorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int C,
                             void* stream);
int hmvit_gelu(const float* pre, float* y, size_t n, void* stream);
int hmvit_gelu_backward(const float* pre, const float* dy, float* dx, size_t n, void* stream);

/* Softmax attention with an additive logit bias, f32: q (batch, Q, heads * 32), k / v (batch, K, heads * 32),
 * bias (heads, Q, K) -> out (batch, Q, heads * 32); logits q . k / sqrt(32) + bias.  The self-attention that closes FAXModule
 * (fax_modules.py:96-180, relative-position bias over the whole BEV map). */
int hmvit_attention_bias(const float* q, const float* k, const float* v, const float* bias, float* out, int batch, int Q, int K,
                         int heads, int dim_head, void* stream);
/* The same under autograd (training of the FAX camera lift, fax_modules.py:136-180): _train also returns lse (batch, heads, Q), the
 * log-sum-exp of every query row; _backward rebuilds the probabilities from it and writes dq / dk / dv (shapes of q / k / v) and
 * d_bias (heads, Q, K) = the logit gradient summed over the batch (every element written, no atomics). */
int hmvit_attention_bias_train(const float* q, const float* k, const float* v, const float* bias, float* out, float* lse, int batch,
                               int Q, int K, int heads, int dim_head, void* stream);
int hmvit_attention_bias_backward(const float* q, const float* k, const float* v, const float* bias, const float* out, const float* lse,
                                  const float* d_out, float* dq, float* dk, float* dv, float* d_bias, int batch, int Q, int K, int heads,
                                  int dim_head, void* stream);
/* Adjoint of hmvit_maxpool2d on f32 NHWC maps (the ResNet stem's pooling under autograd, resnet_ms.py:71): dx (N, H, W, C) written;
 * the gradient of a window goes to its first maximum in row-major order, as torch's does. */
int hmvit_maxpool2d_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int ksize, int stride, int pad,
                             void* stream);

/* debug: lane mapping of ds_read_b64_tr_b16 (used once to pin the V-operand layout) */
int hmvit_debug_tr16(uint16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HMVIT_H */
