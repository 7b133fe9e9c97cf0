// Training entry points of libhmvit: HeteroFusion forward that keeps its activations, and the backward pass
// (bevformer_point_pillar_hetero.py:39-49 under torch.autograd in the reference's train loop, train_camera.py:163-199).
//
// Exact-f32 arithmetic (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), sequential block, apply_head = 1.  The forward runs the
// f32 pipeline of capi.hip stage by stage into per-stage buffers of the caller's `saved` area; Dropout
// (hetero_fusion.py:65-66, base_transformer.py:186-192) is a pure function of (seed, salt, element) and is regenerated
// by the backward pass.  HeteroFusion's last stage computes ego 0 only (its other rows never reach the loss), exactly
// as the inference path does; the K'/V' of all sources still receive gradients there.
#include <string.h>

#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

#define HMVIT_TRY(expr)                  \
    do {                                 \
        int _rc = (expr);                \
        if (_rc != HMVIT_OK) return _rc; \
    } while (0)

namespace {

struct StageInfo {
    bool last;
    int n_ego, E;
    int e_of_type[HMVIT_NUM_TYPES], e_type[HMVIT_NUM_TYPES];
};

struct TrainPlan {
    int B, L, C, H, W, P, mlp, heads, n_slots, max_cav, E_max, n_stages;
    int only_stage;      // -1: the whole HeteroFusion; 0 / 1: one stage of the block on its own (HmvitFusionTrainDesc::only_stage)
    size_t A;            // floats of one (n_slots, P, C) activation
    size_t stage_floats; // floats saved per stage
    // offsets (floats) inside one stage's record
    // (round 3: the normalised rows and the FFN activations are no longer kept - the backward recomputes them from x / x' / pre)
    size_t o_x, o_q, o_kv, o_o, o_lse, o_x1, o_pre;
    // offsets (floats) of the tail of the saved area
    size_t o_xfin, o_hpre, o_hh, o_ainv, total_floats;
    // x16 images of the forward's weight matrices (k_linear16; C = mlp = 256 only): same layout as the backward's transposed
    // weights (WeightLayout), one float of storage per (hi, lo) pair; o_winv: one inverse scale per matrix
    size_t o_img, o_winv;
    bool lin16;
    bool save_pre, save_q;   // the FFN pre-activations / the queries are kept (HmvitFusionTrainDesc::recompute bits 0 / 1 clear)
};

// (C, C) blocks of one set of weights: per stage q (T), kv (T, T, 2), o (T), w_1 (T), w_2 (T); then the head's two (T each)
struct WeightLayout {
    size_t q, kv, o, w1, w2, stage, h1, h2, total;
};
static WeightLayout weight_layout(int C, int mlp) {
    const size_t T = HMVIT_NUM_TYPES;
    WeightLayout wl;
    size_t w = 0;
    wl.q = w; w += T * C * C;
    wl.kv = w; w += T * T * 2 * C * C;
    wl.o = w; w += T * C * C;
    wl.w1 = w; w += T * C * mlp;
    wl.w2 = w; w += T * mlp * C;
    wl.stage = w;
    wl.h1 = 2 * w; wl.h2 = 2 * w + T * C * C;
    wl.total = 2 * w + 2 * T * C * C;
    return wl;
}

// which image goes with a weight pointer: the image areas mirror the weight arrays element for element
struct ImgRegistry {
    struct Entry { const float* w; size_t n_floats; const half_t* img; const float* inv; };
    Entry e[16];
    int n = 0;
    void add(const float* w, size_t n_mat, const float* img_as_float, const float* inv) {
        if (w && n < 16) e[n++] = Entry{w, n_mat * 65536, reinterpret_cast<const half_t*>(img_as_float), inv};
    }
    bool find(const float* w, const half_t*& img, const float*& inv) const {
        for (int i = 0; i < n; ++i)
            if (w >= e[i].w && w < e[i].w + e[i].n_floats && (size_t)(w - e[i].w) % 65536 == 0) {
                img = e[i].img + 2 * (size_t)(w - e[i].w);
                inv = e[i].inv + (size_t)(w - e[i].w) / 65536;
                return true;
            }
        return false;
    }
};

int make_train_plan(const HmvitFusionTrainDesc* t, TrainPlan& pl) {
    const HmvitFusionDesc* d = &t->fwd;
    HMVIT_TRY(check_desc(d));
    HMVIT_CHECK_ARG(d->precision == HMVIT_PREC_F32, "training runs in the exact-f32 mode (precision=%d)", d->precision);
    HMVIT_CHECK_ARG(!d->parallel, "training: architect_mode 'parallel' is not built (the shipped yaml is sequential)");
    // window 4 / 8 with dim_head 32: the tuned kernels; every other shape the reference accepts (hetero_fusion.py:285-327) trains
    // through the generic exact-f32 attention kernels (k_attention_any / k_attention_any_bwd)
    HMVIT_CHECK_ARG(d->window >= 1 && d->window <= 16 && d->dim_head >= 1 && d->dim_head <= 64 && d->C % d->dim_head == 0,
                    "training: window_size=%d / dim_head=%d (window 1 .. 16, dim_head 1 .. 64 dividing C)", d->window, d->dim_head);
    pl.only_stage = t->only_stage >= 1 && t->only_stage <= 2 ? t->only_stage - 1 : -1;
    HMVIT_CHECK_ARG(t->only_stage >= 0 && t->only_stage <= 2, "only_stage=%d (0 = whole fusion, 1 = window stage, 2 = grid stage)", t->only_stage);
    HMVIT_CHECK_ARG(d->apply_head == (pl.only_stage < 0 ? 1 : 0), "training: HeteroFusion (apply_head = 1), or one stage of the block (apply_head = 0)");
    HMVIT_CHECK_ARG(t->drop_p >= 0.f && t->drop_p < 1.f, "drop_p=%f out of [0, 1)", t->drop_p);
    // (the field took what was tail padding before ABI 12: an old caller's uninitialised bytes must not pick a layout)
    HMVIT_CHECK_ARG((t->recompute & ~3) == 0, "recompute=%d (bits 0 and 1 only)", t->recompute);
    pl.B = d->B; pl.L = d->L; pl.C = d->C; pl.H = d->H; pl.W = d->W; pl.P = d->H * d->W; pl.mlp = d->mlp_dim;
    pl.heads = d->heads; pl.n_slots = d->B * d->L; pl.n_stages = pl.only_stage < 0 ? 2 * d->num_iters : 1;
    HMVIT_CHECK_ARG(pl.n_slots <= kMaxSlots, "B*L=%d exceeds %d agent slots per call", pl.n_slots, kMaxSlots);
    pl.max_cav = 0;
    for (int b = 0; b < d->B; ++b) pl.max_cav = d->record_len[b] > pl.max_cav ? d->record_len[b] : pl.max_cav;
    bool seen[HMVIT_NUM_TYPES] = {false, false};
    for (int b = 0; b < d->B; ++b)
        for (int i = 0; i < pl.max_cav; ++i) seen[d->mode[b * d->L + i]] = true;
    pl.E_max = (int)seen[0] + (int)seen[1];
    const size_t tok = (size_t)pl.n_slots * pl.P;
    pl.A = tok * pl.C;
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off = (off + n + 63) / 64 * 64; return o; };
    pl.save_pre = (t->recompute & 1) == 0;
    pl.save_q = (t->recompute & 2) == 0;
    pl.o_x = carve(pl.A); pl.o_q = pl.save_q ? carve(pl.A) : 0;
    pl.o_kv = carve(tok * pl.E_max * 2 * pl.C); pl.o_o = carve(pl.A); pl.o_lse = carve(tok * pl.heads);
    pl.o_x1 = carve(pl.A); pl.o_pre = pl.save_pre ? carve(tok * pl.mlp) : 0;
    pl.stage_floats = off;
    off = pl.stage_floats * pl.n_stages;
    pl.o_xfin = carve(pl.A);
    pl.o_hpre = carve((size_t)pl.B * pl.P * pl.C);
    pl.o_hh = carve((size_t)pl.B * pl.P * pl.C);
    pl.o_ainv = carve((size_t)pl.n_slots * pl.L * 8);
    pl.lin16 = pl.C == 256 && pl.mlp == 256;
    const WeightLayout wl = weight_layout(pl.C, pl.mlp);
    pl.o_img = carve(pl.lin16 ? wl.total : 0);
    pl.o_winv = carve(pl.lin16 ? wl.total / 65536 : 0);
    pl.total_floats = off;
    return HMVIT_OK;
}

StageInfo stage_info(const HmvitFusionDesc* d, const TrainPlan& pl, int st) {
    StageInfo si;
    si.last = pl.only_stage < 0 && st == pl.n_stages - 1;      // only HeteroFusion's last stage is pruned to ego 0
    si.n_ego = si.last ? 1 : pl.max_cav;
    si.E = 0;
    for (int t = 0; t < HMVIT_NUM_TYPES; ++t) { si.e_of_type[t] = -1; si.e_type[t] = 0; }
    for (int b = 0; b < d->B; ++b)
        for (int i = 0; i < si.n_ego; ++i) {
            const int t = d->mode[b * d->L + i];
            if (si.e_of_type[t] < 0) { si.e_of_type[t] = si.E; si.e_type[si.E] = t; ++si.E; }
        }
    return si;
}

struct Jobs {
    GemmJobs jobs;
    LinJobs lin;                 // jobs whose weight has an x16 image (ImgRegistry): k_linear16
    const ImgRegistry* reg;
    bool a_f32, gelu, out_f32;
    hipStream_t st;
    Jobs(hipStream_t s, const ImgRegistry* r = nullptr) : reg(r), a_f32(false), gelu(false), out_f32(true), st(s) { jobs.n = 0; lin.n = 0; }
    int flush() {
        if (lin.n) {
            const int rc = launch_linear16(lin, st);
            lin.n = 0;
            if (rc != HMVIT_OK) return rc;
        }
        if (jobs.n == 0) return HMVIT_OK;
        // f32 operands and results; the products run as split f16 pairs (gemm.hip k_gemm_split: fp32-class accuracy, ~3x the
        // rate of the exact-f32 MFMA), -DHMVIT_TRAIN_EXACT_F32 restores the exact-f32 kernel
#ifdef HMVIT_TRAIN_EXACT_F32
        int rc = launch_gemm(jobs, a_f32, gelu, out_f32, HMVIT_PREC_F32, st);
#else
        int rc = launch_gemm(jobs, true, gelu, true, HMVIT_PREC_SPLIT, st);
#endif
        jobs.n = 0;
        return rc;
    }
    // y (M, N) = a (M, K) w^T (+ bias) (+ residual)
    // (ln_g, ln_b): y = LayerNorm(a) w^T ... - only where can_fuse_ln(w) said so (the x16 kernel normalises the rows in registers)
    bool can_fuse_ln(const float* w, int N, int K) const {
#ifndef HMVIT_TRAIN_EXACT_F32
        const half_t* img;
        const float* inv;
        return reg && N == 256 && K == 256 && reg->find(w, img, inv);
#else
        return false;
#endif
    }
    // drop (x16 path only - ask can_fuse_ln first): y = residual + Dropout(a w^T + bias)
    int add(const float* a, const float* w, const float* bias, const float* residual, float* y, int M, int N, int K,
            const float* ln_g = nullptr, const float* ln_b = nullptr, const DropCfg* drop = nullptr) {
#ifndef HMVIT_TRAIN_EXACT_F32
        const half_t* img;
        const float* inv;
        if (reg && N == 256 && K == 256 && reg->find(w, img, inv)) {
            // products of the same rows (Q / K' / V' of a slot) share one pass over them
            if (lin.n > 0 && !residual) {
                LinJob& q = lin.j[lin.n - 1];
                if (q.a == a && q.M == M && !q.residual && q.n_mat < kMaxLinMats && q.ln_gamma == ln_g) {
                    q.wimg[q.n_mat] = img; q.w_inv[q.n_mat] = inv; q.bias[q.n_mat] = bias; q.y[q.n_mat] = y;
                    ++q.n_mat;
                    return HMVIT_OK;
                }
            }
            if (lin.n == kMaxLinJobs) HMVIT_TRY(flush());
            LinJob& q = lin.j[lin.n++];
            memset(&q, 0, sizeof(q));
            q.a = a; q.wimg[0] = img; q.w_inv[0] = inv; q.bias[0] = bias; q.y[0] = y; q.residual = residual;
            q.M = M; q.n_mat = 1; q.ldy = N; q.ln_gamma = ln_g; q.ln_beta = ln_b;
            if (drop) q.drop = *drop;
            return HMVIT_OK;
        }
#endif
        if (ln_g || drop) { set_error("training: LayerNorm / dropout fusion asked of the generic GEMM%s", ""); return HMVIT_EINVAL; }
        GemmJob j;
        j.a = a; j.w = w; j.bias = bias; j.residual = residual; j.y = y;
        j.M = M; j.N = N; j.K = K; j.n_per_plane = N; j.plane_stride = 0;
        jobs.j[jobs.n++] = j;
        if (jobs.n == kMaxJobs) return flush();
        return HMVIT_OK;
    }
};

// y = sum_i a[i] w[i]^T (+ residual): one x16 job, the sum in registers.  true when every weight has an x16 image (else: nothing queued)
static bool add_sum16(Jobs& J, const float* const* a, const float* const* w, int n_in, const float* residual, float* y, int M, int* rc) {
    *rc = HMVIT_OK;
#ifndef HMVIT_TRAIN_EXACT_F32
    if (!J.reg || n_in < 1 || n_in > kMaxLinMats) return false;
    const half_t* img[kMaxLinMats];
    const float* inv[kMaxLinMats];
    for (int i = 0; i < n_in; ++i)
        if (!J.reg->find(w[i], img[i], inv[i])) return false;
    if (J.lin.n == kMaxLinJobs) { *rc = J.flush(); if (*rc != HMVIT_OK) return true; }
    LinJob& q = J.lin.j[J.lin.n++];
    memset(&q, 0, sizeof(q));
    q.a = a[0]; q.residual = residual; q.y[0] = y; q.M = M; q.n_mat = n_in; q.ldy = 256; q.sum_inputs = 1;
    for (int i = 0; i < n_in; ++i) {
        q.wimg[i] = img[i]; q.w_inv[i] = inv[i];
        if (i > 0) q.a_more[i - 1] = a[i];
    }
    return true;
#else
    return false;
#endif
}

struct TnJobs {
    GemmTnJobs jobs;
    hipStream_t st;
    TnJobs(hipStream_t s) : st(s) { jobs.n = 0; }
    int flush() {
        if (jobs.n == 0) return HMVIT_OK;
        int rc = launch_gemm_tn(jobs, st);
        jobs.n = 0;
        return rc;
    }
    // dw (N, K) += dy^T a;  dbias (N) += colsum(dy)
    int add(const float* dy, const float* a, float* dw, float* dbias, int M, int N, int K) {
        GemmTnJob j;
        j.dy = dy; j.a = a; j.dw = dw; j.dbias = dbias; j.M = M; j.N = N; j.K = K; j.ld_dy = N; j.ld_a = K;
        jobs.j[jobs.n++] = j;
        // jobs of one launch may target the same dw (atomics), so no ordering constraint
        if (jobs.n == kMaxJobs) return flush();
        return HMVIT_OK;
    }
};

void fill_attn(const HmvitFusionDesc* d, const TrainPlan& pl, const StageInfo& si, int s, const float* q, const float* kv,
               float* o, float* lse, const float* ainv, AttnParams& ap) {
    const HmvitStageWeights& wt = d->stage[s];
    memset(&ap, 0, sizeof(ap));
    ap.q = q; ap.kv = kv; ap.b_q = wt.b_q; ap.b_kv = wt.b_kv; ap.bias_frag = wt.bias_frag;
    ap.ainv = ainv; ap.out = o; ap.lse = lse;
    ap.B = d->B; ap.L = d->L; ap.n_ego = si.n_ego; ap.n_src = pl.max_cav; ap.E = si.E; ap.C = d->C; ap.H = d->H; ap.W = d->W;
    ap.window = d->window; ap.dim_head = d->dim_head; ap.partition = s == 0 ? HMVIT_PART_WINDOW : HMVIT_PART_GRID;
    ap.skip_masked = d->skip_masked;
    for (int i = 0; i < pl.n_slots; ++i) {
        ap.mode[i] = (int8_t)d->mode[i];
        ap.cav[i] = (int8_t)(d->cav_mask[i] != 0);
        ap.ego_e[i] = (int8_t)(si.e_of_type[d->mode[i]] < 0 ? 0 : si.e_of_type[d->mode[i]]);
    }
}

// LayerNorm of slots [0, n) of every sample (contiguous per sample)
int ln_slots(const float* x, float* y, const float* g, const float* be, const HmvitFusionDesc* d, const TrainPlan& pl, int n,
             hipStream_t st, int only_b = -1) {
    const size_t me = (size_t)pl.P * pl.C;
    for (int b = 0; b < d->B; ++b) {
        if (only_b >= 0 && b != only_b) continue;
        AgentTypes ty;
        memset(&ty, 0, sizeof(ty));
        for (int i = 0; i < n; ++i) ty.t[i] = (int8_t)d->mode[b * d->L + i];
        HMVIT_TRY(launch_layernorm(x + (size_t)b * d->L * me, y + (size_t)b * d->L * me, g, be, ty, n, pl.P, pl.C,
                                   HMVIT_PREC_F32, st));
    }
    return HMVIT_OK;
}

int ln_bwd_slots(const float* x, const float* dy, const float* g, const float* dres, float* dx, float* dg, float* db,
                 const HmvitFusionDesc* d, const TrainPlan& pl, int n, hipStream_t st) {
    const size_t me = (size_t)pl.P * pl.C;
    for (int b = 0; b < d->B; ++b) {
        AgentTypes ty;
        memset(&ty, 0, sizeof(ty));
        for (int i = 0; i < n; ++i) ty.t[i] = (int8_t)d->mode[b * d->L + i];
        const size_t o = (size_t)b * d->L * me;
        HMVIT_TRY(launch_layernorm_bwd(x + o, dy + o, g, ty, n, dres ? dres + o : nullptr, dx + o, dg, db, pl.P, pl.C, st));
    }
    return HMVIT_OK;
}

DropCfg drop_cfg(const HmvitFusionTrainDesc* t, int st, int which) {
    DropCfg c;
    c.seed = t->seed; c.salt = (unsigned)(st * 4 + which); c.p = t->drop_p;
    return c;
}

}  // namespace

static int train_forward(const HmvitFusionTrainDesc* t, hipStream_t st) {
    const HmvitFusionDesc* d = &t->fwd;
    TrainPlan pl;
    HMVIT_TRY(make_train_plan(t, pl));
    HMVIT_CHECK_ARG(d->x && d->pairwise_t && d->out && t->saved, "x / pairwise_t / out / saved is null");
    if (t->saved_bytes < pl.total_floats * 4) {
        set_error("saved area too small: %zu < %zu bytes", t->saved_bytes, pl.total_floats * 4);
        return HMVIT_ENOMEM;
    }
    const size_t scratch1 = pl.A > (size_t)pl.n_slots * pl.P * pl.mlp ? pl.A : (size_t)pl.n_slots * pl.P * pl.mlp;     // floats
    // recompute: the queries (until the attention has run), then the pre-activations, live in the second half
    const size_t scratch = scratch1 * 4 * (pl.save_pre && pl.save_q ? 1 : 2);
    if (!d->workspace || d->workspace_bytes < scratch) {
        set_error("workspace too small: %zu < %zu bytes", d->workspace_bytes, scratch);
        return HMVIT_ENOMEM;
    }
    HMVIT_CHECK_ARG(pl.only_stage >= 0 || (d->head_w1 && d->head_b1 && d->head_w2 && d->head_b2), "mlp_head weights are null");
    const int B = d->B, L = d->L, C = d->C, P = pl.P, mlp = pl.mlp;
    const size_t me = (size_t)P * C;
    float* S = reinterpret_cast<float*>(t->saved);
    float* tmp = reinterpret_cast<float*>(d->workspace);
    float* ainv = S + pl.o_ainv;

    HMVIT_TRY(launch_transpose(d->x, S + pl.o_x, pl.n_slots, C, P, st));
    HMVIT_TRY(launch_pair_affines(d->pairwise_t, ainv, pl.n_slots * L, d->H, d->W, d->discrete_ratio, d->downsample_rate, st));

    // x16 images of this step's weights (they change every step): k_linear16 serves every Linear of the forward
    ImgRegistry reg;
    const ImgRegistry* regp = nullptr;
    if (pl.lin16) {
        const WeightLayout wl = weight_layout(C, mlp);
        const int T = HMVIT_NUM_TYPES;
        float* IMG = S + pl.o_img;
        float* INV = S + pl.o_winv;
        auto images = [&](const void* w, size_t off, int n_mat) -> int {
            if (!w) return HMVIT_OK;
            const float* wf = reinterpret_cast<const float*>(w);
            reg.add(wf, n_mat, IMG + off, INV + off / 65536);
            return launch_weight_images16(wf, reinterpret_cast<half_t*>(IMG + off), INV + off / 65536, n_mat, st);
        };
        for (int s = 0; s < 2; ++s) {
            if (pl.only_stage >= 0 && s != pl.only_stage) continue;
            const HmvitStageWeights& wt = d->stage[s];
            const size_t base = (size_t)s * wl.stage;
            HMVIT_TRY(images(wt.w_q, base + wl.q, T));
            HMVIT_TRY(images(wt.w_kv, base + wl.kv, T * T * 2));
            HMVIT_TRY(images(wt.w_o, base + wl.o, T));
            HMVIT_TRY(images(wt.w_1, base + wl.w1, T));
            HMVIT_TRY(images(wt.w_2, base + wl.w2, T));
        }
        if (pl.only_stage < 0) {
            HMVIT_TRY(images(d->head_w1, wl.h1, T));
            HMVIT_TRY(images(d->head_w2, wl.h2, T));
        }
        regp = &reg;
    }

    for (int sidx = 0; sidx < pl.n_stages; ++sidx) {
        const int s = pl.only_stage < 0 ? (sidx & 1) : pl.only_stage;
        const HmvitStageWeights& wt = d->stage[s];
        const StageInfo si = stage_info(d, pl, sidx);
        float* R = S + (size_t)sidx * pl.stage_floats;
        float* x_in = R + pl.o_x;
        float* x_out = (sidx + 1 < pl.n_stages) ? S + (size_t)(sidx + 1) * pl.stage_floats + pl.o_x : S + pl.o_xfin;
        float *kv = R + pl.o_kv, *o = R + pl.o_o, *lse = R + pl.o_lse, *x1 = R + pl.o_x1;
        float* q = pl.save_q ? R + pl.o_q : tmp + scratch1;
        float* pre = pl.save_pre ? R + pl.o_pre : tmp + scratch1;
        // transient rows live in the one scratch buffer, each dead before the next is written (stream order): LN(x), the
        // out-projection, LN(x'), the FFN activations
        float *xn = tmp, *xn2 = tmp, *h = tmp;
        const int n_ego = si.n_ego;

        {
            Jobs jb(st, regp);
            // with the x16 Linear the LayerNorm happens on the rows' way into the products (the normalised rows are not written)
            const bool fuse = jb.can_fuse_ln(reinterpret_cast<const float*>(wt.w_q), C, C);
            if (!fuse) HMVIT_TRY(ln_slots(x_in, xn, wt.ln_gamma, wt.ln_beta, d, pl, pl.max_cav, st));
            const float* rows = fuse ? x_in : xn;
            for (int b = 0; b < B; ++b)
                for (int l = 0; l < pl.max_cav; ++l) {
                    const int slot = b * L + l, ty = d->mode[slot];
                    const float* lg = fuse ? wt.ln_gamma + ty * C : nullptr;
                    const float* lb = fuse ? wt.ln_beta + ty * C : nullptr;
                    if (l < n_ego)
                        HMVIT_TRY(jb.add(rows + slot * me, reinterpret_cast<const float*>(wt.w_q) + (size_t)ty * C * C, nullptr, nullptr,
                                         q + slot * me, P, C, C, lg, lb));
                    for (int e = 0; e < si.E; ++e) {
                        const float* w = reinterpret_cast<const float*>(wt.w_kv) + (size_t)(si.e_type[e] * HMVIT_NUM_TYPES + ty) * 2 * C * C;
                        float* y = kv + (size_t)(slot * si.E + e) * 2 * me;
                        HMVIT_TRY(jb.add(rows + slot * me, w, nullptr, nullptr, y, P, C, C, lg, lb));                       // K'
                        HMVIT_TRY(jb.add(rows + slot * me, w + (size_t)C * C, nullptr, nullptr, y + me, P, C, C, lg, lb));  // V'
                    }
                }
            HMVIT_TRY(jb.flush());
        }
        {
            AttnParams ap;
            fill_attn(d, pl, si, s, q, kv, o, lse, ainv, ap);
            // f32 planes either way; with identity self transforms (descriptor flag, window 8, C >= 128) the persistent
            // split-operand kernel (fp32-class accuracy, ~3x faster than the one-window-per-workgroup exact-f32 kernel)
            ap.self_identity = d->self_identity;
#ifdef HMVIT_TRAIN_EXACT_F32
            HMVIT_TRY(launch_attention(ap, HMVIT_PREC_F32, st));
#else
            const bool generic = (d->window != 4 && d->window != 8) || d->dim_head != 32;      // generic shapes: the exact-f32 kernel
            HMVIT_TRY(launch_attention(ap, generic ? HMVIT_PREC_F32 : HMVIT_PREC_SPLIT, st));
#endif
        }
        // x' = x + Dropout(a_linears(O)) on the ego slots: in the Linear's epilogue where the x16 kernel runs it (round 5), else a pass of
        // its own.  Element index of the dropout stream = offset inside the slot's (P, C) map, one seed per slot
        for (int b = 0; b < B; ++b) {
            Jobs jb(st, regp);
            const bool fused = jb.can_fuse_ln(reinterpret_cast<const float*>(wt.w_o), C, C);
            const DropCfg dc = drop_cfg(t, sidx, 0);
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                DropCfg di = dc;
                di.seed = dc.seed + 0x51ED270B1ull * (unsigned long long)(slot + 1);
                if (fused)
                    HMVIT_TRY(jb.add(o + slot * me, reinterpret_cast<const float*>(wt.w_o) + (size_t)ty * C * C, wt.b_o + ty * C, x_in + slot * me,
                                     x1 + slot * me, P, C, C, nullptr, nullptr, &di));
                else
                    HMVIT_TRY(jb.add(o + slot * me, reinterpret_cast<const float*>(wt.w_o) + (size_t)ty * C * C, wt.b_o + ty * C, nullptr,
                                     tmp + slot * me, P, C, C));
            }
            HMVIT_TRY(jb.flush());
            if (!fused)
                for (int i = 0; i < n_ego; ++i) {
                    const size_t off = (size_t)(b * L + i) * me;
                    DropCfg di = dc;
                    di.seed = dc.seed + 0x51ED270B1ull * (unsigned long long)(b * L + i + 1);
                    HMVIT_TRY(launch_add_drop(x_in + off, tmp + off, x1 + off, me, di, st));
                }
        }
        for (int b = 0; b < B; ++b) {
            Jobs j1(st, regp);
            const bool fuse = j1.can_fuse_ln(reinterpret_cast<const float*>(wt.w_1), mlp, C);
            if (!fuse) HMVIT_TRY(ln_slots(x1, xn2, wt.ffn_ln_gamma, wt.ffn_ln_beta, d, pl, n_ego, st, b));
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                HMVIT_TRY(j1.add((fuse ? x1 : xn2) + slot * me, reinterpret_cast<const float*>(wt.w_1) + (size_t)ty * mlp * C, wt.b_1 + ty * mlp,
                                 nullptr, pre + (size_t)slot * P * mlp, P, mlp, C, fuse ? wt.ffn_ln_gamma + ty * C : nullptr,
                                 fuse ? wt.ffn_ln_beta + ty * C : nullptr));
            }
            HMVIT_TRY(j1.flush());
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i;
                DropCfg di = drop_cfg(t, sidx, 1);
                di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                HMVIT_TRY(launch_gelu_drop(pre + (size_t)slot * P * mlp, h + (size_t)slot * P * mlp, (size_t)P * mlp, di, st));
            }
            Jobs j2(st, regp);
            const bool fused2 = j2.can_fuse_ln(reinterpret_cast<const float*>(wt.w_2), C, mlp);
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                DropCfg di = drop_cfg(t, sidx, 2);
                di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                if (fused2)       // x'' = x' + Dropout(h W_2^T + b_2) in the Linear's epilogue
                    HMVIT_TRY(j2.add(h + (size_t)slot * P * mlp, reinterpret_cast<const float*>(wt.w_2) + (size_t)ty * C * mlp, wt.b_2 + ty * C,
                                     x1 + slot * me, x_out + slot * me, P, C, mlp, nullptr, nullptr, &di));
                else
                    HMVIT_TRY(j2.add(h + (size_t)slot * P * mlp, reinterpret_cast<const float*>(wt.w_2) + (size_t)ty * C * mlp, wt.b_2 + ty * C,
                                     nullptr, x_out + slot * me, P, C, mlp));
            }
            HMVIT_TRY(j2.flush());
            if (!fused2)
                for (int i = 0; i < n_ego; ++i) {
                    const int slot = b * L + i;
                    DropCfg di = drop_cfg(t, sidx, 2);
                    di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                    HMVIT_TRY(launch_add_drop(x1 + slot * me, x_out + slot * me, x_out + slot * me, me, di, st));   // in place
                }
        }
    }
    if (pl.only_stage >= 0) {
        // one stage of the block on its own (the parallel block's branches, hetero_fusion.py:459-470): every agent's map goes
        // back as (B, L, C, H, W); padded slots, which no later step reads, are zero
        float* xf = S + pl.o_xfin;
        HMVIT_CHECK_HIP(hipMemsetAsync(d->out, 0, pl.A * 4, st));
        for (int b = 0; b < B; ++b)
            HMVIT_TRY(launch_transpose(xf + (size_t)b * L * me, d->out + (size_t)b * L * me, pl.max_cav, P, C, st));
        return HMVIT_OK;
    }
    // mlp_head on the ego map (no norm, no residual, dropout 0; bevformer_point_pillar_hetero.py:37,47-48)
    {
        float* xf = S + pl.o_xfin;
        float* hpre = S + pl.o_hpre;
        float* hh = S + pl.o_hh;
        Jobs j1(st, regp);
        for (int b = 0; b < B; ++b) {
            const int slot = b * L, ty = d->mode[slot];
            HMVIT_TRY(j1.add(xf + slot * me, reinterpret_cast<const float*>(d->head_w1) + (size_t)ty * C * C, d->head_b1 + ty * C, nullptr,
                             hpre + (size_t)b * me, P, C, C));
        }
        HMVIT_TRY(j1.flush());
        DropCfg none = {0ull, 0u, 0.f};
        HMVIT_TRY(launch_gelu_drop(hpre, hh, (size_t)B * me, none, st));
        Jobs j2(st, regp);
        for (int b = 0; b < B; ++b) {
            const int ty = d->mode[b * L];
            HMVIT_TRY(j2.add(hh + (size_t)b * me, reinterpret_cast<const float*>(d->head_w2) + (size_t)ty * C * C, d->head_b2 + ty * C, nullptr,
                             tmp + (size_t)b * me, P, C, C));
        }
        HMVIT_TRY(j2.flush());
        HMVIT_TRY(launch_transpose(tmp, d->out, B, P, C, st));
    }
    return HMVIT_OK;
}

// ---- backward workspace layout (floats) ----
struct BwdPlan {
    size_t o_G, o_T1, o_T2, o_T3, o_T4, o_h, o_pre, o_dkg, o_dkv, o_wt, o_img, o_winv, o_vbound, o_dbrep, total;
    // transposed weights inside o_wt, per stage s: q (T,C,C), kv (T,T,2,C,C), o (T,C,C), w1t (T,C,mlp), w2t (T,mlp,C); head: w1t, w2t
    size_t wt_stage, wt_q, wt_kv, wt_o, wt_1, wt_2, wt_h1, wt_h2;
};

static void make_bwd_plan(const TrainPlan& pl, BwdPlan& bp) {
    const size_t tok = (size_t)pl.n_slots * pl.P;
    const size_t big = pl.A > tok * pl.mlp ? pl.A : tok * pl.mlp;
    size_t off = 0;
    auto carve = [&](size_t n) { size_t o = off; off = (off + n + 63) / 64 * 64; return o; };
    // G, T1, T3, T4: one map per slot.  The two hidden-width buffers of the FFN section (T2 = d(pre), h = the recomputed activations)
    // are dead before the attention section starts and the gathered-key gradients (dkg: the largest buffer, 7.2 GB at cfg2) are dead
    // once k_warp_adjoint has run, so the three share one region (round 5: 2.9 GB less at cfg2, nothing else changes)
    const size_t dkg_n = (size_t)pl.B * pl.max_cav * pl.max_cav * 2 * pl.P * pl.C;
    bp.o_G = carve(pl.A); bp.o_T1 = carve(pl.A); bp.o_T3 = carve(pl.A); bp.o_T4 = carve(pl.A);
    const size_t ffn_n = (pl.save_pre ? 2 : 3) * big;       // + the recomputed pre-activations (recompute bit 0)
    bp.o_dkg = carve(dkg_n > ffn_n ? dkg_n : ffn_n);
    bp.o_T2 = bp.o_dkg; bp.o_h = bp.o_dkg + big; bp.o_pre = bp.o_dkg + 2 * big;
    bp.o_dkv = carve(tok * pl.E_max * 2 * pl.C);
    const size_t T = HMVIT_NUM_TYPES, C = pl.C, mlp = pl.mlp;
    size_t w = 0;
    bp.wt_q = w; w += T * C * C;
    bp.wt_kv = w; w += T * T * 2 * C * C;
    bp.wt_o = w; w += T * C * C;
    bp.wt_1 = w; w += T * C * mlp;
    bp.wt_2 = w; w += T * mlp * C;
    bp.wt_stage = w;
    bp.wt_h1 = 2 * w; bp.wt_h2 = 2 * w + T * C * C;
    bp.o_wt = carve(2 * w + 2 * T * C * C);
    // x16 images of the transposed weights (k_linear16), one float of storage per (hi, lo) pair, + one inverse scale per matrix
    bp.o_img = carve(pl.lin16 ? 2 * w + 2 * T * C * C : 0);
    bp.o_winv = carve(pl.lin16 ? (2 * w + 2 * T * C * C) / 65536 : 0);
    bp.o_vbound = carve(64);            // one a-priori |V'| bound per stage (launch_v_bound)
    bp.o_dbrep = carve(warp_adjoint_replica_floats(HMVIT_NUM_TYPES, pl.C));   // k_warp_adjoint's partial bias gradients
    bp.total = off;
}

#ifdef HMVIT_DBG_SUMS
// debug aid (tools/probe/r03_grad_bisect.sh): order-independent fingerprint of a device buffer, printed from the host
static void dbg_sum(const char* tag, int sidx, const float* p, size_t n, hipStream_t st) {
    std::vector<float> h(n);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost);
    double s = 0, a = 0;
    unsigned long long x = 0;
    for (size_t i = 0; i < n; ++i) { s += h[i]; a += fabs((double)h[i]); unsigned u; memcpy(&u, &h[i], 4); x ^= (unsigned long long)u * (i + 1); }
    fprintf(stderr, "DBG stage %d %-6s sum %.9e abs %.9e xor %016llx\n", sidx, tag, s, a, x);
    if (sidx == 3 && (tag[0] == 'd' && tag[1] == 'q')) {
        static int run = 0;
        char name[64];
        snprintf(name, sizeof(name), "/tmp/dbg_dq_%d.bin", run++);
        FILE* f = fopen(name, "wb");
        if (f) { fwrite(h.data(), 4, n, f); fclose(f); }
    }
}
#define DBG_SUM(tag, p, n) dbg_sum(tag, sidx, p, n, st)
#else
#define DBG_SUM(tag, p, n) do {} while (0)
#endif

static int train_backward(const HmvitFusionTrainDesc* t, const float* d_out, float* d_x, const HmvitStageGrads* grads,
                          float* d_head_w1, float* d_head_b1, float* d_head_w2, float* d_head_b2, void* workspace,
                          size_t workspace_bytes, hipStream_t st) {
    const HmvitFusionDesc* d = &t->fwd;
    TrainPlan pl;
    HMVIT_TRY(make_train_plan(t, pl));
    BwdPlan bp;
    make_bwd_plan(pl, bp);
    HMVIT_CHECK_ARG(d_out && d_x && grads && workspace && t->saved, "backward: null pointer");
    // the saved area must be the one a forward with THIS plan wrote (same recompute bits, sizes, stages): its offsets come from the plan
    if (t->saved_bytes < pl.total_floats * 4) {
        set_error("backward: saved area too small for this descriptor: %zu < %zu bytes (recompute / sizes differ from the forward's?)",
                  t->saved_bytes, pl.total_floats * 4);
        return HMVIT_ENOMEM;
    }
    const bool generic_attn = (d->window != 4 && d->window != 8) || d->dim_head != 32;
    HMVIT_CHECK_ARG(generic_attn || (t->bias_frag_neg[0] && t->bias_frag_neg[1]), "backward: bias_frag_neg is null");
    HMVIT_CHECK_ARG(pl.only_stage >= 0 || (d_head_w1 && d_head_b1 && d_head_w2 && d_head_b2), "backward: mlp_head gradient buffers are null");
    if (workspace_bytes < bp.total * 4) {
        set_error("backward workspace too small: %zu < %zu bytes", workspace_bytes, bp.total * 4);
        return HMVIT_ENOMEM;
    }
    const int B = d->B, L = d->L, C = d->C, P = pl.P, mlp = pl.mlp, T = HMVIT_NUM_TYPES;
    const size_t me = (size_t)P * C;
    float* S = reinterpret_cast<float*>(t->saved);
    float* Wk = reinterpret_cast<float*>(workspace);
    float *G = Wk + bp.o_G, *T1 = Wk + bp.o_T1, *T2 = Wk + bp.o_T2, *T3 = Wk + bp.o_T3, *T4 = Wk + bp.o_T4, *dkg = Wk + bp.o_dkg,
          *dkv = Wk + bp.o_dkv, *WT = Wk + bp.o_wt;
    const float* ainv = S + pl.o_ainv;
    const DropCfg none = {0ull, 0u, 0.f};

    // transposed weights: dA = dY W  is  k_gemm(dY, W^T)
    for (int s = 0; s < 2; ++s) {
        const HmvitStageWeights& wt = d->stage[s];
        float* base = WT + (size_t)s * bp.wt_stage;
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(wt.w_q), base + bp.wt_q, T, C, C, st));
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(wt.w_kv), base + bp.wt_kv, T * T * 2, C, C, st));
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(wt.w_o), base + bp.wt_o, T, C, C, st));
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(wt.w_1), base + bp.wt_1, T, mlp, C, st));   // (mlp, C) -> (C, mlp)
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(wt.w_2), base + bp.wt_2, T, C, mlp, st));   // (C, mlp) -> (mlp, C)
    }
    if (pl.only_stage < 0) {
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(d->head_w1), WT + bp.wt_h1, T, C, C, st));
        HMVIT_TRY(launch_transpose(reinterpret_cast<const float*>(d->head_w2), WT + bp.wt_h2, T, C, C, st));
    }
    ImgRegistry reg;
    const ImgRegistry* regp = nullptr;
    if (pl.lin16) {
        // every block of WT is a (256, 256) matrix: one launch images them all (the head's blocks only when they were written)
        const int n_mat = (int)((pl.only_stage < 0 ? bp.wt_h1 + 2 * (size_t)T * C * C : 2 * bp.wt_stage) / 65536);
        HMVIT_TRY(launch_weight_images16(WT, reinterpret_cast<half_t*>(Wk + bp.o_img), Wk + bp.o_winv, n_mat, st));
        reg.add(WT, n_mat, Wk + bp.o_img, Wk + bp.o_winv);
        if (!pl.save_pre || !pl.save_q) {      // the forward's own images of W_1 / W_q (kept in the saved area): the recomputation runs on them
            const WeightLayout wl = weight_layout(C, mlp);
            for (int s = 0; s < 2; ++s) {
                if (pl.only_stage >= 0 && s != pl.only_stage) continue;
                const size_t off = (size_t)s * wl.stage + wl.w1, offq = (size_t)s * wl.stage + wl.q;
                if (!pl.save_pre) reg.add(reinterpret_cast<const float*>(d->stage[s].w_1), T, S + pl.o_img + off, S + pl.o_winv + off / 65536);
                if (!pl.save_q) reg.add(reinterpret_cast<const float*>(d->stage[s].w_q), T, S + pl.o_img + offq, S + pl.o_winv + offq / 65536);
            }
        }
        regp = &reg;
    }

    // ---- mlp_head ----
    HMVIT_CHECK_HIP(hipMemsetAsync(G, 0, pl.A * 4, st));
    if (pl.only_stage >= 0) {
        // single stage: d_out is (B, L, C, H, W), the gradient of every agent's map
        for (int b = 0; b < B; ++b)
            HMVIT_TRY(launch_transpose(d_out + (size_t)b * L * me, G + (size_t)b * L * me, pl.max_cav, C, P, st));
    } else {
        float* dy = T1;                 // (B, P, C)
        float* dh = T2;
        const float* xf = S + pl.o_xfin;
        const float* hpre = S + pl.o_hpre;
        const float* hh = S + pl.o_hh;
        HMVIT_TRY(launch_transpose(d_out, dy, B, C, P, st));
        Jobs j1(st, regp);
        TnJobs tn(st);
        for (int b = 0; b < B; ++b) {
            const int ty = d->mode[b * L];
            HMVIT_TRY(j1.add(dy + (size_t)b * me, WT + bp.wt_h2 + (size_t)ty * C * C, nullptr, nullptr, dh + (size_t)b * me, P, C, C));
            HMVIT_TRY(tn.add(dy + (size_t)b * me, hh + (size_t)b * me, d_head_w2 + (size_t)ty * C * C, d_head_b2 + ty * C, P, C, C));
        }
        HMVIT_TRY(j1.flush());
        HMVIT_TRY(tn.flush());
        HMVIT_TRY(launch_gelu_bwd(hpre, dh, dh, (size_t)B * me, none, st));
        Jobs j2(st, regp);
        for (int b = 0; b < B; ++b) {
            const int slot = b * L, ty = d->mode[slot];
            HMVIT_TRY(j2.add(dh + (size_t)b * me, WT + bp.wt_h1 + (size_t)ty * C * C, nullptr, nullptr, G + slot * me, P, C, C));
            HMVIT_TRY(tn.add(dh + (size_t)b * me, xf + slot * me, d_head_w1 + (size_t)ty * C * C, d_head_b1 + ty * C, P, C, C));
        }
        HMVIT_TRY(j2.flush());
        HMVIT_TRY(tn.flush());
    }

    for (int sidx = pl.n_stages - 1; sidx >= 0; --sidx) {
        const int s = pl.only_stage < 0 ? (sidx & 1) : pl.only_stage;
        const HmvitStageWeights& wt = d->stage[s];
        const HmvitStageGrads& gr = grads[s];
        const StageInfo si = stage_info(d, pl, sidx);
        const float* R = S + (size_t)sidx * pl.stage_floats;
        const float *x_in = R + pl.o_x, *q = pl.save_q ? R + pl.o_q : T1, *kv = R + pl.o_kv, *o = R + pl.o_o, *lse = R + pl.o_lse,
                    *x1 = R + pl.o_x1, *pre = pl.save_pre ? R + pl.o_pre : Wk + bp.o_pre;
        // recomputed where a weight gradient needs them: the LayerNorm outputs into T3 (otherwise dO, which does not exist yet / any
        // more), the FFN activations into h (inside the dkg region)
        float *xn = T3, *xn2 = T3, *h = Wk + bp.o_h;
        const float* wts = WT + (size_t)s * bp.wt_stage;
        const int n_ego = si.n_ego;

        // recompute bit 0: pre = W_1 LN(x') + b_1 once more, by the forward's own call (same kernel, same weight images: bit-identical)
        if (!pl.save_pre) {
            float* pre_w = Wk + bp.o_pre;
            for (int b = 0; b < B; ++b) {
                Jobs j1(st, regp);
                const bool fuse = j1.can_fuse_ln(reinterpret_cast<const float*>(wt.w_1), mlp, C);
                if (!fuse) HMVIT_TRY(ln_slots(x1, xn2, wt.ffn_ln_gamma, wt.ffn_ln_beta, d, pl, n_ego, st, b));
                for (int i = 0; i < n_ego; ++i) {
                    const int slot = b * L + i, ty = d->mode[slot];
                    HMVIT_TRY(j1.add((fuse ? x1 : xn2) + slot * me, reinterpret_cast<const float*>(wt.w_1) + (size_t)ty * mlp * C, wt.b_1 + ty * mlp,
                                     nullptr, pre_w + (size_t)slot * P * mlp, P, mlp, C, fuse ? wt.ffn_ln_gamma + ty * C : nullptr,
                                     fuse ? wt.ffn_ln_beta + ty * C : nullptr));
                }
                HMVIT_TRY(j1.flush());
            }
        }
        // G = dL/dx'' (zero on slots the loss does not reach).  FFN: x'' = x' + drop(W_2 drop(gelu(W_1 LN(x') + b_1)) + b_2)
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i;
                DropCfg di = drop_cfg(t, sidx, 2);
                di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                HMVIT_TRY(launch_add_drop(nullptr, G + slot * me, T1 + slot * me, me, di, st));            // df
                DropCfg dh = drop_cfg(t, sidx, 1);
                dh.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                HMVIT_TRY(launch_gelu_drop(pre + (size_t)slot * P * mlp, h + (size_t)slot * P * mlp, (size_t)P * mlp, dh, st));
            }
            Jobs j1(st, regp);
            TnJobs tn(st);
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                HMVIT_TRY(j1.add(T1 + slot * me, wts + bp.wt_2 + (size_t)ty * mlp * C, nullptr, nullptr, T2 + (size_t)slot * P * mlp, P, mlp, C));
                HMVIT_TRY(tn.add(T1 + slot * me, h + (size_t)slot * P * mlp, gr.w_2 + (size_t)ty * C * mlp, gr.b_2 + ty * C, P, C, mlp));
            }
            HMVIT_TRY(j1.flush());
            HMVIT_TRY(tn.flush());
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i;
                DropCfg di = drop_cfg(t, sidx, 1);
                di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                float* dpre = T2 + (size_t)slot * P * mlp;
                HMVIT_TRY(launch_gelu_bwd(pre + (size_t)slot * P * mlp, dpre, dpre, (size_t)P * mlp, di, st));
            }
            HMVIT_TRY(ln_slots(x1, xn2, wt.ffn_ln_gamma, wt.ffn_ln_beta, d, pl, n_ego, st, b));
            Jobs j2(st, regp);
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                HMVIT_TRY(j2.add(T2 + (size_t)slot * P * mlp, wts + bp.wt_1 + (size_t)ty * C * mlp, nullptr, nullptr, T1 + slot * me, P, C, mlp));
                HMVIT_TRY(tn.add(T2 + (size_t)slot * P * mlp, xn2 + slot * me, gr.w_1 + (size_t)ty * mlp * C, gr.b_1 + ty * mlp, P, mlp, C));
            }
            HMVIT_TRY(j2.flush());
            HMVIT_TRY(tn.flush());
        }
        // G <- G + LN2-backward(dxn2 = T1)   (= dL/dx')
        HMVIT_TRY(ln_bwd_slots(x1, T1, wt.ffn_ln_gamma, G, G, gr.ffn_ln_gamma, gr.ffn_ln_beta, d, pl, n_ego, st));

        // x' = x + drop(O W_o^T + b_o): dO = drop'(G) W_o
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i;
                DropCfg di = drop_cfg(t, sidx, 0);
                di.seed += 0x51ED270B1ull * (unsigned long long)(slot + 1);
                HMVIT_TRY(launch_add_drop(nullptr, G + slot * me, T1 + slot * me, me, di, st));            // da
            }
            Jobs j1(st, regp);
            TnJobs tn(st);
            for (int i = 0; i < n_ego; ++i) {
                const int slot = b * L + i, ty = d->mode[slot];
                HMVIT_TRY(j1.add(T1 + slot * me, wts + bp.wt_o + (size_t)ty * C * C, nullptr, nullptr, T3 + slot * me, P, C, C));
                HMVIT_TRY(tn.add(T1 + slot * me, o + slot * me, gr.w_o + (size_t)ty * C * C, gr.b_o + ty * C, P, C, C));
            }
            HMVIT_TRY(j1.flush());
            HMVIT_TRY(tn.flush());
        }

        // recompute bit 1: q = LN(x) W_q once more, by the forward's own call, into T1 (free between the out-projection's gradient and
        // d(xn)); without the x16 kernel the normalised rows pass through T4 (dq: not written yet)
        if (!pl.save_q) {
            Jobs jb(st, regp);
            const bool fuse = jb.can_fuse_ln(reinterpret_cast<const float*>(wt.w_q), C, C);
            if (!fuse) HMVIT_TRY(ln_slots(x_in, T4, wt.ln_gamma, wt.ln_beta, d, pl, n_ego, st));
            for (int b = 0; b < B; ++b)
                for (int l = 0; l < n_ego; ++l) {
                    const int slot = b * L + l, ty = d->mode[slot];
                    HMVIT_TRY(jb.add((fuse ? x_in : T4) + slot * me, reinterpret_cast<const float*>(wt.w_q) + (size_t)ty * C * C, nullptr, nullptr,
                                     T1 + slot * me, P, C, C, fuse ? wt.ln_gamma + ty * C : nullptr, fuse ? wt.ln_beta + ty * C : nullptr));
                }
            HMVIT_TRY(jb.flush());
        }
        // attention backward: dq (T4), gradients of the gathered keys (dkg), bias fragments
        {
            AttnBwdParams ab;
            fill_attn(d, pl, si, s, q, kv, const_cast<float*>(o), const_cast<float*>(lse), ainv, ab.f);
            ab.bias_frag_neg = reinterpret_cast<const float*>(t->bias_frag_neg[s]);
            ab.d_out = T3; ab.dq = T4; ab.dkg = dkg; ab.d_bias_frag = gr.bias_frag;
            ab.probe = 0;
            // the range of the kernel's derived operands follows from the stage's weights (train.hip k_attention_bwd)
            ab.v_bound = nullptr;
            if (!generic_attn) {          // (the generic kernel is exact f32 on the vector ALU: no f16 operand to keep in range)
                ab.v_bound = Wk + bp.o_vbound + sidx;
                HMVIT_TRY(launch_v_bound(reinterpret_cast<const float*>(wt.w_kv), wt.b_kv, wt.ln_gamma, wt.ln_beta, T * T, T, C,
                                         Wk + bp.o_vbound + sidx, st));
            }
            // (k_attention_bwd writes every key row of every (ego, source < max_cav) pair, zeros where nothing is visible)
            DBG_SUM("G", G, pl.A);
            DBG_SUM("dO", T3, pl.A);
            HMVIT_TRY(launch_attention_bwd(ab, st));
            DBG_SUM("dq", T4, (size_t)n_ego * me);
            DBG_SUM("dkg", dkg, (size_t)B * n_ego * pl.max_cav * 2 * me);
            // biases are added after the gather: their gradients are column sums over the EGO pixels.  b_q: with the weight gradient
            // of W_q below (k_gemm_tn_split sums the rows it loads); b_kv: on the way through k_warp_adjoint, which reads every row of
            // dkg anyway.  (Round 4 ran k_colsum over all 11 maps per ego: a second pass over the 7.2 GB buffer, 1.2 ms per stage.)
            WarpAdjParams wa;
            memset(&wa, 0, sizeof(wa));
            wa.dkg = dkg; wa.ainv = ainv; wa.dkv = dkv;
            wa.B = B; wa.L = L; wa.n_ego = n_ego; wa.n_src = pl.max_cav; wa.E = si.E; wa.C = C; wa.H = d->H; wa.W = d->W;
            for (int i = 0; i < pl.n_slots; ++i) wa.ego_e[i] = ab.f.ego_e[i];
            wa.db_kv = gr.b_kv; wa.db_rep = Wk + bp.o_dbrep; wa.T = T;
            for (int i = 0; i < pl.n_slots; ++i) wa.mode[i] = (int8_t)d->mode[i];
            HMVIT_TRY(launch_warp_adjoint(wa, st));
            DBG_SUM("dkv", dkv, (size_t)pl.n_slots * si.E * 2 * me);
        }

        // dxn (T1) = dq W_q + sum_e (dK'_e W_k,e + dV'_e W_v,e);  weight gradients against xn
        {
            HMVIT_TRY(ln_slots(x_in, xn, wt.ln_gamma, wt.ln_beta, d, pl, pl.max_cav, st));
            TnJobs tn(st);
            // The terms of a slot: [dq W_q] + per variant e (dK'_e W_k,e + dV'_e W_v,e).  Where every weight has an x16 image they are
            // summed in registers, up to kMaxLinMats inputs per pass (one pass at one ego type: round 4 made three, each reading and
            // writing the whole of d(xn)); else term by term through the residual input.
            const int n_terms = 1 + 2 * si.E;
            auto term_of = [&](int slot, int l, int term, const float*& g2, const float*& w, float*& dw) {
                const int ty = d->mode[slot];
                if (term == 0) {
                    if (l >= n_ego) return false;
                    g2 = T4 + slot * me; w = wts + bp.wt_q + (size_t)ty * C * C; dw = gr.w_q + (size_t)ty * C * C;
                } else {
                    const int e = (term - 1) >> 1, pln = (term - 1) & 1;
                    const size_t widx = (size_t)((si.e_type[e] * T + ty) * 2 + pln) * C * C;
                    g2 = dkv + ((size_t)(slot * si.E + e) * 2 + pln) * me; w = wts + bp.wt_kv + widx; dw = gr.w_kv + widx;
                }
                return true;
            };
            // weight gradients against xn (b_q's column sums ride on W_q's)
            for (int b = 0; b < B; ++b)
                for (int l = 0; l < pl.max_cav; ++l)
                    for (int term = 0; term < n_terms; ++term) {
                        const int slot = b * L + l;
                        const float *g2, *w;
                        float* dw;
                        if (!term_of(slot, l, term, g2, w, dw)) continue;
                        HMVIT_TRY(tn.add(g2, xn + slot * me, dw, term == 0 ? gr.b_q + d->mode[slot] * C : nullptr, P, C, C));
                    }
            bool summed = true;
            {
                Jobs probe(st, regp);
                summed = probe.can_fuse_ln(wts + bp.wt_q, C, C);
            }
            if (summed) {
                // pass k takes terms [k kMaxLinMats, (k + 1) kMaxLinMats) of every slot's list; later passes add to y through the residual
                int max_terms = 0;
                for (int pass = 0; pass == 0 || pass * kMaxLinMats < max_terms; ++pass) {
                    Jobs jb(st, regp);
                    for (int b = 0; b < B; ++b)
                        for (int l = 0; l < pl.max_cav; ++l) {
                            const int slot = b * L + l;
                            const float* a_in[2 * HMVIT_NUM_TYPES + 1];
                            const float* w_in[2 * HMVIT_NUM_TYPES + 1];
                            int n = 0;
                            for (int term = 0; term < n_terms; ++term) {
                                float* dw;
                                if (term_of(slot, l, term, a_in[n], w_in[n], dw)) ++n;
                            }
                            if (n > max_terms) max_terms = n;
                            const int first = pass * kMaxLinMats, cnt = n - first < kMaxLinMats ? n - first : kMaxLinMats;
                            if (cnt <= 0) continue;
                            float* y = T1 + slot * me;
                            int rc;
                            if (!add_sum16(jb, a_in + first, w_in + first, cnt, pass == 0 ? nullptr : y, y, P, &rc)) {
                                set_error("training: a weight of the d(xn) sum has no x16 image%s", "");
                                return HMVIT_EINVAL;
                            }
                            HMVIT_TRY(rc);
                        }
                    HMVIT_TRY(jb.flush());
                }
            } else {
            // pass 0: first term of every slot (no accumulate), later passes accumulate through the residual input
            for (int term = 0; term < n_terms; ++term) {
                Jobs jb(st, regp);
                for (int b = 0; b < B; ++b)
                    for (int l = 0; l < pl.max_cav; ++l) {
                        const int slot = b * L + l;
                        float* y = T1 + slot * me;
                        const bool has_q = l < n_ego;
                        const int first_term = has_q ? 0 : 1;
                        const float* res = term == first_term ? nullptr : y;
                        const float *g2, *w;
                        float* dw;
                        if (!term_of(slot, l, term, g2, w, dw)) continue;
                        HMVIT_TRY(jb.add(g2, w, nullptr, res, y, P, C, C));
                    }
                HMVIT_TRY(jb.flush());
            }
            }
            HMVIT_TRY(tn.flush());
        }
        // dL/dx = (dL/dx' on the ego slots) + LN-backward(dxn)
        HMVIT_TRY(ln_bwd_slots(x_in, T1, wt.ln_gamma, G, G, gr.ln_gamma, gr.ln_beta, d, pl, pl.max_cav, st));
    }

    // token-major -> (B, L, C, H, W); padded agents receive no gradient
    HMVIT_CHECK_HIP(hipMemsetAsync(d_x, 0, pl.A * 4, st));
    for (int b = 0; b < B; ++b)
        HMVIT_TRY(launch_transpose(G + (size_t)b * L * me, d_x + (size_t)b * L * me, pl.max_cav, P, C, st));
    return HMVIT_OK;
}

}  // namespace hmvit

#ifdef HMVIT_PROBE
namespace hmvit { int debug_x16_trace(unsigned long long* host, int n); int debug_bwd_trace(unsigned long long* host, int n); }
#endif
using namespace hmvit;

extern "C" {

size_t hmvit_fusion_train_saved_bytes(const HmvitFusionTrainDesc* desc) {
    TrainPlan pl;
    if (!desc || make_train_plan(desc, pl) != HMVIT_OK) return 0;
    return pl.total_floats * 4;
}

size_t hmvit_fusion_backward_workspace_bytes(const HmvitFusionTrainDesc* desc) {
    TrainPlan pl;
    if (!desc || make_train_plan(desc, pl) != HMVIT_OK) return 0;
    BwdPlan bp;
    make_bwd_plan(pl, bp);
    return bp.total * 4;
}

int hmvit_fusion_train_forward(const HmvitFusionTrainDesc* desc, void* stream) {
    HMVIT_CHECK_ARG(desc != nullptr, "desc is null");
    return train_forward(desc, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_fusion_backward(const HmvitFusionTrainDesc* desc, const float* d_out, float* d_x, const HmvitStageGrads* grads,
                          float* d_head_w1, float* d_head_b1, float* d_head_w2, float* d_head_b2, void* workspace,
                          size_t workspace_bytes, void* stream) {
    HMVIT_CHECK_ARG(desc != nullptr, "desc is null");
    return train_backward(desc, d_out, d_x, grads, d_head_w1, d_head_b1, d_head_w2, d_head_b2, workspace, workspace_bytes,
                          reinterpret_cast<hipStream_t>(stream));
}

int hmvit_dropout_mask(float* mask, size_t n, uint64_t seed, uint32_t salt, float p, void* stream) {
    HMVIT_CHECK_ARG(mask != nullptr, "dropout_mask: null pointer");
    DropCfg c;
    c.seed = seed; c.salt = salt; c.p = p;
    return launch_dropout_mask(mask, n, c, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_gemm_tn(const float* dy, const float* a, float* dw, float* dbias, int M, int N, int K, int ld_dy, int ld_a, void* stream) {
    HMVIT_CHECK_ARG(dy && a && dw && M > 0 && N > 0 && K > 0 && ld_dy >= N && ld_a >= K, "gemm_tn: bad argument");
    GemmTnJobs jobs;
    jobs.n = 1;
    GemmTnJob& j = jobs.j[0];
    j.dy = dy; j.a = a; j.dw = dw; j.dbias = dbias; j.M = M; j.N = N; j.K = K; j.ld_dy = ld_dy; j.ld_a = ld_a;
    return launch_gemm_tn(jobs, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_linear16(const float* a, const float* w, const float* bias, const float* residual, float* y, int M, int n_mat,
                   float* image_ws, void* stream) {
    HMVIT_CHECK_ARG(a && w && y && image_ws && M > 0 && n_mat >= 1 && n_mat <= kMaxLinMats, "linear16: bad argument");
    HMVIT_CHECK_ARG(!residual || n_mat == 1, "linear16: a residual goes with a single matrix%s", "");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* inv = image_ws + (size_t)65536 * n_mat;
    HMVIT_TRY(launch_weight_images16(w, reinterpret_cast<half_t*>(image_ws), inv, n_mat, st));
    LinJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.n = 1;
    LinJob& j = jobs.j[0];
    j.a = a; j.residual = residual; j.M = M; j.n_mat = n_mat; j.ldy = 256 * n_mat;
    for (int m = 0; m < n_mat; ++m) {
        j.wimg[m] = reinterpret_cast<const half_t*>(image_ws) + (size_t)m * 2 * 65536;
        j.w_inv[m] = inv + m;
        j.bias[m] = bias ? bias + 256 * m : nullptr;
        j.y[m] = y + 256 * m;
    }
    return launch_linear16(jobs, st);
}

int hmvit_bn_train_stats(const float* x, float* sums, int M, int C, void* stream) {
    HMVIT_CHECK_ARG(x && sums, "bn_train_stats: null pointer");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.out = sums; a.M = M; a.C = C;
    return launch_bn(a, 0, 0, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_bn_train_stats_centered(const float* x, const float* pivot, float* sums, int M, int C, void* stream) {
    HMVIT_CHECK_ARG(x && pivot && sums, "bn_train_stats_centered: null pointer");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.mean = pivot; a.out = sums; a.M = M; a.C = C;
    return launch_bn(a, 0, 0, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_bn_train_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                         int M, int C, int relu, void* stream) {
    HMVIT_CHECK_ARG(x && mean && rstd && gamma && beta && y, "bn_train_apply: null pointer");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.out = y; a.M = M; a.C = C; a.relu = relu;
    return launch_bn(a, 0, 1, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_bn_train_backward(const float* x, const float* y, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            float* sums, float* dx, int M, int C, int relu, void* stream) {
    HMVIT_CHECK_ARG(x && dy && mean && rstd && gamma && sums && dx && (y || !relu), "bn_train_backward: null pointer");
    BnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.M = M; a.C = C; a.relu = relu;
    a.out = sums;
    HMVIT_TRY(launch_bn(a, 1, 0, reinterpret_cast<hipStream_t>(stream)));
    a.sums = sums; a.out = dx;
    return launch_bn(a, 1, 1, reinterpret_cast<hipStream_t>(stream));
}

#ifdef HMVIT_PROBE
int hmvit_debug_x16_trace(unsigned long long* host, int n) { return hmvit::debug_x16_trace(host, n); }
int hmvit_debug_bwd_trace(unsigned long long* host, int n) { return hmvit::debug_bwd_trace(host, n); }
#endif

}  // extern "C"
