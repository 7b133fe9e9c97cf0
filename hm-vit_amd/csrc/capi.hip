// C ABI of libhmvit: argument checking, workspace carving and the launch sequence of one
// HeteroFusion forward (bevformer_point_pillar_hetero.py:39-49, hetero_fusion.py:363-474).
#include <string.h>

#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace hmvit {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// bytes per element of the Q / K' / V' / O planes
static size_t elem_size(int precision, int C) {
    return (precision == HMVIT_PREC_F16 || (precision == HMVIT_PREC_MIXED && C == 256)) ? 2 : 4;
}

struct Plan {
    int B, L, C, H, W, P, mlp, n_slots, max_cav, E_max;
    size_t es;
    // byte offsets into the workspace
    size_t off_xs, off_xn, off_q, off_kv, off_o, off_hid, off_ainv, off_ytok, off_xa, off_xb, off_gap, off_sw, off_vis, off_need, off_sched, off_ptab, total;
};

int check_desc(const HmvitFusionDesc* d) {
    HMVIT_CHECK_ARG(d != nullptr, "desc is null");
    HMVIT_CHECK_ARG(d->B > 0 && d->L > 0 && d->L <= HMVIT_MAX_AGENTS, "B=%d L=%d out of range (L <= %d)", d->B,
                    d->L, HMVIT_MAX_AGENTS);
    HMVIT_CHECK_ARG(d->dim_head >= 1 && d->C == d->heads * d->dim_head, "C=%d != heads*dim_head=%d*%d", d->C, d->heads, d->dim_head);
    HMVIT_CHECK_ARG(d->C == 64 || d->C == 128 || d->C == 256, "C=%d unsupported (64, 128, 256)", d->C);
    // the tuned kernels serve window 4 / 8 with dim_head 32 (the shipped yamls); anything else the reference accepts runs on the
    // generic exact-f32 attention kernel: inference, HMVIT_PREC_F32 only (bias_frag then carries the dense (heads, N, N) bias)
    const bool generic_shape = (d->window != 4 && d->window != 8) || d->dim_head != 32;
    HMVIT_CHECK_ARG(d->window >= 1 && d->window * d->window <= 256, "window_size=%d unsupported (window^2 <= 256)", d->window);
    HMVIT_CHECK_ARG(!generic_shape || (d->precision == HMVIT_PREC_F32 && d->dim_head <= 64),
                    "window_size=%d / dim_head=%d: generic shapes run with precision HMVIT_PREC_F32 and dim_head <= 64 only", d->window,
                    d->dim_head);
    HMVIT_CHECK_ARG(d->H > 0 && d->W > 0 && d->H % d->window == 0 && d->W % d->window == 0,
                    "BEV %dx%d must be divisible by window_size %d", d->H, d->W, d->window);
    HMVIT_CHECK_ARG(d->mlp_dim > 0 && d->mlp_dim % 64 == 0, "mlp_dim=%d must be a multiple of 64", d->mlp_dim);
    HMVIT_CHECK_ARG(d->num_iters >= 1, "num_iters=%d", d->num_iters);
    HMVIT_CHECK_ARG(d->precision == HMVIT_PREC_F32 || d->precision == HMVIT_PREC_F16 || d->precision == HMVIT_PREC_SPLIT ||
                        d->precision == HMVIT_PREC_MIXED, "precision=%d", d->precision);
    HMVIT_CHECK_ARG(d->mode && d->record_len && d->cav_mask, "mode / record_len / cav_mask must be host arrays");
    HMVIT_CHECK_ARG(d->discrete_ratio * d->downsample_rate != 0.f, "discrete_ratio * downsample_rate is 0");
    HMVIT_CHECK_ARG(!d->parallel || (d->split_fc1 && d->split_ln_g && d->split_ln_b && d->split_fc2),
                    "architect_mode parallel: SplitAttn weights are null");
    for (int i = 0; i < d->B * d->L; ++i)
        HMVIT_CHECK_ARG(d->mode[i] >= 0 && d->mode[i] < HMVIT_NUM_TYPES, "mode[%d]=%d is not an agent type", i,
                        d->mode[i]);
    for (int b = 0; b < d->B; ++b)
        HMVIT_CHECK_ARG(d->record_len[b] >= 1 && d->record_len[b] <= d->L, "record_len[%d]=%d out of [1, %d]", b,
                        d->record_len[b], d->L);
    return HMVIT_OK;
}

static void make_plan(const HmvitFusionDesc* d, Plan& pl) {
    pl.B = d->B; pl.L = d->L; pl.C = d->C; pl.H = d->H; pl.W = d->W;
    pl.P = d->H * d->W; pl.mlp = d->mlp_dim; pl.n_slots = d->B * d->L;
    pl.es = elem_size(d->precision, d->C);
    pl.max_cav = 0;
    for (int b = 0; b < d->B; ++b) pl.max_cav = d->record_len[b] > pl.max_cav ? d->record_len[b] : pl.max_cav;
    bool seen[HMVIT_NUM_TYPES] = {false, false};
    for (int b = 0; b < d->B; ++b)
        for (int i = 0; i < pl.max_cav; ++i) seen[d->mode[b * d->L + i]] = true;
    pl.E_max = (int)seen[0] + (int)seen[1];
    const size_t tok = (size_t)pl.n_slots * pl.P;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    pl.off_xs = carve(tok * pl.C * 4);
    pl.off_xn = carve(tok * pl.C * pl.es);
    pl.off_q = carve(tok * pl.C * pl.es);
    const size_t kv_bytes = tok * pl.E_max * 2 * pl.C * pl.es;
    const size_t hid_bytes = tok * pl.mlp * pl.es;
    pl.off_kv = carve(kv_bytes > hid_bytes ? kv_bytes : hid_bytes);
    pl.off_hid = pl.off_kv;  // the FFN hidden activations reuse the K/V planes (dead after attention)
    pl.off_o = carve(tok * pl.C * pl.es);
    pl.off_ainv = carve((size_t)pl.n_slots * pl.L * 8 * 4);
    pl.off_ytok = carve((size_t)pl.B * pl.P * pl.C * 4);
    pl.off_vis = carve((size_t)pl.n_slots * pl.P / 64 * 4 + 256);   // visible-chunk bits of the attention windows
    pl.off_need = carve(2 * ((size_t)pl.n_slots * pl.P / 64) + 256);      // reachable windows of the stage before the pruned one
    pl.off_sched = carve(attn_schedule_bytes(pl.B, pl.max_cav, pl.H, pl.W));   // world-ordered item list of the local stages
    // k_attention_patch16's per-item tables (HmvitFusionDesc::rigid_patch == 2, split mode only)
    pl.off_ptab = 0;
    if (d->rigid_patch == 2 && d->precision == HMVIT_PREC_SPLIT && d->window == 8 && d->C == 256 && d->H % 8 == 0 && d->W % 8 == 0)
        pl.off_ptab = carve(patch16_tables_bytes(pl.B, pl.max_cav, pl.H, pl.W));
    pl.off_xa = pl.off_xb = pl.off_gap = pl.off_sw = 0;
    if (d->parallel) {
        // branch outputs of the parallel block + SplitAttn scratch
        pl.off_xa = carve(tok * pl.C * 4);
        pl.off_xb = carve(tok * pl.C * 4);
        pl.off_gap = carve((size_t)pl.n_slots * ((pl.P + 255) / 256) * pl.C * 4);
        pl.off_sw = carve((size_t)pl.n_slots * 2 * pl.C * 4);
    }
    pl.total = off;
}

struct JobBatcher {
    GemmJobs jobs;
    bool a_f32, gelu, out_f32;
    int precision;
    hipStream_t st;
    JobBatcher(bool af, bool ge, bool of, int prec, hipStream_t s) : a_f32(af), gelu(ge), out_f32(of), precision(prec), st(s) {
        jobs.n = 0;
    }
    int flush() {
        if (jobs.n == 0) return HMVIT_OK;
        int rc = launch_gemm(jobs, a_f32, gelu, out_f32, precision, st);
        jobs.n = 0;
        return rc;
    }
    int add(const GemmJob& j) {
        jobs.j[jobs.n++] = j;
        if (jobs.n == kMaxJobs) return flush();
        return HMVIT_OK;
    }
};

// Optional phase timer (hmvit_fusion_profile): one HIP event after every phase's launches, on the
// stream the kernels run on.
struct PhaseTimer {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    std::vector<int> phase;
    std::vector<int> launched;          // 0: the interval closed by this mark held no launch of its phase (counted as time only)
    int* d_items = nullptr;             // device counters: live attention items per stage (launch_count_live)
    int n_stages = 0;
    int total_items[16];
    int mark(int ph, bool did_launch = true) {
        hipEvent_t e;
        HMVIT_CHECK_HIP(hipEventCreate(&e));
        HMVIT_CHECK_HIP(hipEventRecord(e, st));
        ev.push_back(e);
        phase.push_back(ph);
        launched.push_back(did_launch ? 1 : 0);
        return HMVIT_OK;
    }
    ~PhaseTimer() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        if (d_items) (void)hipFree(d_items);
    }
};
#define HMVIT_MARK(ph)                                   \
    do {                                                 \
        if (timer) {                                     \
            int _rc = timer->mark(ph);                   \
            if (_rc != HMVIT_OK) return _rc;             \
        }                                                \
    } while (0)

#define HMVIT_MARK_IF(ph, did)                           \
    do {                                                 \
        if (timer) {                                     \
            int _rc = timer->mark(ph, did);              \
            if (_rc != HMVIT_OK) return _rc;             \
        }                                                \
    } while (0)

#define HMVIT_TRY(expr)             \
    do {                            \
        int _rc = (expr);           \
        if (_rc != HMVIT_OK) return _rc; \
    } while (0)

// SplitAttn merge at the end of a parallel-mode iteration (hetero_fusion.py:459-470): all agent
// slots, or only the ego slots when nothing else is consumed afterwards
static int merge_branches(const HmvitFusionDesc* d, const Plan& pl, bool ego_only, hipStream_t st) {
    char* ws = reinterpret_cast<char*>(d->workspace);
    SplitSlots slots;
    memset(&slots, 0, sizeof(slots));
    int n = 0;
    for (int b = 0; b < pl.B; ++b)
        for (int l = 0; l < (ego_only ? 1 : pl.L); ++l) slots.s[n++] = (int8_t)(b * pl.L + l);
    SplitWeights sw = {d->split_fc1, d->split_ln_g, d->split_ln_b, d->split_fc2};
    return launch_split_attn(reinterpret_cast<float*>(ws + pl.off_xa), reinterpret_cast<float*>(ws + pl.off_xb),
                             reinterpret_cast<float*>(ws + pl.off_xs), slots, n, sw,
                             reinterpret_cast<float*>(ws + pl.off_gap), reinterpret_cast<float*>(ws + pl.off_sw), pl.P,
                             pl.C, st);
}

static int fusion_forward_f16(const HmvitFusionDesc* d, const Plan& pl, hipStream_t st, PhaseTimer* timer);

static int fusion_forward(const HmvitFusionDesc* d, hipStream_t st, PhaseTimer* timer) {
    HMVIT_TRY(check_desc(d));
    Plan pl;
    make_plan(d, pl);
    HMVIT_CHECK_ARG(d->x && d->pairwise_t && d->out && d->workspace, "x / pairwise_t / out / workspace is null");
    if (d->workspace_bytes < pl.total) {
        set_error("workspace too small: %zu < %zu bytes", d->workspace_bytes, pl.total);
        return HMVIT_ENOMEM;
    }
    HMVIT_CHECK_ARG(pl.n_slots <= kMaxSlots, "B*L=%d exceeds %d agent slots per call", pl.n_slots, kMaxSlots);
    if (d->precision != HMVIT_PREC_F32) return fusion_forward_f16(d, pl, st, timer);   // fused chains: f16 or split operands

    const int B = pl.B, L = pl.L, C = pl.C, P = pl.P, mlp = pl.mlp, prec = d->precision;
    const size_t es = pl.es;
    char* ws = reinterpret_cast<char*>(d->workspace);
    float* xs = reinterpret_cast<float*>(ws + pl.off_xs);
    char* xn = ws + pl.off_xn;
    char* qb = ws + pl.off_q;
    char* kvb = ws + pl.off_kv;
    char* ob = ws + pl.off_o;
    char* hid = ws + pl.off_hid;
    float* ainv = reinterpret_cast<float*>(ws + pl.off_ainv);
    float* ytok = reinterpret_cast<float*>(ws + pl.off_ytok);
    const size_t map_elems = (size_t)P * C;
    const bool par = d->parallel != 0;
    float* xa = reinterpret_cast<float*>(ws + pl.off_xa);
    float* xb = reinterpret_cast<float*>(ws + pl.off_xb);

    AgentTypes all_types;
    memset(&all_types, 0, sizeof(all_types));
    for (int i = 0; i < pl.n_slots; ++i) all_types.t[i] = (int8_t)d->mode[i];

    // NCHW -> token-major residual stream; sampling maps of every (source, ego) pair
    HMVIT_MARK(-1);
    HMVIT_TRY(launch_transpose(d->x, xs, pl.n_slots, C, P, st));
    HMVIT_TRY(launch_pair_affines(d->pairwise_t, ainv, pl.n_slots * L, d->H, d->W, d->discrete_ratio,
                                  d->downsample_rate, st));
    HMVIT_MARK(HMVIT_PHASE_LAYOUT_IN);

    for (int it = 0; it < d->num_iters; ++it) {
        for (int s = 0; s < 2; ++s) {
            const HmvitStageWeights& wt = d->stage[s];
            // In the last stage of HeteroFusion only ego 0 is consumed (x[:, 0],
            // bevformer_point_pillar_hetero.py:47): the other egos' rows are dead code.
            // (parallel mode: both branches of the last iteration only feed ego 0)
            const bool last = d->apply_head && it == d->num_iters - 1 && (par || s == 1);
            float* x_in = xs;                                   // stage input
            float* x_out = par ? (s == 0 ? xa : xb) : xs;       // stage output (in place when sequential)
            const int n_ego = last ? 1 : pl.max_cav;
            const int n_src = pl.max_cav;

            // K/V variants needed = ego types taking part in this stage
            int e_of_type[HMVIT_NUM_TYPES] = {-1, -1};
            int e_type[HMVIT_NUM_TYPES] = {0, 0};
            int E = 0;
            for (int b = 0; b < B; ++b)
                for (int i = 0; i < n_ego; ++i) {
                    const int t = d->mode[b * L + i];
                    if (e_of_type[t] < 0) { e_of_type[t] = E; e_type[E] = t; ++E; }
                }

            // 1. typed LayerNorm of every agent map (sources j < max_cav are all that is read)
            HMVIT_TRY(launch_layernorm(x_in, xn, wt.ln_gamma, wt.ln_beta, all_types, pl.n_slots, P, C, prec, st));
            HMVIT_MARK(HMVIT_PHASE_LN_ATTN);

            // 2. Q and relation-folded K/V projections (no bias: added after the gather)
            {
                JobBatcher jb(false, false, false, prec, st);
                for (int b = 0; b < B; ++b)
                    for (int l = 0; l < pl.max_cav; ++l) {
                        const int slot = b * L + l, t = d->mode[slot];
                        GemmJob j;
                        j.a = xn + (size_t)slot * map_elems * es;
                        j.bias = nullptr; j.residual = nullptr;
                        j.M = P; j.K = C; j.n_per_plane = C; j.plane_stride = (long long)map_elems;
                        if (l < n_ego) {
                            j.w = reinterpret_cast<const char*>(wt.w_q) + (size_t)t * C * C * es;
                            j.y = qb + (size_t)slot * map_elems * es;
                            j.N = C;
                            HMVIT_TRY(jb.add(j));
                        }
                        for (int e = 0; e < E; ++e) {
                            j.w = reinterpret_cast<const char*>(wt.w_kv) +
                                  (size_t)(e_type[e] * HMVIT_NUM_TYPES + t) * 2 * C * C * es;
                            j.y = kvb + (size_t)(slot * E + e) * 2 * map_elems * es;
                            j.N = 2 * C;
                            HMVIT_TRY(jb.add(j));
                        }
                    }
                HMVIT_TRY(jb.flush());
            }
            HMVIT_MARK(HMVIT_PHASE_QKV);

            // 3. fused warp + partition + attention for every ego
            {
                AttnParams ap;
                memset(&ap, 0, sizeof(ap));
                ap.q = qb; ap.kv = kvb; ap.b_q = wt.b_q; ap.b_kv = wt.b_kv; ap.bias_frag = wt.bias_frag;
                ap.ainv = ainv; ap.out = ob;
                ap.B = B; ap.L = L; ap.n_ego = n_ego; ap.n_src = n_src; ap.E = E; ap.C = C; ap.H = d->H; ap.W = d->W;
                ap.window = d->window; ap.partition = s == 0 ? HMVIT_PART_WINDOW : HMVIT_PART_GRID;
                ap.skip_masked = d->skip_masked; ap.dim_head = d->dim_head;
                for (int i = 0; i < pl.n_slots; ++i) {
                    ap.mode[i] = (int8_t)d->mode[i];
                    ap.cav[i] = (int8_t)(d->cav_mask[i] != 0);
                    ap.ego_e[i] = (int8_t)(e_of_type[d->mode[i]] < 0 ? 0 : e_of_type[d->mode[i]]);
                }
                HMVIT_TRY(launch_attention(ap, prec, st));
            }
            HMVIT_MARK(HMVIT_PHASE_ATTENTION);

            // 4. typed output projection + residual (in place on the f32 stream)
            {
                JobBatcher jb(false, false, true, prec, st);
                for (int b = 0; b < B; ++b)
                    for (int i = 0; i < n_ego; ++i) {
                        const int slot = b * L + i, t = d->mode[slot];
                        GemmJob j;
                        j.a = ob + (size_t)slot * map_elems * es;
                        j.w = reinterpret_cast<const char*>(wt.w_o) + (size_t)t * C * C * es;
                        j.bias = wt.b_o + t * C;
                        j.residual = x_in + (size_t)slot * map_elems;
                        j.y = x_out + (size_t)slot * map_elems;
                        j.M = P; j.N = C; j.K = C; j.n_per_plane = C; j.plane_stride = 0;
                        HMVIT_TRY(jb.add(j));
                    }
                HMVIT_TRY(jb.flush());
                if (par && !last)   // agents without an attention update enter the FFN unchanged
                    for (int b = 0; b < B; ++b)
                        for (int l = n_ego; l < L; ++l)
                            HMVIT_CHECK_HIP(hipMemcpyAsync(x_out + (size_t)(b * L + l) * map_elems,
                                                           x_in + (size_t)(b * L + l) * map_elems, map_elems * 4,
                                                           hipMemcpyDeviceToDevice, st));
            }
            HMVIT_MARK(HMVIT_PHASE_OUT_PROJ);

            // 5. pre-norm typed FFN + residual for every agent (only the egos in the last stage)
            {
                const int n_ffn = last ? 1 : L;
                if (n_ffn == L) {
                    HMVIT_TRY(launch_layernorm(x_out, xn, wt.ffn_ln_gamma, wt.ffn_ln_beta, all_types, pl.n_slots, P, C,
                                               prec, st));
                } else {
                    for (int b = 0; b < B; ++b) {
                        AgentTypes one;
                        memset(&one, 0, sizeof(one));
                        one.t[0] = (int8_t)d->mode[b * L];
                        HMVIT_TRY(launch_layernorm(x_out + (size_t)b * L * map_elems, xn + (size_t)b * L * map_elems * es,
                                                   wt.ffn_ln_gamma, wt.ffn_ln_beta, one, 1, P, C, prec, st));
                    }
                }
                HMVIT_MARK(HMVIT_PHASE_LN_FFN);
                JobBatcher j1(false, true, false, prec, st);
                for (int b = 0; b < B; ++b)
                    for (int l = 0; l < n_ffn; ++l) {
                        const int slot = b * L + l, t = d->mode[slot];
                        GemmJob j;
                        j.a = xn + (size_t)slot * map_elems * es;
                        j.w = reinterpret_cast<const char*>(wt.w_1) + (size_t)t * mlp * C * es;
                        j.bias = wt.b_1 + t * mlp;
                        j.residual = nullptr;
                        j.y = hid + (size_t)slot * P * mlp * es;
                        j.M = P; j.N = mlp; j.K = C; j.n_per_plane = mlp; j.plane_stride = 0;
                        HMVIT_TRY(j1.add(j));
                    }
                HMVIT_TRY(j1.flush());
                HMVIT_MARK(HMVIT_PHASE_FFN1);
                JobBatcher j2(false, false, true, prec, st);
                for (int b = 0; b < B; ++b)
                    for (int l = 0; l < n_ffn; ++l) {
                        const int slot = b * L + l, t = d->mode[slot];
                        GemmJob j;
                        j.a = hid + (size_t)slot * P * mlp * es;
                        j.w = reinterpret_cast<const char*>(wt.w_2) + (size_t)t * C * mlp * es;
                        j.bias = wt.b_2 + t * C;
                        j.residual = x_out + (size_t)slot * map_elems;
                        j.y = x_out + (size_t)slot * map_elems;
                        j.M = P; j.N = C; j.K = mlp; j.n_per_plane = C; j.plane_stride = 0;
                        HMVIT_TRY(j2.add(j));
                    }
                HMVIT_TRY(j2.flush());
                HMVIT_MARK(HMVIT_PHASE_FFN2);
            }
        }
        if (par) {
            HMVIT_TRY(merge_branches(d, pl, d->apply_head && it == d->num_iters - 1, st));
            HMVIT_MARK(HMVIT_PHASE_LAYOUT_OUT);
        }
    }

    if (!d->apply_head) {
        HMVIT_TRY(launch_transpose(xs, d->out, pl.n_slots, P, C, st));
        HMVIT_MARK(HMVIT_PHASE_LAYOUT_OUT);
        return HMVIT_OK;
    }

    // mlp_head on the ego map: Linear -> GELU -> Linear, no norm, no residual
    // (bevformer_point_pillar_hetero.py:37,47-48)
    HMVIT_CHECK_ARG(d->head_w1 && d->head_b1 && d->head_w2 && d->head_b2, "mlp_head weights are null");
    {
        JobBatcher j1(true, true, false, prec, st);
        for (int b = 0; b < B; ++b) {
            const int slot = b * L, t = d->mode[slot];
            GemmJob j;
            j.a = xs + (size_t)slot * map_elems;
            j.w = reinterpret_cast<const char*>(d->head_w1) + (size_t)t * C * C * es;
            j.bias = d->head_b1 + t * C;
            j.residual = nullptr;
            j.y = hid + (size_t)slot * map_elems * es;
            j.M = P; j.N = C; j.K = C; j.n_per_plane = C; j.plane_stride = 0;
            HMVIT_TRY(j1.add(j));
        }
        HMVIT_TRY(j1.flush());
        JobBatcher j2(false, false, true, prec, st);
        for (int b = 0; b < B; ++b) {
            const int slot = b * L, t = d->mode[slot];
            GemmJob j;
            j.a = hid + (size_t)slot * map_elems * es;
            j.w = reinterpret_cast<const char*>(d->head_w2) + (size_t)t * C * C * es;
            j.bias = d->head_b2 + t * C;
            j.residual = nullptr;
            j.y = ytok + (size_t)b * map_elems;
            j.M = P; j.N = C; j.K = C; j.n_per_plane = C; j.plane_stride = 0;
            HMVIT_TRY(j2.add(j));
        }
        HMVIT_TRY(j2.flush());
    }
    HMVIT_MARK(HMVIT_PHASE_HEAD);
    HMVIT_TRY(launch_transpose(ytok, d->out, B, P, C, st));
    HMVIT_MARK(HMVIT_PHASE_LAYOUT_OUT);
    return HMVIT_OK;
}

// ------------------------------------------------------------------------------------------
// fused modes (f16 operands, or split hi / lo f16 operands = fp32-class products): three launches per stage
// (chain.hip + attn.hip), no layout kernels
// ------------------------------------------------------------------------------------------
struct QkvBatcher {
    QkvParams p;
    int n, C;
    int split;
    hipStream_t st;
    int* pull_base = nullptr;    // zeroed ints, 8 apart (the padding word of consecutive affine records); pull_left of them
    int pull_left = 0;
    int flush() {
        if (n == 0) return HMVIT_OK;
        p.pull = nullptr;
        if (pull_left > 0) { p.pull = pull_base; pull_base += 8; --pull_left; }
        int rc = launch_ln_qkv(p, n, C, split, st);
        n = 0;
        return rc;
    }
    int add(const QkvJob& j) {
        p.job[n++] = j;
        if (n == kMaxChainJobs) return flush();
        return HMVIT_OK;
    }
};
struct FfnBatcher {
    FfnParams p;
    int n, C, variant;
    int split;
    hipStream_t st;
    int flush() {
        if (n == 0) return HMVIT_OK;
        int rc = launch_out_ffn(p, n, C, variant, split, st);
        n = 0;
        return rc;
    }
    int add(const FfnJob& j) {
        p.job[n++] = j;
        if (n == kMaxChainJobs) return flush();
        return HMVIT_OK;
    }
};

static int fusion_forward_f16(const HmvitFusionDesc* d, const Plan& pl, hipStream_t st, PhaseTimer* timer) {
    HMVIT_CHECK_ARG(d->mlp_dim == d->C, "fused modes: mlp_dim=%d must equal input_dim=%d (fused FFN kernel)", d->mlp_dim, d->C);
    for (int s = 0; s < 2; ++s)
        HMVIT_CHECK_ARG(d->stage[s].img_q && d->stage[s].img_kv && d->stage[s].img_o && d->stage[s].img_ffn,
                        "fused modes: stage %d weight images are null", s);
    const int B = pl.B, L = pl.L, C = pl.C, P = pl.P;
    // 0: f16 operands; 1: split (hi + lo) operands, f32 Q / K' / V' / O planes; 2 ("mixed", C = 256): split operands in every
    // Linear / FFN, f16 planes and the f16 attention kernels (log2(e) folded into W_q / bias as in f16 mode)
    const int split = d->precision == HMVIT_PREC_SPLIT ? 1 : d->precision == HMVIT_PREC_MIXED ? (C == 256 ? 2 : 1) : 0;
    const size_t es = pl.es;
    char* ws = reinterpret_cast<char*>(d->workspace);
    float* xs = reinterpret_cast<float*>(ws + pl.off_xs);
    char* qb = ws + pl.off_q;
    char* kvb = ws + pl.off_kv;
    char* ob = ws + pl.off_o;
    float* ainv = reinterpret_cast<float*>(ws + pl.off_ainv);
    const size_t map_elems = (size_t)P * C;
    const size_t map_bytes = map_elems * es;    // one projected (P, C) plane
    const size_t img_elems = (size_t)C * C * (split ? 2 : 1);   // halves of one (C, C) matrix image

    HMVIT_MARK(-1);
    HMVIT_TRY(launch_pair_affines(d->pairwise_t, ainv, pl.n_slots * L, d->H, d->W, d->discrete_ratio,
                                  d->downsample_rate, st));
    if (!d->apply_head && pl.max_cav < L) {
        // block output covers padded agents too: they never pass through k_ln_qkv, bring them in here
        for (int b = 0; b < B; ++b)
            HMVIT_TRY(launch_transpose(d->x + (size_t)(b * L + pl.max_cav) * map_elems,
                                       xs + (size_t)(b * L + pl.max_cav) * map_elems, L - pl.max_cav, C, P, st));
    }
    HMVIT_MARK(HMVIT_PHASE_LAYOUT_IN);

    // ---- per-stage descriptions ----
    struct StageInfo {
        bool last;
        int n_ego, E;
        int e_of_type[HMVIT_NUM_TYPES], e_type[HMVIT_NUM_TYPES];
    };
    const bool par = d->parallel != 0;
    auto stage_info = [&](int it, int s) {
        StageInfo si;
        si.last = d->apply_head && it == d->num_iters - 1 && (par || s == 1);
        si.n_ego = si.last ? 1 : pl.max_cav;
        si.E = 0;
        for (int t = 0; t < HMVIT_NUM_TYPES; ++t) { si.e_of_type[t] = -1; si.e_type[t] = 0; }
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < si.n_ego; ++i) {
                const int t = d->mode[b * L + i];
                if (si.e_of_type[t] < 0) { si.e_of_type[t] = si.E; si.e_type[si.E] = t; ++si.E; }
            }
        return si;
    };
    // The first stage's fused tail can take its residual straight from the (C, P) input maps: k_ln_qkv then writes no
    // token-major f32 copy of the input (721 MB at cfg2).  Needs the fused tail (sequential block, C = 256, a next stage).
    const bool direct_x = !par && C == 256 && !HMVIT_ENV("HMVIT_NO_FUSE") && !HMVIT_ENV("HMVIT_NO_DIRECT_X");   // = `fuse` of stage 1

    // LayerNorm + Q / folded K', V' projections of agent `slot` for the stage (wt, si)
    auto qkv_job = [&](const HmvitStageWeights& wt, const StageInfo& si, int slot, int l, bool first) {
        const int t = d->mode[slot];
        QkvJob j;
        memset(&j, 0, sizeof(j));
        j.x = first ? d->x + (size_t)slot * map_elems : xs + (size_t)slot * map_elems;
        j.xs_out = (first && direct_x) ? nullptr : xs + (size_t)slot * map_elems;
        j.type = t;
        int nm = 0;
        const HmvitStageScales* sc = split ? wt.scales : nullptr;
        for (int i = 0; i < 5; ++i) j.c[i] = 1.f;
        if (l < si.n_ego) {
            j.w[nm] = reinterpret_cast<const half_t*>(wt.img_q) + (size_t)t * img_elems;
            j.y[nm] = qb + (size_t)slot * map_bytes;
            if (sc) j.c[nm] = sc->c_q[t];
            ++nm;
        }
        for (int e = 0; e < si.E; ++e) {
            const half_t* wkv = reinterpret_cast<const half_t*>(wt.img_kv) +
                                (size_t)(si.e_type[e] * HMVIT_NUM_TYPES + t) * 2 * img_elems;
            char* ykv = kvb + (size_t)(slot * si.E + e) * 2 * map_bytes;
            if (sc) { j.c[nm] = sc->c_k[si.e_type[e]][t]; j.c[nm + 1] = sc->c_v[si.e_type[e]][t]; }
            j.w[nm] = wkv;             j.y[nm] = ykv;             ++nm;
            j.w[nm] = wkv + img_elems; j.y[nm] = ykv + map_bytes; ++nm;
        }
        j.n_mat = nm;
        return j;
    };

    // Reachability tables (k_window_need): HeteroFusion's last stage only computes ego 0, so in the local stage before it
    // the windows of the other agents that ego 0's taps cannot reach are dead code (need_last: attention items and chain
    // tail skipped), and the chain tail of the stage before that only has to produce what those surviving windows read
    // (need_prev).  Identical output; off with skip_masked = 0.
    unsigned char* need_last = nullptr;
    unsigned char* need_prev = nullptr;
    if (!par && d->apply_head && d->skip_masked == 1 && d->window == 8 && pl.max_cav > 1 && d->H % 8 == 0 && d->W % 8 == 0 &&
        !HMVIT_ENV("HMVIT_NO_PRUNE")) {
        AttnParams ap;
        memset(&ap, 0, sizeof(ap));
        ap.ainv = ainv; ap.B = B; ap.L = L; ap.n_ego = pl.max_cav; ap.H = d->H; ap.W = d->W; ap.window = d->window;
        for (int i = 0; i < pl.n_slots; ++i) ap.cav[i] = (int8_t)(d->cav_mask[i] != 0);
        const size_t nb = (size_t)B * pl.max_cav * (P / 64);
        need_last = reinterpret_cast<unsigned char*>(ws + pl.off_need);
        HMVIT_CHECK_HIP(hipMemsetAsync(need_last, 0, 2 * nb, st));
        HMVIT_TRY(launch_window_need(ap, nullptr, need_last, st));
        if (d->num_iters >= 2 && !HMVIT_ENV("HMVIT_NO_PRUNE_PREV")) {
            need_prev = need_last + nb;
            HMVIT_TRY(launch_window_need(ap, need_last, need_prev, st));
        }
    }

    // World-ordered item list of the local stages (launch_attn_schedule): one list per forward, for the launches that run all
    // max_cav egos through a persistent kernel
    const int* sched = nullptr;
    int sched_sub = 2;
    if (const char* e = HMVIT_ENV("HMVIT_SCHED_SUB")) sched_sub = atoi(e);
    if (d->window == 8 && C > 64 && pl.max_cav > 1 && pl.max_cav <= 8 && d->skip_masked && sched_sub > 0 && d->H % 8 == 0 && d->W % 8 == 0 &&
        B <= 128 && d->H / 8 <= 1024 && d->W / 8 <= 1024) {   // the packed item word's field widths (launch_attn_schedule); larger scenes keep the tile order
        AttnParams ap;
        memset(&ap, 0, sizeof(ap));
        ap.ainv = ainv; ap.B = B; ap.L = L; ap.n_ego = pl.max_cav; ap.H = d->H; ap.W = d->W;
        int* ws_s = reinterpret_cast<int*>(ws + pl.off_sched);
        HMVIT_TRY(launch_attn_schedule(ap, ws_s, st));
        sched = ws_s;
    }

    // k_attention_patch16's tables: one set per forward (they depend on the pair transforms and the window, not on the stage)
    const void* patch_tab = nullptr;
    if (pl.off_ptab && d->self_identity && pl.max_cav <= 5 && pl.n_slots * L <= 128) {
        AttnParams ap;
        memset(&ap, 0, sizeof(ap));
        ap.ainv = ainv; ap.B = B; ap.L = L; ap.n_ego = pl.max_cav; ap.n_src = pl.max_cav; ap.H = d->H; ap.W = d->W; ap.window = d->window;
        for (int i = 0; i < pl.n_slots; ++i) ap.cav[i] = (int8_t)(d->cav_mask[i] != 0);
        HMVIT_TRY(launch_patch16_tables(ap, ws + pl.off_ptab, st));
        patch_tab = ws + pl.off_ptab;
    }

    bool qkv_done = false;   // this stage's Q / K' / V' were produced by the previous stage's fused tail
    bool head_done = false;  // mlp_head rode on the last stage's tail
    for (int it = 0; it < d->num_iters; ++it) {
        for (int s = 0; s < 2; ++s) {
            const HmvitStageWeights& wt = d->stage[s];
            const bool first = it == 0 && s == 0;
            const StageInfo si = stage_info(it, s);
            const bool last = si.last;
            float* x_out = par ? reinterpret_cast<float*>(ws + (s == 0 ? pl.off_xa : pl.off_xb)) : xs;
            const int n_ego = si.n_ego, E = si.E;
            const int n_src = pl.max_cav;
            const int* e_of_type = si.e_of_type;

            // 1+2. LayerNorm + Q / folded K,V projections, activations in registers
            if (!qkv_done) {
                QkvBatcher qb_;
                memset(&qb_.p, 0, sizeof(qb_.p));
                qb_.n = 0; qb_.C = C; qb_.st = st; qb_.split = split;
                qb_.p.gamma = wt.ln_gamma; qb_.p.beta = wt.ln_beta; qb_.p.P = P; qb_.p.in_nchw = first ? 1 : 0;
                // first stage: pulled tiles; the counters are the padding words of the affine records, zeroed by k_pair_affines
                // a moment ago (one per launch)
                if (first) { qb_.pull_base = reinterpret_cast<int*>(ainv) + 7; qb_.pull_left = pl.n_slots * L; }
                for (int b = 0; b < B; ++b)
                    for (int l = 0; l < pl.max_cav; ++l) HMVIT_TRY(qb_.add(qkv_job(wt, si, b * L + l, l, first)));
                HMVIT_TRY(qb_.flush());
            }
            HMVIT_MARK_IF(HMVIT_PHASE_QKV, !qkv_done);     // a stage whose projections rode on the previous tail launches nothing here
            qkv_done = false;

            // 3. fused warp + partition + attention
            unsigned char* need = nullptr;
            int* tail_pull = nullptr;              // zeroed pull counters for this stage's tail launches (k_tile_vis), one per launch
            int n_tail_pull = 0;
            {
                AttnParams ap;
                memset(&ap, 0, sizeof(ap));
                ap.q = qb; ap.kv = kvb; ap.b_q = wt.b_q; ap.b_kv = wt.b_kv; ap.bias_frag = wt.bias_frag;
                ap.ainv = ainv; ap.out = ob;
                ap.B = B; ap.L = L; ap.n_ego = n_ego; ap.n_src = n_src; ap.E = E; ap.C = C; ap.H = d->H; ap.W = d->W;
                ap.window = d->window; ap.partition = s == 0 ? HMVIT_PART_WINDOW : HMVIT_PART_GRID;
                ap.skip_masked = d->skip_masked;
                ap.k_logit = (split == 1 && wt.scales) ? wt.scales->k_logit : 1.f;   // f32 planes at their own power of two
                for (int i = 0; i < pl.n_slots; ++i) {
                    ap.mode[i] = (int8_t)d->mode[i];
                    ap.cav[i] = (int8_t)(d->cav_mask[i] != 0);
                    ap.ego_e[i] = (int8_t)(e_of_type[d->mode[i]] < 0 ? 0 : e_of_type[d->mode[i]]);
                }
                // the stage after this one is the pruned last stage (ego 0 only): unreachable windows are dead code
                need = (it == d->num_iters - 1 && s == 0) ? need_last : nullptr;
                ap.self_identity = d->self_identity;
                ap.rigid_patch = d->rigid_patch;
                ap.patch_tab = (n_ego == pl.max_cav) ? patch_tab : nullptr;
                // split mode: the persistent split kernel needs the table and identity self transforms; otherwise (and for
                // window 4 / C = 64) the exact-f32 kernel runs on the f32 planes
                const bool pc_split = split == 1 && d->self_identity && pl.n_slots * L <= 128;
                if ((split != 1 || pc_split) && d->skip_masked && d->window == 8 && C > 64 && n_src <= 8 && !HMVIT_ENV("HMVIT_ATTN_DEBUG")) {
                    // tiles without a visible key are skipped by the persistent kernel (launch_tile_vis)
                    unsigned* vis = reinterpret_cast<unsigned*>(ws + pl.off_vis);
                    // dynamic item assignment of the split kernel (16 pull counters in the slack behind the table, zeroed by
                    // k_tile_vis) - for the dilated-grid stage over all egos only, its longest launch: 2.53 -> 2.39 ms there (the
                    // static shares finish 8 % of the launch apart, pulled ones 4 %).  The other launches lose 2 - 10 % with it: their
                    // items are short (2 - 6 steps instead of 10) and in the pruned stage a third of the sequence positions are
                    // holes, so the pull latency - one scalar atomic per position - shows
#ifndef HMVIT_EXP_STATIC_ITEMS
                    // (the ticket decode divides by multiply-high: exact while sequence length x divisor < 2^32, pcs2_magic)
                    const bool q_fits = (size_t)B * n_ego * (d->H / 8 + 4) * (d->W / 8 + 8) < ((size_t)1 << 19) && d->H / 8 < 4096 && d->W / 8 < 4096;
#ifdef HMVIT_EXP_DYN_ALL
                    const bool q_stage = true;
#else
                    const bool q_stage = (s == 1 && n_ego > 1 && !HMVIT_ENV("HMVIT_PCS_STATIC")) || HMVIT_ENV("HMVIT_PCS_DYN_ALL");
#endif
                    if (pc_split && q_fits && q_stage)
                        ap.queue = reinterpret_cast<int*>(vis + (size_t)pl.n_slots * pl.P / 64);
#endif
                    if (split && C == 256 && !HMVIT_ENV("HMVIT_X16_STATIC"))
                        ap.tail_pull = tail_pull = reinterpret_cast<int*>(vis + (size_t)pl.n_slots * pl.P / 64) + 16;
                    HMVIT_TRY(launch_tile_vis(ap, vis, need, st));
                    ap.vis_mask = vis;
                    ap.prune = need != nullptr;
                    if (s == 0 && sched && n_ego == pl.max_cav) {   // local stage over all egos: world-ordered items
                        ap.sched = sched;
                        ap.n_sched = B * n_ego * (d->H / 8) * (d->W / 8);
                        ap.sched_sub = sched_sub;
                    }
                }
                if (timer && timer->d_items && timer->n_stages < 16) {      // profile: (ego, window) items this launch runs
                    const int n_items = B * n_ego * (d->H / d->window) * (d->W / d->window);
                    timer->total_items[timer->n_stages] = n_items;
                    if (ap.vis_mask && ap.prune) HMVIT_TRY(launch_count_live(ap.vis_mask, n_items, timer->d_items + timer->n_stages, st));
                    else HMVIT_CHECK_HIP(hipMemcpyAsync(timer->d_items + timer->n_stages, &timer->total_items[timer->n_stages], sizeof(int), hipMemcpyHostToDevice, st));
                    ++timer->n_stages;
                }
                HMVIT_TRY(launch_attention(ap, split == 1 ? HMVIT_PREC_SPLIT : HMVIT_PREC_F16, st));
            }
            HMVIT_MARK(HMVIT_PHASE_ATTENTION);

            // 4+5. output projection + residual + LayerNorm + FFN + residual, in place on xs; in the sequential
            // block the next stage's LayerNorm + projections ride on the same kernel (k_out_ffn_qkv)
            {
                const bool has_next = !(it == d->num_iters - 1 && s == 1);
                const bool fuse = !par && has_next && !last && C == 256 && !HMVIT_ENV("HMVIT_NO_FUSE");
                FfnBatcher fb;
                memset(&fb.p, 0, sizeof(fb.p));
                fb.n = 0; fb.C = C; fb.st = st; fb.split = split;
                fb.p.w_o = reinterpret_cast<const half_t*>(wt.img_o); fb.p.b_o = wt.b_o;
                fb.p.ln_g = wt.ffn_ln_gamma; fb.p.ln_b = wt.ffn_ln_beta;
                fb.p.w_ffn = reinterpret_cast<const half_t*>(wt.img_ffn); fb.p.b_1 = wt.b_1; fb.p.b_2 = wt.b_2;
                fb.p.P = P; fb.p.W = d->W;
                set_ffn_scales(fb.p, split ? wt.scales : nullptr, split ? d->head_scales : nullptr);
                auto next_pull = [&]() -> int* { return (tail_pull && n_tail_pull < 48) ? tail_pull + n_tail_pull++ : nullptr; };
                fb.variant = FFN_FULL;
                if (fuse) {
                    const int it2 = s == 1 ? it + 1 : it, s2 = 1 - s;
                    const HmvitStageWeights& wn = d->stage[s2];
                    const StageInfo sn = stage_info(it2, s2);
                    QkvParams qp;
                    memset(&qp, 0, sizeof(qp));
                    qp.gamma = wn.ln_gamma; qp.beta = wn.ln_beta; qp.P = P; qp.in_nchw = 0;
                    int n = 0;
                    for (int b = 0; b < B; ++b)
                        for (int i = 0; i < n_ego; ++i) {   // n_ego == max_cav here: every source agent of the next stage
                            const int slot = b * L + i;
                            FfnJob j;
                            j.need = nullptr; j.x_nchw = 0;
                            j.o = ob + (size_t)slot * map_bytes;
                            j.x = xs + (size_t)slot * map_elems;
                            if (first && direct_x) { j.x = d->x + (size_t)slot * map_elems; j.x_nchw = 1; }
                            j.out = x_out + (size_t)slot * map_elems;
                            j.type = d->mode[slot];
                            j.pad = (sn.last && i >= sn.n_ego) ? 1 : 0;   // x'' of a pure K/V source of the pruned stage is never read
                            const unsigned char* nd = need ? need : (it == d->num_iters - 2 && s == 1) ? need_prev : nullptr;
                            j.need = nd ? nd + (size_t)(b * n_ego + i) * (P / 64) : nullptr;
                            fb.p.job[n] = j;
                            qp.job[n] = qkv_job(wn, sn, slot, i, false);
                            if (++n == kMaxChainJobs) { fb.p.pull = next_pull(); HMVIT_TRY(launch_out_ffn_qkv(fb.p, qp, n, C, split, st)); n = 0; }
                        }
                    fb.p.pull = next_pull();
                    HMVIT_TRY(launch_out_ffn_qkv(fb.p, qp, n, C, split, st));
                    qkv_done = true;
                } else if (last && !par && C == 256 && d->head_img_ffn && !HMVIT_ENV("HMVIT_NO_FUSE")) {
                    // last stage: only the ego row is alive and mlp_head follows immediately (k_out_ffn_head)
                    fb.p.w_head = reinterpret_cast<const half_t*>(d->head_img_ffn); fb.p.hb_1 = d->head_b1; fb.p.hb_2 = d->head_b2;
                    int n = 0;
                    for (int b = 0; b < B; ++b) {
                        const int slot = b * L;
                        FfnJob j;
                            j.need = nullptr; j.x_nchw = 0;
                        j.o = ob + (size_t)slot * map_bytes;
                        j.x = xs + (size_t)slot * map_elems;
                        j.out = d->out + (size_t)b * map_elems;
                        j.type = d->mode[slot]; j.pad = 0;
                        fb.p.job[n] = j;
                        if (++n == kMaxChainJobs) { fb.p.pull = next_pull(); HMVIT_TRY(launch_out_ffn_head(fb.p, n, C, split, st)); n = 0; }
                    }
                    fb.p.pull = next_pull();
                    HMVIT_TRY(launch_out_ffn_head(fb.p, n, C, split, st));
                    head_done = true;
                } else {
                    for (int b = 0; b < B; ++b)
                        for (int i = 0; i < n_ego; ++i) {
                            const int slot = b * L + i;
                            FfnJob j;
                            j.need = nullptr; j.x_nchw = 0;
                            j.o = ob + (size_t)slot * map_bytes;
                            j.x = xs + (size_t)slot * map_elems;
                            j.out = x_out + (size_t)slot * map_elems;
                            j.type = d->mode[slot]; j.pad = 0;
                            HMVIT_TRY(fb.add(j));
                        }
                    HMVIT_TRY(fb.flush());
                }
                if (!last && n_ego < L) {
                    fb.variant = FFN_NO_ATTN;   // agents without an attention update (padding)
                    for (int b = 0; b < B; ++b)
                        for (int l = n_ego; l < L; ++l) {
                            const int slot = b * L + l;
                            if (d->apply_head) continue;   // never consumed by HeteroFusion
                            FfnJob j;
                            j.need = nullptr; j.x_nchw = 0;
                            j.o = nullptr;
                            j.x = xs + (size_t)slot * map_elems;
                            j.out = x_out + (size_t)slot * map_elems;
                            j.type = d->mode[slot]; j.pad = 0;
                            HMVIT_TRY(fb.add(j));
                        }
                    HMVIT_TRY(fb.flush());
                }
            }
            // the last stage's tail with mlp_head appended (k_out_ffn_head) is the HEAD phase: one launch, ego rows only
            HMVIT_MARK(head_done ? HMVIT_PHASE_HEAD : HMVIT_PHASE_FFN2);
        }
        if (d->parallel) {
            HMVIT_TRY(merge_branches(d, pl, d->apply_head && it == d->num_iters - 1, st));
            HMVIT_MARK(HMVIT_PHASE_LAYOUT_OUT);
        }
    }

    if (!d->apply_head) {
        HMVIT_TRY(launch_transpose(xs, d->out, pl.n_slots, P, C, st));
        HMVIT_MARK(HMVIT_PHASE_LAYOUT_OUT);
        return HMVIT_OK;
    }
    HMVIT_CHECK_ARG(d->head_img_ffn && d->head_b1 && d->head_b2, "f16 mode: mlp_head image / biases are null");
    if (!head_done) {
        FfnBatcher fb;
        memset(&fb.p, 0, sizeof(fb.p));
        fb.n = 0; fb.C = C; fb.st = st; fb.variant = FFN_HEAD_NCHW; fb.split = split;
        fb.p.w_ffn = reinterpret_cast<const half_t*>(d->head_img_ffn); fb.p.b_1 = d->head_b1; fb.p.b_2 = d->head_b2;
        fb.p.P = P;
        set_ffn_scales(fb.p, nullptr, split ? d->head_scales : nullptr);
        for (int b = 0; b < B; ++b) {
            FfnJob j;
                            j.need = nullptr; j.x_nchw = 0;
            j.o = nullptr;
            j.x = xs + (size_t)(b * L) * map_elems;
            j.out = d->out + (size_t)b * map_elems;
            j.type = d->mode[b * L]; j.pad = 0;
            HMVIT_TRY(fb.add(j));
        }
        HMVIT_TRY(fb.flush());
    }
    HMVIT_MARK_IF(HMVIT_PHASE_HEAD, !head_done);
    return HMVIT_OK;
}

}  // namespace hmvit

using namespace hmvit;

extern "C" {

int hmvit_abi_version(void) { return HMVIT_ABI_VERSION; }

const char* hmvit_last_error(void) { return hmvit::g_err; }

size_t hmvit_fusion_workspace_bytes(const HmvitFusionDesc* desc) {
    if (check_desc(desc) != HMVIT_OK) return 0;
    Plan pl;
    make_plan(desc, pl);
    return pl.total;
}

int hmvit_fusion_forward(const HmvitFusionDesc* desc, void* stream) {
    return fusion_forward(desc, reinterpret_cast<hipStream_t>(stream), nullptr);
}

// (ego, window) attention items per stage of the last hmvit_fusion_profile of this thread: run / in the stage
static thread_local int g_prof_items[2][16];
static thread_local int g_prof_stages = 0;
int hmvit_fusion_profile_items(int32_t* live, int32_t* total, int capacity) {
    HMVIT_CHECK_ARG(live && total && capacity > 0, "fusion_profile_items: bad argument");
    const int n = g_prof_stages < capacity ? g_prof_stages : capacity;
    for (int i = 0; i < n; ++i) { live[i] = g_prof_items[0][i]; total[i] = g_prof_items[1][i]; }
    return n;
}

int hmvit_fusion_profile(const HmvitFusionDesc* desc, void* stream, float* phase_ms, int32_t* phase_launches) {
    HMVIT_CHECK_ARG(phase_ms && phase_launches, "fusion_profile: null output");
    PhaseTimer timer;
    timer.st = reinterpret_cast<hipStream_t>(stream);
    HMVIT_CHECK_HIP(hipMalloc(&timer.d_items, 16 * sizeof(int)));
    HMVIT_CHECK_HIP(hipMemsetAsync(timer.d_items, 0, 16 * sizeof(int), timer.st));
    int rc = fusion_forward(desc, timer.st, &timer);
    if (rc != HMVIT_OK) return rc;
    HMVIT_CHECK_HIP(hipStreamSynchronize(timer.st));
    g_prof_stages = timer.n_stages;
    if (timer.n_stages > 0) HMVIT_CHECK_HIP(hipMemcpy(g_prof_items[0], timer.d_items, timer.n_stages * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < timer.n_stages; ++i) g_prof_items[1][i] = timer.total_items[i];
    for (int i = 0; i < HMVIT_NUM_PHASES; ++i) { phase_ms[i] = 0.f; phase_launches[i] = 0; }
    for (size_t i = 1; i < timer.ev.size(); ++i) {
        float ms = 0.f;
        HMVIT_CHECK_HIP(hipEventElapsedTime(&ms, timer.ev[i - 1], timer.ev[i]));
        phase_ms[timer.phase[i]] += ms;
        phase_launches[timer.phase[i]] += timer.launched[i];
    }
    return HMVIT_OK;
}

int hmvit_pack_small(const void* mode, int mode_dtype, int n_mode, const void* record_len, int rl_dtype, int n_rl, const void* mask,
                     int mask_dtype, int n_mask, const void* pairwise, int pw_dtype, int B, int L, int64_t* out, void* stream) {
    HMVIT_CHECK_ARG(mode && record_len && mask && out && n_mode >= 0 && n_rl >= 0 && n_mask >= 0, "pack_small: bad argument");
    for (int dt : {mode_dtype, rl_dtype, mask_dtype}) HMVIT_CHECK_ARG(dt >= 0 && dt <= 5, "pack_small: dtype code %d (0..5)", dt);
    HMVIT_CHECK_ARG(!pairwise || ((pw_dtype == 0 || pw_dtype == 1) && B > 0 && L > 0), "pack_small: pairwise dtype %d / B=%d L=%d", pw_dtype, B, L);
    SmallPack a;
    a.src[0] = mode; a.src[1] = record_len; a.src[2] = mask;
    a.dtype[0] = mode_dtype; a.dtype[1] = rl_dtype; a.dtype[2] = mask_dtype;
    a.n[0] = n_mode; a.n[1] = n_rl; a.n[2] = n_mask;
    a.pairwise = pairwise; a.pw_dtype = pw_dtype; a.B = B; a.L = L;
    a.out = reinterpret_cast<long long*>(out);
    return launch_pack_small(a, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_nchw_to_tokens(const float* x, float* y, int n_agents, int C, int P, void* stream) {
    HMVIT_CHECK_ARG(x && y && n_agents > 0 && C > 0 && P > 0, "nchw_to_tokens: bad argument");
    return launch_transpose(x, y, n_agents, C, P, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_tokens_to_nchw(const float* x, float* y, int n_agents, int C, int P, void* stream) {
    HMVIT_CHECK_ARG(x && y && n_agents > 0 && C > 0 && P > 0, "tokens_to_nchw: bad argument");
    return launch_transpose(x, y, n_agents, P, C, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_layernorm(const float* x, void* y, const int32_t* types, const float* gamma, const float* beta,
                    int n_agents, int P, int C, int precision, void* stream) {
    HMVIT_CHECK_ARG(x && y && types && gamma && beta, "layernorm: null pointer");
    HMVIT_CHECK_ARG(n_agents > 0 && n_agents <= kMaxSlots, "layernorm: n_agents=%d out of (0, %d]", n_agents, kMaxSlots);
    AgentTypes t;
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < n_agents; ++i) {
        HMVIT_CHECK_ARG(types[i] >= 0 && types[i] < HMVIT_NUM_TYPES, "layernorm: types[%d]=%d", i, types[i]);
        t.t[i] = (int8_t)types[i];
    }
    return launch_layernorm(x, y, gamma, beta, t, n_agents, P, C, precision, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_linear(const void* a, const void* w, const float* bias, const float* residual, void* y, int M, int N,
                 int K, int gelu, int out_f32, int precision, void* stream) {
    HMVIT_CHECK_ARG(a && w && y && M > 0 && N > 0 && K > 0, "linear: bad argument");
    GemmJobs jobs;
    jobs.n = 1;
    GemmJob& j = jobs.j[0];
    j.a = a; j.w = w; j.bias = bias; j.residual = residual; j.y = y;
    j.M = M; j.N = N; j.K = K; j.n_per_plane = N; j.plane_stride = 0;
    return launch_gemm(jobs, precision == HMVIT_PREC_SPLIT, gelu != 0, out_f32 != 0, precision, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_pair_affines(const float* pairwise_t, float* ainv, int n, int H, int W, float discrete_ratio,
                       float downsample_rate, void* stream) {
    HMVIT_CHECK_ARG(pairwise_t && ainv && n > 0 && H > 0 && W > 0, "pair_affines: bad argument");
    HMVIT_CHECK_ARG(discrete_ratio * downsample_rate != 0.f, "pair_affines: zero scale");
    return launch_pair_affines(pairwise_t, ainv, n, H, W, discrete_ratio, downsample_rate,
                               reinterpret_cast<hipStream_t>(stream));
}

int hmvit_warp_affine(const float* src, const float* ainv, float* dst, float* roi, int n, int H, int W, int C,
                      void* stream) {
    HMVIT_CHECK_ARG(src && ainv && dst && roi && n > 0 && H > 0 && W > 0 && C > 0, "warp_affine: bad argument");
    return launch_warp(src, ainv, dst, roi, n, H, W, C, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_window_attention(const void* q, const void* kv, const float* b_q, const float* b_kv,
                           const float* bias_frag, const float* ainv, const int32_t* mode,
                           const int32_t* cav_mask, const int32_t* ego_e, void* out, int B, int L, int n_ego,
                           int n_src, int E, int C, int H, int W, int window, int partition, int precision,
                           int skip_masked, void* stream) {
    HMVIT_CHECK_ARG(q && kv && b_q && b_kv && bias_frag && ainv && mode && cav_mask && ego_e && out,
                    "window_attention: null pointer");
    HMVIT_CHECK_ARG(B > 0 && L > 0 && B * L <= kMaxSlots && n_ego <= L && n_src <= L && E >= 1 && E <= 2,
                    "window_attention: bad sizes");
    AttnParams ap;
    memset(&ap, 0, sizeof(ap));
    ap.q = q; ap.kv = kv; ap.b_q = b_q; ap.b_kv = b_kv; ap.bias_frag = bias_frag; ap.ainv = ainv; ap.out = out;
    ap.B = B; ap.L = L; ap.n_ego = n_ego; ap.n_src = n_src; ap.E = E; ap.C = C; ap.H = H; ap.W = W;
    ap.window = window; ap.partition = partition; ap.skip_masked = skip_masked;
    for (int i = 0; i < B * L; ++i) {
        ap.mode[i] = (int8_t)mode[i];
        ap.cav[i] = (int8_t)(cav_mask[i] != 0);
        ap.ego_e[i] = (int8_t)ego_e[i];
    }
    return launch_attention(ap, precision, reinterpret_cast<hipStream_t>(stream));
}

// range information of the NEXT convolution call of this thread (HMVIT_PREC_SPLIT), consumed by that call
static thread_local const unsigned* g_next_x_absmax = nullptr;
static thread_local unsigned* g_next_y_absmax = nullptr;
static thread_local float g_next_w_absmax = 0.f;
static thread_local const void* g_next_w_image = nullptr;
static thread_local int g_next_w_image_kind = 0;
int hmvit_pfn_scatter(const float* voxels, const int32_t* coords, const int32_t* num_points, const float* w,
                      const float* shift, void* canvas, float* pillar_out, int n_pillars, int nx, int ny, int n_agents,
                      int32_t* oob_count, const float* voxel_size, const float* lidar_range, int precision, void* stream) {
    PfnParams p;
    // a y_absmax slot left by hmvit_conv_range is consumed here too (first: an early return must not leave it to the next call):
    // max of the scattered values, for the first convolution of the backbone
    p.canvas_absmax = canvas ? g_next_y_absmax : nullptr;
    g_next_x_absmax = nullptr; g_next_y_absmax = nullptr; g_next_w_absmax = 0.f; g_next_w_image = nullptr;
    HMVIT_CHECK_ARG(voxels && coords && num_points && w && shift && (canvas || pillar_out) && voxel_size && lidar_range,
                    "pfn_scatter: null pointer");
    HMVIT_CHECK_ARG(n_pillars >= 0 && nx > 0 && ny > 0 && n_agents > 0, "pfn_scatter: bad sizes");
    p.voxels = voxels; p.coords = coords; p.num_points = num_points; p.w = w; p.shift = shift;
    p.canvas = canvas; p.pillar_out = pillar_out; p.n_pillars = n_pillars; p.nx = nx; p.ny = ny;
    p.n_agents = n_agents; p.oob_count = oob_count;
    p.vx = voxel_size[0]; p.vy = voxel_size[1]; p.vz = voxel_size[2];
    p.x_off = voxel_size[0] / 2 + lidar_range[0];
    p.y_off = voxel_size[1] / 2 + lidar_range[1];
    p.z_off = voxel_size[2] / 2 + lidar_range[2];
    return launch_pfn_scatter(p, precision, reinterpret_cast<hipStream_t>(stream));
}

static void take_conv_range(ConvParams& p) {
    p.x_absmax = g_next_x_absmax; p.w_absmax = g_next_w_absmax; p.y_absmax = g_next_y_absmax; p.w_image = g_next_w_image; p.w_image_kind = g_next_w_image_kind;
    g_next_x_absmax = nullptr; g_next_y_absmax = nullptr; g_next_w_absmax = 0.f; g_next_w_image = nullptr;
}
size_t hmvit_conv3x3_image_bytes(int Cout, int Cin, int precision) {
    if (Cout <= 0 || Cin <= 0 || (precision != HMVIT_PREC_SPLIT && precision != HMVIT_PREC_F16)) return 0;
    if (Cin % (precision == HMVIT_PREC_SPLIT ? 32 : 64)) return 0;
    return conv3_image_size(Cout, Cin, precision);
}
int hmvit_conv3x3_image(const void* w, int Cout, int Cin, int precision, void* image, void* stream) {
    return launch_conv3_pack(w, Cout, Cin, precision, image, reinterpret_cast<hipStream_t>(stream));
}
size_t hmvit_conv_gemm_image_bytes(int Ncols, int Ktot) {
    if (Ncols <= 0 || Ktot <= 0 || Ktot % 32) return 0;
    return conv_gemm_image_size(Ncols, Ktot);
}
int hmvit_conv_gemm_image(const float* w, int Ncols, int Ktot, void* image, void* stream) {
    return launch_conv_gemm_pack(w, Ncols, Ktot, image, reinterpret_cast<hipStream_t>(stream));
}
int hmvit_conv_weight_image(const void* image, int kind) {
    HMVIT_CHECK_ARG(kind == 0 || kind == 1, "conv_weight_image: kind %d (0: 3 x 3 ring order, 1: GEMM order)", kind);
    g_next_w_image = image;
    g_next_w_image_kind = kind;
    return HMVIT_OK;
}
int hmvit_conv_range(const void* x_absmax, float w_absmax, void* y_absmax) {
    g_next_x_absmax = reinterpret_cast<const unsigned*>(x_absmax);
    g_next_w_absmax = w_absmax;
    g_next_y_absmax = reinterpret_cast<unsigned*>(y_absmax);
    return HMVIT_OK;
}
int hmvit_absmax(const float* x, size_t n, void* slot, void* stream) {
    HMVIT_CHECK_ARG(x && slot, "absmax: null pointer");
    return launch_absmax(x, n, reinterpret_cast<unsigned*>(slot), reinterpret_cast<hipStream_t>(stream));
}

int hmvit_conv2d(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                 int ksize, int stride, int pad, int relu, int y_ctot, int y_coff, int deconv_stride, int out_f32,
                 int precision, void* stream) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    take_conv_range(p);   // first: an early return must not leave the range of this call to the next one
    HMVIT_CHECK_ARG(x && w && y, "conv2d: null pointer");
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = ksize; p.stride = stride; p.pad = pad;
    p.relu = relu; p.y_ctot = y_ctot; p.y_coff = y_coff; p.deconv_s = deconv_stride; p.out_f32 = out_f32;
    p.res = nullptr; p.up2 = 0;
    if (deconv_stride) {
        p.Ho = H; p.Wo = W;
    } else {
        p.Ho = (H + 2 * pad - ksize) / stride + 1;
        p.Wo = (W + 2 * pad - ksize) / stride + 1;
    }
    HMVIT_CHECK_ARG(y_ctot >= y_coff + Cout, "conv2d: output channel window [%d, %d) exceeds %d", y_coff, y_coff + Cout, y_ctot);
    return launch_conv(p, precision, reinterpret_cast<hipStream_t>(stream));
}

/* Conv2d + folded BatchNorm with the two extras a ResNet / the up-sampling decoder need. */
int hmvit_conv2d_ex(const void* x, const void* w, const float* bias, const void* residual, void* y, int N, int H, int W, int Cin,
                    int Cout, int ksize, int stride, int pad, int relu, int upsample2, int out_f32, int precision, void* stream) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    take_conv_range(p);   // first: an early return must not leave the range of this call to the next one
    HMVIT_CHECK_ARG(x && w && y, "conv2d_ex: null pointer");
    HMVIT_CHECK_ARG(!(upsample2 & 1) || (H % 2 == 0 && W % 2 == 0), "conv2d_ex: upsampled size %dx%d must be even", H, W);
    HMVIT_CHECK_ARG(!(residual && out_f32 && precision == HMVIT_PREC_F16), "conv2d_ex: residual needs the precision's element type");
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = p.KW = ksize; p.stride = stride; p.pad = pad;
    p.relu = relu; p.y_ctot = Cout; p.y_coff = 0; p.deconv_s = 0; p.out_f32 = out_f32;
    p.res = residual; p.up2 = upsample2 & 1; p.no_patch = (upsample2 >> 1) & 1; p.force_patch = (upsample2 >> 2) & 1;
    p.Ho = (H + 2 * pad - ksize) / stride + 1;
    p.Wo = (W + 2 * pad - ksize) / stride + 1;
    return launch_conv(p, precision, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_conv2d_rowpack(const void* x, const void* w, const float* bias, void* y, int N, int Hp, int Wp, int Ho, int Wo, int Cout,
                         int krows, int stride, int relu, int precision, void* stream) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    take_conv_range(p);   // first: an early return must not leave the range of this call to the next one
    HMVIT_CHECK_ARG(x && w && y && N > 0 && Ho > 0 && Wo > 0 && Cout > 0 && krows > 0 && stride > 0, "conv2d_rowpack: bad argument");
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.N = N; p.H = Hp; p.W = Wp; p.Cin = 4; p.Cout = Cout; p.KH = krows; p.KW = 8; p.stride = stride; p.pad = 0;
    p.relu = relu; p.y_ctot = Cout; p.y_coff = 0; p.Ho = Ho; p.Wo = Wo; p.rowpack = 1;
    return launch_conv(p, precision, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_maxpool2d(const void* x, void* y, int N, int H, int W, int C, int ksize, int stride, int pad, int precision, void* stream) {
    HMVIT_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0, "maxpool2d: bad argument");
    return launch_maxpool(x, y, N, H, W, C, ksize, stride, pad, precision, reinterpret_cast<hipStream_t>(stream));
}

/* ---- detection post-processing (post.hip) ---- */

int hmvit_box_decode(const float* psm, const float* rm, const float* anchors, const float* transform, int H, int W, int A,
                     float score_threshold, int order_hwl, float* corners, float* scores, int32_t* index, int32_t* count,
                     int capacity, void* stream) {
    HMVIT_CHECK_ARG(psm && rm && anchors && corners && scores && index && count && H > 0 && W > 0 && A > 0 && capacity > 0,
                    "box_decode: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HMVIT_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), st));
    BoxDecodeParams p;
    p.psm = psm; p.rm = rm; p.anchors = anchors; p.T = transform; p.H = H; p.W = W; p.A = A; p.thresh = score_threshold;
    p.order_hwl = order_hwl; p.corners = corners; p.scores = scores; p.index = index; p.count = count; p.capacity = capacity;
    return launch_box_decode(p, st);
}

size_t hmvit_nms_workspace_bytes(int n) {
    if (n <= 0) return 256;
    const size_t K = n < 1000 ? n : 1000;
    return align_up((size_t)n * 4, 256) + align_up(K * 4, 256) + align_up(K * K * 4, 256);
}

int hmvit_nms_rotated(const float* corners, const float* scores, const int32_t* index, int n, float iou_threshold,
                      const float* range_xy, void* workspace, size_t workspace_bytes, int32_t* keep, int32_t* n_keep,
                      void* stream) {
    HMVIT_CHECK_ARG(n >= 0 && range_xy && keep && n_keep, "nms_rotated: bad argument");
    HMVIT_CHECK_ARG(n == 0 || (corners && scores && workspace), "nms_rotated: null buffer");
    HMVIT_CHECK_ARG(workspace_bytes >= hmvit_nms_workspace_bytes(n), "nms_rotated: workspace too small");
    const size_t K = n < 1000 ? n : 1000;
    char* ws = reinterpret_cast<char*>(workspace);
    int* rank = reinterpret_cast<int*>(ws);
    int* sorted = reinterpret_cast<int*>(ws + align_up((size_t)n * 4, 256));
    float* iou = reinterpret_cast<float*>(ws + align_up((size_t)n * 4, 256) + align_up(K * 4, 256));
    return launch_nms_rotated(corners, scores, index, n, iou_threshold, range_xy, rank, sorted, iou, keep, n_keep,
                              reinterpret_cast<hipStream_t>(stream));
}

int hmvit_quad_iou(const float* a, const float* b, int na, int nb, int stride_box, int stride_pt, float* iou, void* stream) {
    HMVIT_CHECK_ARG(na >= 0 && nb >= 0 && stride_box >= 4 * stride_pt && stride_pt >= 2, "quad_iou: bad argument");
    HMVIT_CHECK_ARG(na == 0 || nb == 0 || (a && b && iou), "quad_iou: null buffer");
    return launch_quad_iou(a, b, na, nb, stride_box, stride_pt, iou, reinterpret_cast<hipStream_t>(stream));
}

/* ---- pillariser (vox.hip) ---- */

size_t hmvit_voxelize_workspace_bytes(int n_points, int nx, int ny, int nz) {
    const size_t cells = (size_t)nx * ny * nz, n = n_points > 0 ? n_points : 1;
    return 3 * align_up(n * 4, 256) + 4 * align_up(cells * 4, 256);
}

int hmvit_voxelize(const float* points, int n_points, const float* voxel_size, const float* lidar_range, int max_points,
                   int max_voxels, void* workspace, size_t workspace_bytes, float* voxels, int32_t* coords, int32_t* num_points,
                   int32_t* n_voxels, void* stream) {
    HMVIT_CHECK_ARG(n_points >= 0 && voxel_size && lidar_range && max_points > 0 && max_voxels > 0 && workspace && voxels &&
                    coords && num_points && n_voxels && (n_points == 0 || points), "voxelize: bad argument");
    VoxParams p;
    p.points = points; p.n_points = n_points;
    int grid[3];
    for (int k = 0; k < 3; ++k) {
        p.rmin[k] = lidar_range[k]; p.vsize[k] = voxel_size[k];
        grid[k] = (int)lroundf((lidar_range[3 + k] - lidar_range[k]) / voxel_size[k]);   // sp_voxel_preprocessor.py:28-30
        HMVIT_CHECK_ARG(grid[k] > 0, "voxelize: empty grid on axis %d", k);
    }
    p.nx = grid[0]; p.ny = grid[1]; p.nz = grid[2]; p.max_points = max_points; p.max_voxels = max_voxels;
    HMVIT_CHECK_ARG(workspace_bytes >= hmvit_voxelize_workspace_bytes(n_points, p.nx, p.ny, p.nz), "voxelize: workspace too small");
    const size_t cells = (size_t)p.nx * p.ny * p.nz, n = n_points > 0 ? n_points : 1;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t sn = align_up(n * 4, 256), sc = align_up(cells * 4, 256);
    p.cell = reinterpret_cast<int*>(ws); p.placed = reinterpret_cast<int*>(ws + sn); p.scan = reinterpret_cast<int*>(ws + 2 * sn);
    p.first = reinterpret_cast<int*>(ws + 3 * sn); p.count = reinterpret_cast<int*>(ws + 3 * sn + sc);
    p.cmin = reinterpret_cast<int*>(ws + 3 * sn + 2 * sc); p.vox_id = reinterpret_cast<int*>(ws + 3 * sn + 3 * sc);
    p.voxels = voxels; p.coords = coords; p.num_points = num_points; p.n_voxels = n_voxels;
    return launch_voxelize(p, reinterpret_cast<hipStream_t>(stream));
}

/* ---- camera -> BEV lift (cvt.hip) ---- */

int hmvit_cvt_embed(int mode, const float* I_inv, const float* E_inv, const float* grid, const float* w_in, const float* w_bias,
                    const float* w_cam, const float* x, float* out, int n_agents, int n_cam, int H, int W, int dim,
                    float image_w, float image_h, void* stream) {
    HMVIT_CHECK_ARG((mode == 0 || mode == 1) && E_inv && w_in && w_cam && out && n_agents > 0 && n_cam > 0 && H > 1 && W > 1,
                    "cvt_embed: bad argument");
    HMVIT_CHECK_ARG(mode == 1 ? grid != nullptr : I_inv != nullptr, "cvt_embed: missing %s", mode ? "grid" : "I_inv");
    CvtEmbedParams p;
    p.mode = mode; p.bn = n_agents * n_cam; p.n_cam = n_cam; p.P = H * W; p.H = H; p.W = W; p.dim = dim;
    p.img_w = image_w; p.img_h = image_h; p.I_inv = I_inv; p.E_inv = E_inv; p.grid = grid; p.w_in = w_in; p.w_bias = w_bias;
    p.w_cam = w_cam; p.x = x; p.out = out;
    return launch_cvt_embed(p, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_bn_relu_tokens(const float* x, const float* scale, const float* shift, float* y, int n, int C, int P, void* stream) {
    HMVIT_CHECK_ARG(x && scale && shift && y && n > 0 && C > 0 && P > 0, "bn_relu_tokens: bad argument");
    return launch_bn_relu_tokens(x, scale, shift, y, n, C, P, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_cross_attention(const void* q, const void* k, const void* v, float* out, int n_agents, int n_cam, int Q, int K,
                          int heads, int dim_head, int precision, void* stream) {
    HMVIT_CHECK_ARG(q && k && v && out && n_agents > 0 && n_cam > 0 && Q > 0 && K > 0 && heads > 0, "cross_attention: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (precision == HMVIT_PREC_F16)
        return launch_cross_attention_f16(reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                                          reinterpret_cast<const half_t*>(v), out, n_agents, n_cam, Q, K, heads, dim_head, st);
    // split mode: fp32-class products on the f16 matrix pipes where the tile sizes allow; the exact-f32 MFMA kernel otherwise
    if (precision == HMVIT_PREC_SPLIT && dim_head == 32 && Q % 64 == 0 && K % 64 == 0)
        return launch_cross_attention_split(reinterpret_cast<const float*>(q), reinterpret_cast<const float*>(k),
                                            reinterpret_cast<const float*>(v), out, n_agents, n_cam, Q, K, heads, dim_head, st);
    return launch_cross_attention(reinterpret_cast<const float*>(q), reinterpret_cast<const float*>(k),
                                  reinterpret_cast<const float*>(v), out, n_agents, n_cam, Q, K, heads, dim_head, nullptr, st);
}

/* ---- training-mode operators of the camera lift (cvt_modules.py:95-165 under autograd; hm-vit_amd/camera_train.py) ---- */
int hmvit_cross_attention_train(const float* q, const float* k, const float* v, float* out, float* lse, int n_agents, int n_cam, int Q,
                                int K, int heads, int dim_head, void* stream) {
    HMVIT_CHECK_ARG(q && k && v && out && lse && n_agents > 0 && n_cam > 0 && Q > 0 && K > 0 && heads > 0, "cross_attention_train: bad argument");
    return launch_cross_attention(q, k, v, out, n_agents, n_cam, Q, K, heads, dim_head, nullptr, reinterpret_cast<hipStream_t>(stream), lse);
}
int hmvit_cross_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* d_out,
                                   float* dq, float* dk, float* dv, int n_agents, int n_cam, int Q, int K, int heads, int dim_head,
                                   void* stream) {
    HMVIT_CHECK_ARG(q && k && v && out && lse && d_out && dq && dk && dv && n_agents > 0 && n_cam > 0, "cross_attention_backward: bad argument");
    return launch_cross_attention_bwd(q, k, v, out, lse, d_out, dq, dk, dv, n_agents, n_cam, Q, K, heads, dim_head,
                                      reinterpret_cast<hipStream_t>(stream));
}
int hmvit_layernorm_backward(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int M, int C,
                             void* stream) {
    HMVIT_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && M > 0, "layernorm_backward: bad argument");
    AgentTypes ty;
    memset(&ty, 0, sizeof(ty));
    return launch_layernorm_bwd(x, dy, gamma, ty, 1, nullptr, dx, dgamma, dbeta, M, C, reinterpret_cast<hipStream_t>(stream));
}
int hmvit_gelu(const float* pre, float* y, size_t n, void* stream) {
    HMVIT_CHECK_ARG(pre && y, "gelu: null pointer");
    DropCfg none = {0ull, 0u, 0.f};
    return launch_gelu_drop(pre, y, n, none, reinterpret_cast<hipStream_t>(stream));
}
int hmvit_gelu_backward(const float* pre, const float* dy, float* dx, size_t n, void* stream) {
    HMVIT_CHECK_ARG(pre && dy && dx, "gelu_backward: null pointer");
    DropCfg none = {0ull, 0u, 0.f};
    return launch_gelu_bwd(pre, dy, dx, n, none, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_attention_bias(const float* q, const float* k, const float* v, const float* bias, float* out, int batch, int Q, int K,
                         int heads, int dim_head, void* stream) {
    HMVIT_CHECK_ARG(q && k && v && bias && out && batch > 0 && Q > 0 && K > 0 && heads > 0, "attention_bias: bad argument");
    return launch_cross_attention(q, k, v, out, batch, 1, Q, K, heads, dim_head, bias, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_attention_bias_train(const float* q, const float* k, const float* v, const float* bias, float* out, float* lse, int batch,
                               int Q, int K, int heads, int dim_head, void* stream) {
    HMVIT_CHECK_ARG(q && k && v && bias && out && lse && batch > 0 && Q > 0 && K > 0 && heads > 0, "attention_bias_train: bad argument");
    return launch_cross_attention(q, k, v, out, batch, 1, Q, K, heads, dim_head, bias, reinterpret_cast<hipStream_t>(stream), lse);
}
int hmvit_attention_bias_backward(const float* q, const float* k, const float* v, const float* bias, const float* out, const float* lse,
                                  const float* d_out, float* dq, float* dk, float* dv, float* d_bias, int batch, int Q, int K, int heads,
                                  int dim_head, void* stream) {
    HMVIT_CHECK_ARG(q && k && v && bias && out && lse && d_out && dq && dk && dv && d_bias && batch > 0 && Q > 0 && K > 0 && heads > 0,
                    "attention_bias_backward: bad argument");
    return launch_cross_attention_bwd(q, k, v, out, lse, d_out, dq, dk, dv, batch, 1, Q, K, heads, dim_head,
                                      reinterpret_cast<hipStream_t>(stream), bias, d_bias);
}
int hmvit_maxpool2d_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int ksize, int stride, int pad,
                             void* stream) {
    HMVIT_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2d_backward: bad argument");
    return launch_maxpool_bwd(x, dy, dx, N, H, W, C, ksize, stride, pad, reinterpret_cast<hipStream_t>(stream));
}

int hmvit_debug_tr16(uint16_t* out, void* stream) {
    HMVIT_CHECK_ARG(out != nullptr, "debug_tr16: null pointer");
    return launch_debug_tr16(out, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
