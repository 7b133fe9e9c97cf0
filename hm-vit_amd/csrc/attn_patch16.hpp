// k_attention_patch16 (round 6): the patch kernel at FOUR wavefronts per SIMD.  Included by attn.hip behind attn_patch.hpp (whose
// request / blend layout, tables and conventions it shares).
//
// k_attention_patch showed that the split attention is bound by how many wavefronts a SIMD holds, and that the two
// things which cap it at two are the registers of a wave that owns a head's 64 queries AND all of a half chunk's operand fragments
// (245), and the LDS of two f32 planes of patch per head (128 KB).  Here a window is worked on by SIXTEEN waves of <= 128 registers:
//   * wave = (head, key tile kt): it owns the 16 keys kt 16 .. kt 16 + 15 of every half chunk - its own patch (the <= 33 source pixels
//     those keys touch: 40 rows of 128 bytes), its own tables, its own online-softmax state over ITS keys for all 64 queries.  The two
//     waves of a head never synchronise inside an item; they merge their (maximum, sum, O) states once, at the item's end, through
//     their patch buffers (each finishes two query tiles);
//   * ONE plane of patch per wave (5 KB): K' is blended into registers, then the V' rows are requested into the same bytes and land
//     during the logits + softmax work; V' is blended (staging tile in the same bytes), then the next step's K' rows are requested
//     and land during the O products;
//   * the relative-position bias comes from a 15 x 16 table per head in LDS (7.5 KB for all heads) instead of seven fragments in
//     registers; the products over the 16 keys of a tile are v_mfma_f32_16x16x16_f16.
// Same arithmetic per element as k_attention_patch / k_attention_pcs2; the softmax runs over two partial key sets per head and is
// merged exactly (max / rescale), so results differ from the other kernels at fp32 round-off only.

#ifdef HMVIT_PROBE
#define PATCH16_TRACE(iter, slot)                                                                                      \
    do {                                                                                                               \
        if (p.trace && blockIdx.x == 0 && (wave == 0 || wave == 13) && lane == 0 && (iter) < 64)                       \
            p.trace[1024 + (wave ? 1024 : 0) + (iter) * 16 + (slot)] = __builtin_readcyclecounter();                   \
    } while (0)
#else
#define PATCH16_TRACE(iter, slot) do {} while (0)
#endif

// tables of one (item, source chunk, 16-key tile): built by k_patch16_tables once per forward (they depend on the pair transforms and the
// window only, not on the stage), read by the attention kernel through LDS
struct Patch16Tab {
    int list[40];                                    // token index of patch row R
    unsigned short taddr[16][4];                     // 128 R + 16 ((R >> 1) & 3) of tap k's row
    float tw[16][4];                                 // tap weights (0: out of range / masked key)
    unsigned meta[4];                                // rows, visible-key bits (16), bit 0 = identity chunk, -
};
static_assert(sizeof(Patch16Tab) == 560, "Patch16Tab layout");
struct Patch16Item {
    static constexpr int NCH = 4;
    static constexpr int BYTES = 9 * 1024;           // 16 tables (8960 bytes), padded to whole 1 KB requests
    Patch16Tab tab[NCH][4];
};

struct Patch16Shared {
    static constexpr int WAVES = 16;
    static constexpr int ROWS = 40;                  // patch capacity of a 16-key tile: 5 requests of 8 rows (33 at most for rigid pairs)
    static constexpr int SLOT = 5 * 1024;
    static constexpr int NCH = 4;
    static constexpr int VS = 40;                    // halves per row of the V' staging tile (16 keys x 32 channels)
    unsigned char slot[WAVES][SLOT];                 // wave-private patch (K', then V'), staging tile, merge exchange
    // per item parity: the item's tables as k_patch16_tables wrote them (Patch16Tab[chunk - 1][tile group = 2 half + kt]), one 9 KB block
    unsigned char tabraw[2][Patch16Item::BYTES];
    float bkv[HMVIT_NUM_TYPES * HMVIT_NUM_TYPES][2][256];
    float bq[HMVIT_NUM_TYPES][256];
    float biastab[8][15][16];                        // [head][q row - k row + 7][7 - (q col - k col)]
    half8 qlo[8][4][64];                             // low halves of the item's query operands [head][query tile][lane] (the high halves stay in
                                                     // registers: 16 of the 128 a wave has); written identically by both waves of a head
    int mode[kMaxSlots], cav[kMaxSlots], ego_e[kMaxSlots];
    int iconst[kMaxSlots][2];
};
static_assert(sizeof(Patch16Shared) <= 160 * 1024, "Patch16Shared exceeds the LDS of a CU");

struct Patch16Step {
    const float* kpl;
    int valid, ident, nk, rows, par, ci, grp, h, tsel;
    int wx, wy;
    unsigned vis;              // bit k: key k of the tile is visible
};

__device__ __forceinline__ PcItemC patch16_item_consts(const AttnParams& p, const Patch16Shared& sm, const PcItem& it) {
    const int2 v = *reinterpret_cast<const int2*>(sm.iconst[it.b * p.L + it.ego]);
    PcItemC r;
    r.tev = __builtin_amdgcn_readfirstlane(v.x);
    r.tsel = __builtin_amdgcn_readfirstlane(v.y);
    return r;
}

// Pre-pass: the tables of every (sample, ego, window) item -> out[item] (Patch16Item, stride Patch16Item::BYTES); one workgroup of 4
// waves per item, wave w = source chunk w + 1; lane = key n of the window (half = lane >> 5, tile = (lane >> 4) & 1), every group of 16
// lanes builds the bitmap / ranks of ITS tile.  Same sampling arithmetic as the gather kernels (make_taps_xy).
__global__ __launch_bounds__(256) void k_patch16_tables(AttnParams p, unsigned char* __restrict__ out) {
    __shared__ unsigned bm[4][4][16];
    __shared__ int bb[4][4][2];
    const int H = p.H, W = p.W, L = p.L, X = H / 8, Y = W / 8;
    const int pos = blockIdx.x;
    int r_ = pos;
    const int wy = r_ % Y; r_ /= Y;
    const int wx = r_ % X; r_ /= X;
    const int ego = r_ % p.n_ego, b = r_ / p.n_ego;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = wave + 1;
    if (c >= p.n_src) return;
    const int ci = wave, grp = lane >> 4, k = lane & 15;
    Patch16Tab& T = reinterpret_cast<Patch16Item*>(out + (size_t)pos * Patch16Item::BYTES)->tab[ci][grp];
    const int src = pc_src(c, ego);
    const float* a = p.ainv + ((size_t)(b * L + src) * L + ego) * 8;
    const bool cav = p.cav[b * L + src] != 0;
    if (a[6] != 0.f) {     // a source at the ego's own pose: its tiles are the window's own pixels (no tables)
        if (k == 0) *reinterpret_cast<uint4*>(T.meta) = make_uint4(16u, cav ? 0xffffu : 0u, 1u, 0u);
        return;
    }
    int row, col;
    token_pixel(HMVIT_PART_WINDOW, 8, X, Y, wx, wy, lane, row, col);
    const TapsXY t = make_taps_xy(a, col, row, H, W);
    const bool vis = cav && t.roi != 0.f;
    bool tv[4];
    int tx[4], ty[4];
    int xmin = 0x7fffffff, ymin = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        tv[q] = vis && t.w[q] != 0.f;
        tx[q] = t.x0 + (q & 1);
        ty[q] = t.y0 + (q >> 1);
        if (tv[q]) { xmin = min(xmin, tx[q]); ymin = min(ymin, ty[q]); }
    }
    bm[ci][grp][k] = 0u;
    if (k == 0) { bb[ci][grp][0] = 0x7fffffff; bb[ci][grp][1] = 0x7fffffff; }
    patch_wave_sync();
    if (xmin != 0x7fffffff) {
        atomicMin(&bb[ci][grp][0], xmin);
        atomicMin(&bb[ci][grp][1], ymin);
    }
    patch_wave_sync();
    const int bx = bb[ci][grp][0], by = bb[ci][grp][1];
    int lx[4], ly[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // inside [0, 15] for the transforms the kernel is launched for; clamped so that nothing else can leave the tables
        lx[q] = min(max(tx[q] - bx, 0), 15);
        ly[q] = min(max(ty[q] - by, 0), 15);
    }
    const unsigned b0 = (tv[0] ? 1u << lx[0] : 0u) | (tv[1] ? 1u << lx[1] : 0u);
    const unsigned b1 = (tv[2] ? 1u << lx[2] : 0u) | (tv[3] ? 1u << lx[3] : 0u);
    if (b0) atomicOr(&bm[ci][grp][ly[0]], b0);
    if (b1) atomicOr(&bm[ci][grp][ly[2]], b1);
    patch_wave_sync();
    int pre0 = 0, pre1 = 0, total = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = __builtin_popcount(bm[ci][grp][r]);
        total += n;
        pre0 += (r < ly[0]) ? n : 0;
        pre1 += (r < ly[2]) ? n : 0;
    }
    const unsigned m0 = bm[ci][grp][ly[0]], m1 = bm[ci][grp][ly[2]];
    int R[4];
    R[0] = pre0 + __builtin_popcount(m0 & ((1u << lx[0]) - 1u));
    R[1] = pre0 + __builtin_popcount(m0 & ((1u << lx[1]) - 1u));
    R[2] = pre1 + __builtin_popcount(m1 & ((1u << lx[2]) - 1u));
    R[3] = pre1 + __builtin_popcount(m1 & ((1u << lx[3]) - 1u));
    unsigned short ta[4];
    float w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = tv[q] ? min(R[q], Patch16Shared::ROWS - 1) : 0;
        ta[q] = (unsigned short)(r * 128 + ((r >> 1) & 3) * 16);
        w4[q] = tv[q] ? t.w[q] : 0.f;
        if (tv[q]) T.list[r] = ty[q] * W + tx[q];
    }
    *reinterpret_cast<uint2*>(T.taddr[k]) = make_uint2((unsigned)ta[0] | ((unsigned)ta[1] << 16), (unsigned)ta[2] | ((unsigned)ta[3] << 16));
    *reinterpret_cast<float4*>(T.tw[k]) = make_float4(w4[0], w4[1], w4[2], w4[3]);
    const unsigned vb = (unsigned)(__ballot(vis) >> (16 * grp)) & 0xffffu;
    if (k == 0) {
        // a tile without a visible key (walked when its half chunk has one elsewhere) still gets one finite row to blend with weight 0
        if (total == 0) T.list[0] = 0;
        *reinterpret_cast<uint4*>(T.meta) = make_uint4((unsigned)min(max(total, 1), Patch16Shared::ROWS), vb, 0u, 0u);
    }
}

// the tables of item `it` -> LDS block `dst` (9 requests of 1 KB: waves 0-8 send one each)
__device__ __forceinline__ void patch16_fetch_tables(const AttnParams& p, const PcItem& it, unsigned dst, int wave, int lane) {
    if (wave >= 9) return;
    const int X = p.H / 8, Y = p.W / 8;
    const size_t pos = (size_t)((it.b * p.n_ego + it.ego) * X + it.wx) * Y + it.wy;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.patch_tab) + pos * Patch16Item::BYTES + wave * 1024 + lane * 16;
    const unsigned d = __builtin_amdgcn_readfirstlane(dst + wave * 1024);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(d) : "memory");
}

__device__ __forceinline__ Patch16Step patch16_describe(const AttnParams& p, const Patch16Shared& sm, const PcItem& it, const PcItemC& ic,
                                                        int par, int slot, int hl, int kt) {
    const int L = p.L, C = p.C;
    const size_t PC = (size_t)p.H * p.W * C;
    Patch16Step S;
    const int c = slot >> 1, h = slot & 1;
    const int src = pc_src(c, it.ego), ev = (ic.tev >> 4) & 15;
    S.valid = 1;
    S.tsel = (ic.tsel >> (4 * c)) & 15;
    S.kpl = reinterpret_cast<const float*>(p.kv) + ((size_t)((it.b * L + src) * p.E + ev) * 2) * PC + hl * 32;
    S.par = par; S.ci = c - 1; S.h = h; S.grp = 2 * h + kt; S.wx = it.wx; S.wy = it.wy;
    if (c == 0) {
        S.ident = 1; S.rows = 16;
        S.vis = ((ic.tev >> 8) & 1) ? 0xffffu : 0u;
    } else {
        const uint4 m = *reinterpret_cast<const uint4*>(reinterpret_cast<const Patch16Item*>(sm.tabraw[par])->tab[c - 1][S.grp].meta);
        S.rows = __builtin_amdgcn_readfirstlane((int)m.x);
        S.vis = (unsigned)__builtin_amdgcn_readfirstlane((int)m.y);
        S.ident = __builtin_amdgcn_readfirstlane((int)m.z) & 1;
    }
    S.nk = S.ident ? 2 : (S.rows + 7) >> 3;
    return S;
}

// one plane (0: K', 1: V') of step S into this wave's patch: S.nk requests of 8 rows x 128 bytes (lane mapping of PatchShared)
__device__ __forceinline__ void patch16_request(const AttnParams& p, const Patch16Shared& sm, const Patch16Step& S, int plane, int kt, unsigned slot, int lane) {
#ifdef HMVIT_EXP_PATCH_NODMA
    return;
#endif
    const int W = p.W, C = p.C;
    const int r = lane >> 3, q = lane & 7;
    const int sx = (q & 3) ^ ((r >> 1) & 3);
    const int piece = 4 * (sx >> 1) + 2 * (q >> 2) + (sx & 1);
    const unsigned vplane = plane ? (unsigned)((size_t)p.H * W * C * 4) : 0u;
    unsigned off[5];
#pragma unroll
    for (int blk = 0; blk < 5; ++blk) {
        int tok = 0;
        if (blk < S.nk) {
            if (S.ident) tok = (S.wx * 8 + 4 * S.h + 2 * kt + blk) * W + S.wy * 8 + r;
            else tok = reinterpret_cast<const Patch16Item*>(sm.tabraw[S.par])->tab[S.ci][S.grp].list[min(blk * 8 + r, S.rows - 1)];
        }
        off[blk] = (unsigned)tok * (unsigned)(C * 4) + (unsigned)(piece * 16) + vplane;
    }
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) : : "memory");
#pragma unroll
    for (int blk = 0; blk < 5; ++blk) {
        if (blk < S.nk) {
            const unsigned d = __builtin_amdgcn_readfirstlane(slot + blk * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off[blk]), "s"(S.kpl), "s"(d) : "memory");
        }
    }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
}

// the 8 channels of octet g of key lq of the tile, bias added, as f32 (one tile of patch_blend)
__device__ __forceinline__ void patch16_blend(const Patch16Shared& sm, const Patch16Step& S, const unsigned char* slot, const float* bias, int lq, int lxor,
                                              float (&k8)[8]) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias), b1 = *reinterpret_cast<const float4*>(bias + 4);
    k8[0] = b0.x; k8[1] = b0.y; k8[2] = b0.z; k8[3] = b0.w; k8[4] = b1.x; k8[5] = b1.y; k8[6] = b1.z; k8[7] = b1.w;
#ifdef HMVIT_EXP_PATCH_NOBLEND
    if (k8[0] == 1.2345f) {
#else
    if (S.ident) {
#endif
        const unsigned a0 = (unsigned)(lq * 128 + ((lq >> 1) & 3) * 16) ^ (unsigned)lxor;
        const float4 v0 = *reinterpret_cast<const float4*>(slot + a0), v1 = *reinterpret_cast<const float4*>(slot + (a0 ^ 16u));
        k8[0] += v0.x; k8[1] += v0.y; k8[2] += v0.z; k8[3] += v0.w;
        k8[4] += v1.x; k8[5] += v1.y; k8[6] += v1.z; k8[7] += v1.w;
#ifdef HMVIT_EXP_PATCH_NOBLEND
    } else if (k8[1] == 1.2345f) {
#else
    } else {
#endif
        const Patch16Tab& T = reinterpret_cast<const Patch16Item*>(sm.tabraw[S.par])->tab[S.ci][S.grp];
        const uint2 tq = *reinterpret_cast<const uint2*>(T.taddr[lq]);
        const float4 wq = *reinterpret_cast<const float4*>(T.tw[lq]);
        const unsigned ta[4] = {tq.x & 0xffffu, tq.x >> 16, tq.y & 0xffffu, tq.y >> 16};
        const float ww[4] = {wq.x, wq.y, wq.z, wq.w};
        float2v a01 = {k8[0], k8[1]}, a23 = {k8[2], k8[3]}, a45 = {k8[4], k8[5]}, a67 = {k8[6], k8[7]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned a0 = ta[t] ^ (unsigned)lxor;
            const float4v v0 = *reinterpret_cast<const float4v*>(slot + a0), v1 = *reinterpret_cast<const float4v*>(slot + (a0 ^ 16u));
            const float2v wv = (float2v)(ww[t]);
            a01 = __builtin_elementwise_fma(wv, v0.xy, a01);
            a23 = __builtin_elementwise_fma(wv, v0.zw, a23);
            a45 = __builtin_elementwise_fma(wv, v1.xy, a45);
            a67 = __builtin_elementwise_fma(wv, v1.zw, a67);
        }
        k8[0] = a01.x; k8[1] = a01.y; k8[2] = a23.x; k8[3] = a23.y; k8[4] = a45.x; k8[5] = a45.y; k8[6] = a67.x; k8[7] = a67.y;
    }
}

__device__ __forceinline__ void patch16_loop(const AttnParams& p, Patch16Shared& sm, int wave, int lane) {
    using SM = Patch16Shared;
    constexpr int VS = SM::VS;
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;
    const float LOG2E = 1.4426950408889634f * kl;
    const int hl = wave & 7, kt = wave >> 3;               // head, key tile of every half chunk
    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / 8, Y = W / 8;
    const int lq = lane & 15, g = lane >> 4;
    const bool ego_fastest = (p.variant & 0x200) == 0;
    unsigned char* const sbase = sm.slot[wave];
    unsigned char* const pbase = sm.slot[wave ^ 8];        // the partner's buffer (same head, other key tile)
    const unsigned slds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sbase;
    const int lxor = 64 * (g & 1) + 32 * (g >> 1);
    half_t* const vth = reinterpret_cast<half_t*>(sbase);  // V' staging tile over the patch: hi, then lo (16 keys x 32 channels)
    half_t* const vtl = vth + 16 * VS;
    // this lane's part of the bias table index: row (q row - k row + 7) of 16 floats, column 7 - q col + k col (+ r)
    const float* btab = &sm.biastab[hl][0][0] + ((lq >> 3) - (g >> 1) - 2 * kt + 7) * 16 + 7 - (lq & 7) + 4 * (g & 1);

    PcCursor cur = pcs2_cursor();
    PcItem it, itn;
    if (!pc_fetch(p, X, Y, 1, ego_fastest, cur, it)) return;
    bool nvalid = pc_fetch(p, X, Y, 1, ego_fastest, cur, itn);
    int par = 0;
    const unsigned tlds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm.tabraw[0];
    patch16_fetch_tables(p, it, tlds, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pc_wg_barrier();
    PcItemC ic = patch16_item_consts(p, sm, it);
    unsigned rest = pcs2_bits(p, it, X, Y), restn = nvalid ? pcs2_bits(p, itn, X, Y) : 0u;
    // the item after the next one is fetched (scalar loads of the work list and the visibility words: ~1 us each under load) in the middle
    // of an item, not at its boundary, where every wave of the workgroup would wait for them
    PcItem itn2 = itn;
    bool nvalid2 = false, fetch_due = true;
    unsigned restn2 = 0;
    Patch16Step S = patch16_describe(p, sm, it, ic, par, __builtin_ctz(rest), hl, kt);
    rest &= rest - 1;
    patch16_request(p, sm, S, 0, kt, slds, lane);

    half8 qhh[4];
    float m_run[4], l_run[4];
    float4v o_acc[4][2];
    bool first = true;
    int tstep = 0;

    while (true) {
        PATCH16_TRACE(tstep, 0);
        if (first) {
            // ---- item prologue (all waves have passed the barrier that ended the item before: its table set may be overwritten) ----
            // the item's queries first (8 loads per lane in flight), the next item's tables (waves 0-3) while they travel, then the split
            const float* qpl = reinterpret_cast<const float*>(p.q) + (size_t)(it.b * L + it.ego) * P * C + hl * 32 + g * 8;
            float4 qa[4], qb[4];
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                int row, col;
                token_pixel(HMVIT_PART_WINDOW, 8, X, Y, it.wx, it.wy, qt * 16 + lq, row, col);
                const float* a = qpl + (size_t)(row * W + col) * C;
                qa[qt] = *reinterpret_cast<const float4*>(a);
                qb[qt] = *reinterpret_cast<const float4*>(a + 4);
            }
            PATCH16_TRACE(tstep, 12);
            PATCH16_TRACE(tstep, 13);
            const int te = sm.mode[it.b * L + it.ego];
            const float4 b0 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8]);
            const float4 b1 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8 + 4]);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const float q8[8] = {qa[qt].x + b0.x, qa[qt].y + b0.y, qa[qt].z + b0.z, qa[qt].w + b0.w,
                                     qb[qt].x + b1.x, qb[qt].y + b1.y, qb[qt].z + b1.z, qb[qt].w + b1.w};
                half8 ql;
                split_pk8(q8, qhh[qt], ql);
                sm.qlo[hl][qt][lane] = ql;
                m_run[qt] = -INFINITY;
                l_run[qt] = 0.f;
                o_acc[qt][0] = (float4v)(0.f);
                o_acc[qt][1] = (float4v)(0.f);
            }
            // the next item's tables into the other block - behind the queries: at the item boundary sixteen waves fill the vector-memory
            // queue with their query loads, and a request issued among them waited 4-8 k cycles just to be accepted (patch16_trace.py)
            if (nvalid) patch16_fetch_tables(p, itn, tlds + (par ^ 1) * Patch16Item::BYTES, wave, lane);
            first = false;
            PATCH16_TRACE(tstep, 11);
        }
        // ---- the step after S ----
        const bool last = rest == 0;
        Patch16Step N;
        N.valid = 0; N.nk = 0;
        if (!last) {
            N = patch16_describe(p, sm, it, ic, par, __builtin_ctz(rest), hl, kt);
            rest &= rest - 1;
        }

        PATCH16_TRACE(tstep, 1);
        // ---- K' of step S ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PATCH16_TRACE(tstep, 2);
        half8 khh, khl;
        {
            float kf[8];
            patch16_blend(sm, S, sbase, &sm.bkv[S.tsel][0][hl * 32 + g * 8], lq, lxor, kf);
            split_pk8(kf, khh, khl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the K' rows are in registers: the V' rows may replace them
        PATCH16_TRACE(tstep, 3);
        patch16_request(p, sm, S, 1, kt, slds, lane);
        PATCH16_TRACE(tstep, 4);

        // ---- two query tiles at a time: S^T = K' Q^T + bias (+ mask), running maximum, exponentials as operand halves; the V' rows land
        // during the first pair's logits, the next step's K' rows are requested right behind the V' blend ----
        auto logits = [&](auto qt_c, half4& ph, half4& pl) {
            constexpr int qt = decltype(qt_c)::value;
            const float* bt = btab + (2 * qt - 4 * S.h) * 16;
            float4v acc = {bt[0], bt[1], bt[2], bt[3]};
            if (S.vis != 0xffffu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = ((S.vis >> (4 * g + r)) & 1u) ? acc[r] : -INFINITY;
            }
            const half8 qhl = sm.qlo[hl][qt][lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khl, qhh[qt], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh, qhl, acc, 0, 0, 0);
            const float4v s = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh, qhh[qt], acc, 0, 0, 0);
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = max_over_lane_groups(mx);
            const float m_new = max_raw(m_run[qt], mx * LOG2E);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_safe);
            const float e0 = __builtin_amdgcn_exp2f(fmaf(s[0], LOG2E, -m_safe)), e1 = __builtin_amdgcn_exp2f(fmaf(s[1], LOG2E, -m_safe));
            const float e2 = __builtin_amdgcn_exp2f(fmaf(s[2], LOG2E, -m_safe)), e3 = __builtin_amdgcn_exp2f(fmaf(s[3], LOG2E, -m_safe));
            split_pk4(e0, e1, e2, e3, ph, pl);
            m_run[qt] = m_new;
            l_run[qt] = fmaf(l_run[qt], alpha, (e0 + e1) + (e2 + e3));
            o_acc[qt][0] *= alpha;
            o_acc[qt][1] *= alpha;
        };
        half4 vhh[2], vhl[2];
        auto products = [&](auto qt_c, const half4& ph, const half4& pl) {
            constexpr int qt = decltype(qt_c)::value;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vhl[dt], ph, o_acc[qt][dt], 0, 0, 0);
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vhh[dt], pl, o_acc[qt][dt], 0, 0, 0);
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(vhh[dt], ph, o_acc[qt][dt], 0, 0, 0);
            }
        };
        half4 ph0, pl0, ph1, pl1;
#ifndef HMVIT_EXP_PATCH_NOMATH
        // (scheduling fences between the tiles: left alone hipcc overlaps them and the wave does not fit its 128 registers)
        logits(std::integral_constant<int, 0>{}, ph0, pl0);
        __builtin_amdgcn_sched_barrier(0);
        logits(std::integral_constant<int, 1>{}, ph1, pl1);
        __builtin_amdgcn_sched_barrier(0);
#else
        ph0 = ph1 = half4{khh[0], khh[1], khh[2], khh[3]}; pl0 = pl1 = half4{khl[0], khl[1], khl[2], khl[3]};
#endif

        PATCH16_TRACE(tstep, 5);
        // ---- V' of step S ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PATCH16_TRACE(tstep, 6);
        {
            float vf[8];
            patch16_blend(sm, S, sbase, &sm.bkv[S.tsel][1][hl * 32 + g * 8], lq, lxor, vf);
            half8 th, tl;
            split_pk8(vf, th, tl);
            patch_wave_sync();                                   // every tap is in registers: the bytes become the staging tile
            *reinterpret_cast<half8*>(vth + lq * VS + g * 8) = th;
            *reinterpret_cast<half8*>(vtl + lq * VS + g * 8) = tl;
            patch_wave_sync();
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // lane (lq, g) receives keys 4 g .. 4 g + 3 of channel 8 (lq >> 2) + 4 dt + (lq & 3) (the channel order of k_attention_pcs2's V^T tiles)
                const int off = (4 * g + (lq >> 2)) * VS + (lq & 3) * 8 + dt * 4;
                const fp16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vth + off));
                const fp16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(vtl + off));
                vhh[dt] = half4{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                vhl[dt] = half4{(half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PATCH16_TRACE(tstep, 7);
        if (N.valid) patch16_request(p, sm, N, 0, kt, slds, lane);
        PATCH16_TRACE(tstep, 8);

#ifndef HMVIT_EXP_PATCH_NOMATH
        products(std::integral_constant<int, 0>{}, ph0, pl0);
        products(std::integral_constant<int, 1>{}, ph1, pl1);
        __builtin_amdgcn_sched_barrier(0);
        logits(std::integral_constant<int, 2>{}, ph0, pl0);
        __builtin_amdgcn_sched_barrier(0);
        logits(std::integral_constant<int, 3>{}, ph1, pl1);
        __builtin_amdgcn_sched_barrier(0);
        products(std::integral_constant<int, 2>{}, ph0, pl0);
        products(std::integral_constant<int, 3>{}, ph1, pl1);
        __builtin_amdgcn_sched_barrier(0);
#else
        for (int qt = 0; qt < 4; ++qt) { o_acc[qt][0][0] += (float)vhh[0][0] * (float)ph0[0]; o_acc[qt][1][0] += (float)vhl[1][0] + (float)pl1[0]; l_run[qt] += 1.f; }
#endif

        if (fetch_due) {
            fetch_due = false;
            nvalid2 = nvalid && pc_fetch(p, X, Y, 1, ego_fastest, cur, itn2);
            restn2 = nvalid2 ? pcs2_bits(p, itn2, X, Y) : 0u;
        }
        PATCH16_TRACE(tstep, 9);
        if (last) {
            // ---- merge with the partner (same head, other key tile): this wave finishes query tiles 2 kt, 2 kt + 1 and hands the other
            // two over through its own buffer ([j][lane][4 floats]: O tiles of the two handed-over query tiles, then maxima and sums) ----
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the key tile is a compile-time constant inside: a run-time index into the state arrays would put them into scratch memory)
            auto hand_over = [&](auto kt_c) {
                constexpr int q0 = 2 * (decltype(kt_c)::value ^ 1);
                float4* ex = reinterpret_cast<float4*>(sbase);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        ex[(2 * j + dt) * 64 + lane] = make_float4(o_acc[q0 + j][dt][0], o_acc[q0 + j][dt][1], o_acc[q0 + j][dt][2], o_acc[q0 + j][dt][3]);
                ex[4 * 64 + lane] = make_float4(m_run[q0], m_run[q0 + 1], l_run[q0], l_run[q0 + 1]);
            };
            if (kt) hand_over(std::integral_constant<int, 1>{}); else hand_over(std::integral_constant<int, 0>{});
            pc_wg_barrier();
            float* outp = reinterpret_cast<float*>(p.out) + (size_t)(it.b * L + it.ego) * P * C;
            auto finish = [&](auto kt_c) {
                constexpr int KT = decltype(kt_c)::value;
                const float4* ex = reinterpret_cast<const float4*>(pbase);
                const float4 ml = ex[4 * 64 + lane];
                const float mo[2] = {ml.x, ml.y}, lo[2] = {ml.z, ml.w};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    constexpr int qb = 2 * KT;
                    const int qt = qb + j;
                    const float m = max_raw(m_run[qb + j], mo[j]);
                    const float ms = (m == -INFINITY) ? 0.f : m;
                    const float a1 = __builtin_amdgcn_exp2f(m_run[qb + j] - ms), a2 = __builtin_amdgcn_exp2f(mo[j] - ms);
                    const float lsum = xor32_sum(xor16_sum(fmaf(l_run[qb + j], a1, lo[j] * a2)));
                    const float inv = 1.f / lsum;
                    const float4 oa = ex[(2 * j + 0) * 64 + lane], ob = ex[(2 * j + 1) * 64 + lane];
                    float a[4] = {fmaf(o_acc[qb + j][0][0], a1, oa.x * a2) * inv, fmaf(o_acc[qb + j][0][1], a1, oa.y * a2) * inv,
                                  fmaf(o_acc[qb + j][0][2], a1, oa.z * a2) * inv, fmaf(o_acc[qb + j][0][3], a1, oa.w * a2) * inv};
                    float b[4] = {fmaf(o_acc[qb + j][1][0], a1, ob.x * a2) * inv, fmaf(o_acc[qb + j][1][1], a1, ob.y * a2) * inv,
                                  fmaf(o_acc[qb + j][1][2], a1, ob.z * a2) * inv, fmaf(o_acc[qb + j][1][3], a1, ob.w * a2) * inv};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        swap16_rows(a[e], b[e]);
                        swap32_rows(a[e], b[e]);
                    }
                    int row, col;
                    token_pixel(HMVIT_PART_WINDOW, 8, X, Y, it.wx, it.wy, qt * 16 + lq, row, col);
                    float* o = outp + (size_t)(row * W + col) * C + hl * 32 + 4 * g;
                    *reinterpret_cast<float4*>(o) = make_float4(a[0], a[1], a[2], a[3]);
                    *reinterpret_cast<float4*>(o + 16) = make_float4(b[0], b[1], b[2], b[3]);
                    if (p.lse && g == 0)
                        p.lse[((size_t)(it.b * L + it.ego) * P + row * W + col) * (C / 32) + hl] = ms * 0.6931471805599453f + logf(lsum);
                }
            };
            if (kt) finish(std::integral_constant<int, 1>{}); else finish(std::integral_constant<int, 0>{});
            PATCH16_TRACE(tstep, 10);
            if (!nvalid) break;
            // everyone has read its partner's buffer and (waves 0-3, in this item's prologue) finished the next item's tables
            pc_wg_barrier();
            it = itn; ic = patch16_item_consts(p, sm, it); par ^= 1;
            rest = restn;
            N = patch16_describe(p, sm, it, ic, par, __builtin_ctz(rest), hl, kt);
            rest &= rest - 1;
            patch16_request(p, sm, N, 0, kt, slds, lane);
            itn = itn2; nvalid = nvalid2; restn = restn2; fetch_due = true;
            first = true;
        }
        S = N;
        ++tstep;
    }
}

__global__ __launch_bounds__(1024) void k_attention_patch16(AttnParams p) {
    using SM = Patch16Shared;
    __shared__ __attribute__((aligned(16))) SM sm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const int C = p.C;       // 256
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * HMVIT_NUM_TYPES * 2 * 256; i += blockDim.x) {
            const int e = i / 512, pl = (i / 256) & 1, c = i % 256;
            sm.bkv[e][pl][c] = p.b_kv[(size_t)e * 2 * C + pl * C + c];
        }
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * 256; i += blockDim.x) sm.bq[i / 256][i % 256] = p.b_q[(i / 256) * C + (i % 256)];
        // the relative-position bias as a table: fragment v = (query tile - key tile) + 3, lane (ql, 4 g' + r) of bias_frag holds the entry of
        // row offset 2 (v - 3) + (ql >> 3) - (kl >> 3), column offset (ql & 7) - (kl & 7) (weights.bias_fragments); equal offsets hold equal values
        for (int i = threadIdx.x; i < 8 * 7 * 64 * 4; i += blockDim.x) {
            const int r = i & 3, ln = (i >> 2) & 63, v = (i >> 8) % 7, head = i / (7 * 256);
            const int ql = ln & 15, kk = 4 * (ln >> 4) + r;
            const int drow = 2 * (v - 3) + (ql >> 3) - (kk >> 3), dcol = (ql & 7) - (kk & 7);
            sm.biastab[head][drow + 7][7 - dcol] = p.bias_frag[i];
        }
        if (threadIdx.x < kMaxSlots) {
            sm.mode[threadIdx.x] = p.mode[threadIdx.x];
            sm.cav[threadIdx.x] = p.cav[threadIdx.x];
            sm.ego_e[threadIdx.x] = p.ego_e[threadIdx.x];
        }
    }
    __syncthreads();
    {
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
    __syncthreads();
    patch16_loop(p, sm, wave, threadIdx.x & 63);
}

size_t patch16_tables_bytes(int B, int n_ego, int H, int W) { return (size_t)B * n_ego * (H / 8) * (W / 8) * Patch16Item::BYTES; }
int launch_patch16_tables(const AttnParams& p, void* ws, hipStream_t st) {
    const int n = p.B * p.n_ego * (p.H / 8) * (p.W / 8);
    if (n <= 0) return HMVIT_OK;
    HMVIT_CHECK_ARG(p.n_src <= Patch16Item::NCH + 1 && p.H % 8 == 0 && p.W % 8 == 0, "patch16 tables: n_src=%d (<= 5), %dx%d", p.n_src, p.H, p.W);
    hipLaunchKernelGGL(k_patch16_tables, dim3(n), dim3(256), 0, st, p, reinterpret_cast<unsigned char*>(ws));
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}

static int launch_attn_patch16(const AttnParams& p_in, hipStream_t st) {
    AttnParams p = p_in;
    if (const char* e = HMVIT_ENV("HMVIT_ATTN_TRACE")) p.trace = (unsigned long long*)strtoull(e, nullptr, 0);
    hipLaunchKernelGGL(k_attention_patch16, dim3(kPcs2Grid), dim3(1024), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
