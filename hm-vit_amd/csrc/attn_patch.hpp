// k_attention_patch (round 6): the split-precision attention of the LOCAL (window) stages with a de-duplicated source patch.
// Included by attn.hip inside namespace hmvit, behind k_attention_pcs2 (whose item walk, visibility words, query / bias / output
// conventions and matrix-instruction operand layouts it shares).
//
// What it replaces and why: k_attention_pcs2 gathers every key's four bilinear taps - four 1 KB rows per key and
// plane - into 128 registers of a loader wave, one wave per role and SIMD; per launch 20.8 GB of tap rows pass the vector-memory
// return path (64 B per clock and CU, TD busy 72 % of the launch) and neither role can hide its own latencies.  In a local window
// neighbouring keys share most of their taps: the 32 keys of a half chunk (4 x 8 pixels of the ego's window) touch 47 distinct
// source pixels on average (54 at most for a rigid transform; 128 tap loads before).  Here
//   * the EXACT touched set of a (window, source, half chunk) is found once per item: a 16 x 16 bitmap over the taps' bounding box
//     (LDS atomics), prefix popcounts = compact row numbers, one table entry per key and tap (LDS address of the row + weight);
//   * every wave is one head: it fetches ITS 128-byte slice of each touched row once by LDS-DMA (global_load_lds_dwordx4: 8 rows
//     per instruction, no registers while in flight) into a wave-private patch, and blends its own operand fragments from there:
//     K' straight into the A-operand registers of S^T = K' Q^T, V' through a wave-private staging tile (the transposing read).
//     Everything a wave touches between two items is its own: no roles, no per-step barriers (two workgroup barriers per ITEM, for
//     the shared tap tables), eight waves per CU that all do matrix + softmax work and drift apart freely;
//   * the patches are single-buffered: both are blended into operand registers at the top of a step, the rows of step s + 1 are
//     requested right behind that and have the step's matrix + softmax work to land (one vmcnt(0) per step).
// Arithmetic is the pcs2 kernel's (split operands, three products, f32 softmax); the softmax denominator is summed on the VALU.
// Preconditions (launch_attention): C = 256, window 8, local partition, identity self transforms, n_src <= 5, and transforms whose
// linear part is orthonormal to 2 % (HmvitFusionDesc::rigid_patch: then a half chunk never touches more than 64 source pixels).

// probe builds: cycle stamps of waves 0 and 5 of workgroup 0, 16 slots per step (tests/tools/patch_trace.py)
#ifdef HMVIT_PROBE
#define PATCH_TRACE(iter, slot)                                                                                        \
    do {                                                                                                               \
        if (p.trace && blockIdx.x == 0 && (hl == 0 || hl == 5) && lane == 0 && (iter) < 64)                            \
            p.trace[1024 + (hl ? 1024 : 0) + (iter) * 16 + (slot)] = __builtin_readcyclecounter();                     \
    } while (0)
#else
#define PATCH_TRACE(iter, slot) do {} while (0)
#endif

struct PatchShared {
    static constexpr int WAVES = 8;                  // = heads (C = 256, dim_head 32)
    static constexpr int ROWS = 64;                  // patch capacity per half chunk: 8 requests of 8 rows
    static constexpr int BLK = 8 * 128;              // bytes per request
    static constexpr int SLOT = 8 * BLK;
    static constexpr int NCH = 4;                    // source chunks with tables (n_src - 1)
    static constexpr int VS = 40;                    // halves per row of the V' staging tile (32 keys x 32 channels)
    // Wave-private patches.  Row R (this head's 32 channels = 8 pieces of 16 bytes) occupies bytes [128 R, 128 R + 128); piece
    // p = 4 g1 + 2 g0 + e (g = 2 g1 + g0: the channel octet of blend lane (key, g), e: its first / second b128 read) sits at position
    // q = 4 g0 + ((2 g1 + e) ^ s), s = (R >> 1) & 3, inside the row: a request fetches a row with 8 consecutive lanes (one 128-byte line
    // per 8 lanes: 20 cycles of the vector-memory path per request against 66 with the pieces of a row spread over the wave,
    // tools/probe/dma_map_probe.hip), and the 16 lanes a ds_read_b128 services together - 8 keys on octet g0 = 0, 8 on g0 = 1 - find 16
    // different banks when their rows differ in R & 7.
    unsigned char kslot[WAVES][SLOT];
    unsigned char vslot[WAVES][SLOT];                // V' patch; after its blend the same bytes hold the (hi | lo) staging tiles
    // per item parity and (chunk - 1, half): the touched rows and every key's taps
    int list[2][NCH][2][ROWS];                       // token index of patch row R
    unsigned short taddr[2][NCH][2][32][4];          // 128 R + 16 s of tap k's row
    float tw[2][NCH][2][32][4];                      // tap weights (0: out of range / masked key)
    unsigned meta[2][NCH][2][4];                     // rows, visible-key bits, bit 0 = identity chunk, -
    unsigned bm[NCH][2][16];                         // scratch of the table build: bitmap rows, bounding-box corner
    int bb[NCH][2][2];
    float bkv[HMVIT_NUM_TYPES * HMVIT_NUM_TYPES][2][256];
    float bq[HMVIT_NUM_TYPES][256];
    int mode[kMaxSlots], cav[kMaxSlots], ego_e[kMaxSlots];
    int iconst[kMaxSlots][2];
};
static_assert(sizeof(PatchShared) <= 160 * 1024, "PatchShared exceeds the LDS of a CU");

// one half chunk of an item as this wave sees it (all fields wave-uniform)
struct PatchStep {
    const float* kpl;          // K' plane of the source at this wave's head (V' = + P C floats)
    int valid, ident, nk, rows, par, ci, h, tsel;
    int wx, wy;
    unsigned vis;              // bit k: key k of the half chunk is visible
};

__device__ __forceinline__ void patch_wait_vm(int n) {     // wave-uniform n: at most n vector-memory operations still in flight
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void patch_wave_sync() {        // LDS operations of one wave execute in order; this keeps hipcc from reordering them
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ PcItemC patch_item_consts(const AttnParams& p, const PatchShared& sm, const PcItem& it) {
    const int2 v = *reinterpret_cast<const int2*>(sm.iconst[it.b * p.L + it.ego]);
    PcItemC r;
    r.tev = __builtin_amdgcn_readfirstlane(v.x);
    r.tsel = __builtin_amdgcn_readfirstlane(v.y);
    return r;
}

// Tables of one item: wave w < n_src - 1 takes source chunk w + 1, lanes 0-31 its first half (keys 0-31 of the window), lanes 32-63 the second.
__device__ __forceinline__ void patch_tables(const AttnParams& p, PatchShared& sm, const PcItem& it, int par, int wave, int lane) {
    using SM = PatchShared;
    // (measured and dropped: the build on waves 4-7 instead, so that the two waves of a SIMD run the item half a step out of phase -
    // four launches 6.27 against 6.2 ms: the waves that do not build wait for the tables at the item's second barrier either way)
    const int c = wave + 1;
    if (c >= p.n_src) return;
    const int H = p.H, W = p.W, L = p.L, X = H / 8, Y = W / 8;
    const int ci = wave, h = lane >> 5, k = lane & 31;
#ifdef HMVIT_EXP_PATCH_NOTABLES
    if (k == 0) *reinterpret_cast<uint4*>(sm.meta[par][ci][h]) = make_uint4(48u, 0xffffffffu, 0u, 0u);
    return;
#endif
    const int src = pc_src(c, it.ego);
    // the pair's sampling map through the scalar cache: a vector load would queue behind this wave's patch requests in flight
    const float* ag = p.ainv + __builtin_amdgcn_readfirstlane(((it.b * L + src) * L + it.ego) * 8);
    float a[8];
    {
        typedef float float8s __attribute__((ext_vector_type(8)));
        float8s av;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(av) : "s"(ag) : "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = av[i];
    }
    const bool cav = sm.cav[it.b * L + src] != 0;
    if (a[6] != 0.f) {     // a source at the ego's own pose: its half chunks are the window's own pixels (no tables)
        if (k == 0) *reinterpret_cast<uint4*>(sm.meta[par][ci][h]) = make_uint4(32u, cav ? 0xffffffffu : 0u, 1u, 0u);
        return;
    }
    int row, col;
    token_pixel(HMVIT_PART_WINDOW, 8, X, Y, it.wx, it.wy, h * 32 + k, row, col);
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
    if (k < 16) sm.bm[ci][h][k] = 0u;
    if (k == 0) { sm.bb[ci][h][0] = 0x7fffffff; sm.bb[ci][h][1] = 0x7fffffff; }
    patch_wave_sync();
    if (xmin != 0x7fffffff) {
        atomicMin(&sm.bb[ci][h][0], xmin);
        atomicMin(&sm.bb[ci][h][1], ymin);
    }
    patch_wave_sync();
    const int bx = sm.bb[ci][h][0], by = sm.bb[ci][h][1];
    int lx[4], ly[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // inside [0, 15] for the transforms this kernel is launched for; clamped so that nothing else can leave the tables
        lx[q] = min(max(tx[q] - bx, 0), 15);
        ly[q] = min(max(ty[q] - by, 0), 15);
    }
    const unsigned b0 = (tv[0] ? 1u << lx[0] : 0u) | (tv[1] ? 1u << lx[1] : 0u);
    const unsigned b1 = (tv[2] ? 1u << lx[2] : 0u) | (tv[3] ? 1u << lx[3] : 0u);
    if (b0) atomicOr(&sm.bm[ci][h][ly[0]], b0);
    if (b1) atomicOr(&sm.bm[ci][h][ly[2]], b1);
    patch_wave_sync();
    // rows above the two bitmap rows of this key, and all rows
    int pre0 = 0, pre1 = 0, total = 0;
    {
        unsigned bmv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(&sm.bm[ci][h][4 * i]);
            bmv[4 * i] = v.x; bmv[4 * i + 1] = v.y; bmv[4 * i + 2] = v.z; bmv[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = __builtin_popcount(bmv[r]);
            total += n;
            pre0 += (r < ly[0]) ? n : 0;
            pre1 += (r < ly[2]) ? n : 0;
        }
    }
    const unsigned m0 = sm.bm[ci][h][ly[0]], m1 = sm.bm[ci][h][ly[2]];
    int R[4];
    R[0] = pre0 + __builtin_popcount(m0 & ((1u << lx[0]) - 1u));
    R[1] = pre0 + __builtin_popcount(m0 & ((1u << lx[1]) - 1u));
    R[2] = pre1 + __builtin_popcount(m1 & ((1u << lx[2]) - 1u));
    R[3] = pre1 + __builtin_popcount(m1 & ((1u << lx[3]) - 1u));
    unsigned short ta[4];
    float w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = tv[q] ? min(R[q], SM::ROWS - 1) : 0;
        ta[q] = (unsigned short)(r * 128 + ((r >> 1) & 3) * 16);
        w4[q] = tv[q] ? t.w[q] : 0.f;
        if (tv[q]) sm.list[par][ci][h][r] = ty[q] * W + tx[q];
    }
    *reinterpret_cast<uint2*>(sm.taddr[par][ci][h][k]) = make_uint2((unsigned)ta[0] | ((unsigned)ta[1] << 16), (unsigned)ta[2] | ((unsigned)ta[3] << 16));
    *reinterpret_cast<float4*>(sm.tw[par][ci][h][k]) = make_float4(w4[0], w4[1], w4[2], w4[3]);
    const unsigned vb = (unsigned)(__ballot(vis) >> (32 * h));
    if (k == 0) {
        // a half chunk without a visible key (walked only when masked tiles are not skipped) still gets one finite row to blend with weight 0
        if (total == 0) sm.list[par][ci][h][0] = 0;
        *reinterpret_cast<uint4*>(sm.meta[par][ci][h]) = make_uint4((unsigned)min(max(total, 1), SM::ROWS), vb, 0u, 0u);
    }
}

__device__ __forceinline__ PatchStep patch_describe(const AttnParams& p, const PatchShared& sm, const PcItem& it, const PcItemC& ic,
                                                    int par, int slot, int hl) {
    const int L = p.L, C = p.C;
    const size_t PC = (size_t)p.H * p.W * C;
    PatchStep S;
    const int c = slot >> 1, h = slot & 1;
    const int src = pc_src(c, it.ego), ev = (ic.tev >> 4) & 15;
    S.valid = 1;
    S.tsel = (ic.tsel >> (4 * c)) & 15;
    S.kpl = reinterpret_cast<const float*>(p.kv) + ((size_t)((it.b * L + src) * p.E + ev) * 2) * PC + hl * 32;
    S.par = par; S.ci = c - 1; S.h = h; S.wx = it.wx; S.wy = it.wy;
    if (c == 0) {
        S.ident = 1; S.rows = 32;
        S.vis = ((ic.tev >> 8) & 1) ? 0xffffffffu : 0u;
    } else {
        const uint4 m = *reinterpret_cast<const uint4*>(sm.meta[par][c - 1][h]);
        S.rows = __builtin_amdgcn_readfirstlane((int)m.x);
        S.vis = (unsigned)__builtin_amdgcn_readfirstlane((int)m.y);
        S.ident = __builtin_amdgcn_readfirstlane((int)m.z) & 1;
    }
    S.nk = S.ident ? 4 : (S.rows + 7) >> 3;
    return S;
}

// The rows of step S, both planes, into this wave's patches: S.nk requests of 8 rows x 128 bytes per plane.  Lane l fetches, for row
// 8 blk + (l >> 3), the piece that belongs at position l & 7 of the row (PatchShared): 8 consecutive lanes read one 128-byte line.
// Two stages: patch_offsets reads the row list (one batch of LDS reads) into this lane's byte offsets inside the source's K' plane;
// patch_issue sends the requests of blocks [b0, b1) - scalar plane base + 32-bit lane offset, V' = the same + one plane - so that the
// step can spread them between its query tiles: eight waves sending their 12-14 requests in one burst each waited ~140 cycles per
// request for the vector-memory path (20 cycles per request and CU: tools/probe/dma_map_probe.hip, tests/tools/patch_trace.py).
struct PatchReq {
    unsigned off[8];
};
__device__ __forceinline__ PatchReq patch_offsets(const AttnParams& p, const PatchShared& sm, const PatchStep& S, int lane) {
    const int W = p.W, C = p.C;
    const int r = lane >> 3, q = lane & 7;
    const int sx = (q & 3) ^ ((r >> 1) & 3);                     // 2 g1 + e
    const int piece = 4 * (sx >> 1) + 2 * (q >> 2) + (sx & 1);
    PatchReq R;
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        int tok = 0;
        if (blk < S.nk) {
            if (S.ident) tok = (S.wx * 8 + 4 * S.h + blk) * W + S.wy * 8 + r;
            else tok = sm.list[S.par][S.ci][S.h][min(blk * 8 + r, S.rows - 1)];
        }
        R.off[blk] = (unsigned)tok * (unsigned)(C * 4) + (unsigned)(piece * 16);
    }
    return R;
}
__device__ __forceinline__ void patch_issue(const AttnParams& p, const PatchStep& S, const PatchReq& R, int b0, int b1, unsigned kslot, unsigned vslot) {
    using SM = PatchShared;
#ifdef HMVIT_EXP_PATCH_NODMA
    return;
#endif
    const unsigned vplane = (unsigned)((size_t)p.H * p.W * p.C * 4);
    // m0 carries the LDS address of a request; saved once around the batch (between the requests only the next address is moved in)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) : : "memory");
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        if (blk >= b0 && blk < b1 && blk < S.nk) {
            const unsigned dk = __builtin_amdgcn_readfirstlane(kslot + blk * SM::BLK), dv = __builtin_amdgcn_readfirstlane(vslot + blk * SM::BLK);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(R.off[blk]), "s"(S.kpl), "s"(dk) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(R.off[blk] + vplane), "s"(S.kpl), "s"(dv) : "memory");
        }
    }
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
}

// One plane of step S from this wave's patch: the 8 channels of octet g of keys kt 16 + lq (kt = 0, 1), bias added, as f32.
__device__ __forceinline__ void patch_blend(const PatchShared& sm, const PatchStep& S, const unsigned char* slot, const float* bias, int lq, int lxor,
                                            float (&o)[2][8]) {
    const float4 b0 = *reinterpret_cast<const float4*>(bias), b1 = *reinterpret_cast<const float4*>(bias + 4);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kt * 16 + lq;
        float* k8 = o[kt];
        k8[0] = b0.x; k8[1] = b0.y; k8[2] = b0.z; k8[3] = b0.w; k8[4] = b1.x; k8[5] = b1.y; k8[6] = b1.z; k8[7] = b1.w;
#ifdef HMVIT_EXP_PATCH_NOBLEND
        if (k8[0] == 1.2345f) {
#else
        if (S.ident) {
#endif
            const unsigned a0 = (unsigned)(key * 128 + ((key >> 1) & 3) * 16) ^ (unsigned)lxor;
            const float4 v0 = *reinterpret_cast<const float4*>(slot + a0), v1 = *reinterpret_cast<const float4*>(slot + (a0 ^ 16u));
            k8[0] += v0.x; k8[1] += v0.y; k8[2] += v0.z; k8[3] += v0.w;
            k8[4] += v1.x; k8[5] += v1.y; k8[6] += v1.z; k8[7] += v1.w;
#ifdef HMVIT_EXP_PATCH_NOBLEND
        } else if (k8[1] == 1.2345f) {
#else
        } else {
#endif
            const uint2 tq = *reinterpret_cast<const uint2*>(sm.taddr[S.par][S.ci][S.h][key]);
            const float4 wq = *reinterpret_cast<const float4*>(sm.tw[S.par][S.ci][S.h][key]);
            const unsigned ta[4] = {tq.x & 0xffffu, tq.x >> 16, tq.y & 0xffffu, tq.y >> 16};
            const float ww[4] = {wq.x, wq.y, wq.z, wq.w};
            // packed f32 multiply-adds (two channels per instruction, the tap weight broadcast): half the issue slots of the blend
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
}

__device__ __forceinline__ void patch_loop(const AttnParams& p, PatchShared& sm, int wave, int lane) {
    using SM = PatchShared;
    constexpr int VS = SM::VS;
    const float kl = p.k_logit != 0.f ? p.k_logit : 1.f;
    const float LOG2E = 1.4426950408889634f * kl;
    const int hl = wave;                                   // this wave's head
    const int C = p.C, H = p.H, W = p.W, L = p.L, P = H * W;
    const int X = H / 8, Y = W / 8;
    const int lq = lane & 15, g = lane >> 4;
    const bool ego_fastest = (p.variant & 0x200) == 0;
    unsigned char* const kbase = sm.kslot[hl];
    unsigned char* const vbase = sm.vslot[hl];
    const unsigned klds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)kbase;
    const unsigned vlds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)vbase;
    const int lxor = 64 * (g & 1) + 32 * (g >> 1);         // this lane's channel octet inside a patch row (PatchShared)
    half_t* const vth = reinterpret_cast<half_t*>(vbase);  // staging tiles over the V' patch: hi, then lo
    half_t* const vtl = vth + 32 * VS;

    PcCursor cur = pcs2_cursor();
    PcItem it, itn;
    if (!pc_fetch(p, X, Y, 1, ego_fastest, cur, it)) return;
    bool nvalid = pc_fetch(p, X, Y, 1, ego_fastest, cur, itn);
    int par = 0;
    patch_tables(p, sm, it, par, wave, lane);
    pc_wg_barrier();
    PcItemC ic = patch_item_consts(p, sm, it), icn = ic;
    unsigned rest = pcs2_bits(p, it, X, Y), restn = 0;
    PatchStep S = patch_describe(p, sm, it, ic, par, __builtin_ctz(rest), hl);
    rest &= rest - 1;
    {
        const PatchReq R0 = patch_offsets(p, sm, S, lane);
        patch_issue(p, S, R0, 0, 8, klds, vlds);
    }

    float4v biasf[7];
#pragma unroll
    for (int v = 0; v < 7; ++v) biasf[v] = *reinterpret_cast<const float4v*>(p.bias_frag + ((size_t)(hl * 7 + v) * 64 + lane) * 4);
    half8 qhh[4], qhl[4];
    float m_run[4], l_run[4];    // running maximum (exponent units) and this lane's share of the row sum, per query column
    float4v o_acc[4][2];
    bool first = true, first_item = true;
    int tstep = 0;

    while (true) {
        PATCH_TRACE(tstep, 0);
        if (first) {
            // ---- item prologue: tables of the next item, this item's queries ----
            if (!first_item) pc_wg_barrier();     // every wave is done with the item before: its table set may be overwritten
            first_item = false;
            PATCH_TRACE(tstep, 12);
            if (nvalid) patch_tables(p, sm, itn, par ^ 1, wave, lane);
            PATCH_TRACE(tstep, 13);
            const int te = sm.mode[it.b * L + it.ego];
            const float4 b0 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8]);
            const float4 b1 = *reinterpret_cast<const float4*>(&sm.bq[te][hl * 32 + g * 8 + 4]);
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
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const float q8[8] = {qa[qt].x + b0.x, qa[qt].y + b0.y, qa[qt].z + b0.z, qa[qt].w + b0.w,
                                     qb[qt].x + b1.x, qb[qt].y + b1.y, qb[qt].z + b1.z, qb[qt].w + b1.w};
                split_pk8(q8, qhh[qt], qhl[qt]);
                m_run[qt] = -INFINITY;
                l_run[qt] = 0.f;
                o_acc[qt][0] = (float4v)(0.f);
                o_acc[qt][1] = (float4v)(0.f);
            }
            first = false;
            PATCH_TRACE(tstep, 11);
        }
        // ---- the step after S ----
        const bool last = rest == 0;
        PatchStep N;
        N.valid = 0; N.nk = 0;
        if (!last) {
            N = patch_describe(p, sm, it, ic, par, __builtin_ctz(rest), hl);
            rest &= rest - 1;
        } else if (nvalid) {
            pc_wg_barrier();                      // the next item's tables are complete
            icn = patch_item_consts(p, sm, itn);
            restn = pcs2_bits(p, itn, X, Y);
            N = patch_describe(p, sm, itn, icn, par ^ 1, __builtin_ctz(restn), hl);
            restn &= restn - 1;
        }
        PATCH_TRACE(tstep, 1);

        // ---- both patches of step S have landed when nothing of this wave is in flight any more (requested a step ago) ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PATCH_TRACE(tstep, 2);
        half8 khh[2], khl[2], vhh[2], vhl[2];
        {
            float kf[2][8];
            patch_blend(sm, S, kbase, &sm.bkv[S.tsel][0][hl * 32 + g * 8], lq, lxor, kf);
            split_pk8(kf[0], khh[0], khl[0]);
            split_pk8(kf[1], khh[1], khl[1]);
        }
        PATCH_TRACE(tstep, 3);
        {
            float vf[2][8];
            patch_blend(sm, S, vbase, &sm.bkv[S.tsel][1][hl * 32 + g * 8], lq, lxor, vf);
            half8 th[2], tl[2];
            split_pk8(vf[0], th[0], tl[0]);
            split_pk8(vf[1], th[1], tl[1]);
            // every tap of the step is in registers: the V' patch's bytes become the staging tiles [key][channel] (hi, lo)
            patch_wave_sync();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                *reinterpret_cast<half8*>(vth + (kt * 16 + lq) * VS + g * 8) = th[kt];
                *reinterpret_cast<half8*>(vtl + (kt * 16 + lq) * VS + g * 8) = tl[kt];
            }
            patch_wave_sync();
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // V^T tile dt takes the head's channels 8 q4 + 4 dt + r (q4 = lq & 3) as its rows 4 q4 + r (as k_attention_pcs2)
                const int off = (4 * g + (lq >> 2)) * VS + (lq & 3) * 8 + dt * 4;
#pragma unroll
                for (int hlx = 0; hlx < 2; ++hlx) {
                    const half_t* base = (hlx ? vtl : vth) + off;
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 16 * VS));
                    half8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (half_t)lo[e];
                        v[4 + e] = (half_t)hi[e];
                    }
                    if (hlx) vhl[dt] = v; else vhh[dt] = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // both patches have been read: their bytes may be replaced
        PATCH_TRACE(tstep, 4);
        PatchReq RN;
        if (N.valid) {
            RN = patch_offsets(p, sm, N, lane);
            patch_issue(p, N, RN, 0, 2, klds, vlds);
        }
        PATCH_TRACE(tstep, 5);

        // ---- per 16-query tile: S^T = K' Q^T + bias (+ mask), running maximum, exponentials as operand halves, O^T += V'^T P^T ----
        float4v madd[2];
        madd[0] = (float4v)(0.f); madd[1] = (float4v)(0.f);
        if (S.vis != 0xffffffffu) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) madd[kt][r] = ((S.vis >> (kt * 16 + 4 * g + r)) & 1u) ? 0.f : -INFINITY;
        }
#ifdef HMVIT_EXP_PATCH_NOMATH
        for (int qt = 0; qt < 4; ++qt) { o_acc[qt][0][0] += (float)vhh[0][0] * (float)khh[qt & 1][0]; o_acc[qt][1][0] += (float)vhl[1][0] + (float)khl[qt & 1][0]; l_run[qt] += 1.f; }
        if (false)
#endif
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            float4v s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const float4v b0 = biasf[qt - kt + 3], b1 = biasf[qt - kt + 1 >= 0 ? qt - kt + 1 : 0];
                float4v acc = (S.h & 1) ? b1 : b0;
                acc += madd[kt];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khl[kt], qhh[qt], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh[kt], qhl[qt], acc, 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(khh[kt], qhh[qt], acc, 0, 0, 0);
            }
            float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
            mx = fmaxf(mx, fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
            mx = max_over_lane_groups(mx);
            const float m_new = max_raw(m_run[qt], mx * LOG2E);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_safe);
            half8 ph, pl;
            float e8[8];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) e8[4 * kt + r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], LOG2E, -m_safe));
            split_pk8(e8, ph, pl);
            m_run[qt] = m_new;
            l_run[qt] = fmaf(l_run[qt], alpha, ((e8[0] + e8[1]) + (e8[2] + e8[3])) + ((e8[4] + e8[5]) + (e8[6] + e8[7])));
            o_acc[qt][0] *= alpha;
            o_acc[qt][1] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhl[dt], ph, o_acc[qt][dt], 0, 0, 0);
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhh[dt], pl, o_acc[qt][dt], 0, 0, 0);
                o_acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhh[dt], ph, o_acc[qt][dt], 0, 0, 0);
            }
            // the next step's rows, a quarter behind each of the first three tiles
            if (N.valid && qt < 3) patch_issue(p, N, RN, 2 * qt + 2, 2 * qt + 4, klds, vlds);
        }
        PATCH_TRACE(tstep, 9);

        if (last) {
            // ---- epilogue: normalise (row sums over the four lanes of a query column), 64 contiguous bytes per token, store ----
            float* outp = reinterpret_cast<float*>(p.out) + (size_t)(it.b * L + it.ego) * P * C;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                int row, col;
                token_pixel(HMVIT_PART_WINDOW, 8, X, Y, it.wx, it.wy, qt * 16 + lq, row, col);
                const float lsum = xor32_sum(xor16_sum(l_run[qt]));
                const float inv = 1.f / lsum;
                float a[4] = {o_acc[qt][0][0] * inv, o_acc[qt][0][1] * inv, o_acc[qt][0][2] * inv, o_acc[qt][0][3] * inv};
                float b[4] = {o_acc[qt][1][0] * inv, o_acc[qt][1][1] * inv, o_acc[qt][1][2] * inv, o_acc[qt][1][3] * inv};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    swap16_rows(a[e], b[e]);
                    swap32_rows(a[e], b[e]);
                }
                float* o = outp + (size_t)(row * W + col) * C + hl * 32 + 4 * g;
                *reinterpret_cast<float4*>(o) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(o + 16) = make_float4(b[0], b[1], b[2], b[3]);
                if (p.lse && g == 0)
                    p.lse[((size_t)(it.b * L + it.ego) * P + row * W + col) * (C / 32) + hl] = m_run[qt] * 0.6931471805599453f + logf(lsum);
            }
            PATCH_TRACE(tstep, 10);
            if (!nvalid) break;
            it = itn; ic = icn; rest = restn; par ^= 1;
            nvalid = pc_fetch(p, X, Y, 1, ego_fastest, cur, itn);
            first = true;
        }
        S = N;
        ++tstep;
    }
}

__global__ __launch_bounds__(512) void k_attention_patch(AttnParams p) {
    using SM = PatchShared;
    __shared__ __attribute__((aligned(16))) SM sm;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        const int C = p.C;       // 256
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * HMVIT_NUM_TYPES * 2 * 256; i += blockDim.x) {
            const int e = i / 512, pl = (i / 256) & 1, c = i % 256;
            sm.bkv[e][pl][c] = p.b_kv[(size_t)e * 2 * C + pl * C + c];
        }
        for (int i = threadIdx.x; i < HMVIT_NUM_TYPES * 256; i += blockDim.x) sm.bq[i / 256][i % 256] = p.b_q[(i / 256) * C + (i % 256)];
        if (threadIdx.x < kMaxSlots) {
            sm.mode[threadIdx.x] = p.mode[threadIdx.x];
            sm.cav[threadIdx.x] = p.cav[threadIdx.x];
            sm.ego_e[threadIdx.x] = p.ego_e[threadIdx.x];
        }
    }
    __syncthreads();
    {   // per (sample, ego): te | ev << 4 | self_vis << 8, and the (te, ts) pair of every chunk's source (as pcs2_fill_consts)
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
    patch_loop(p, sm, wave, threadIdx.x & 63);
}

static int launch_attn_patch(const AttnParams& p_in, hipStream_t st) {
    AttnParams p = p_in;
    if (const char* e = HMVIT_ENV("HMVIT_ATTN_TRACE")) p.trace = (unsigned long long*)strtoull(e, nullptr, 0);
    hipLaunchKernelGGL(k_attention_patch, dim3(kPcs2Grid), dim3(512), 0, st, p);
    HMVIT_CHECK_LAUNCH();
    return HMVIT_OK;
}
