"""Host-side weight preparation for libhmvit: the exact algebraic folds of SURVEY.md 8(a).

Runs once per parameter version (cached by the module), on whatever device the parameters
live on, with plain torch tensor algebra; nothing here is on the per-scene hot path.

Folds (all exact in real arithmetic, reference hetero_fusion.py line numbers):
  * q scale ``dim_head**-0.5`` (:217) goes into the q weights and bias;
  * ``relation_att[e]`` (:221-223) is applied to the K projection of source type ``ts`` for
    ego type ``te`` (e = te*2 + ts, :154-155): k' = blockdiag_h(W_att[e,h]) k;
  * ``relation_msg[e]`` (:263-264) is applied to the V projection: v' = blockdiag_h(W_msg[e,h]^T) v;
  * the relative-position bias table (:82-109, 227-233) is expanded into the accumulator
    fragment order of the 16x16 MFMA tile (row = key 4*(lane>>4)+r, col = query lane&15).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _lib

NUM_TYPES = _lib.NUM_TYPES


def bias_fragments(table: torch.Tensor, window: int, keep_graph: bool = False) -> torch.Tensor:
    """table ((2w-1)^2, heads) -> (heads, NB, 64, 4) f32; NB = 7 for window 8 (one fragment per
    query-tile minus key-tile offset -3..3), 1 for window 4."""
    w = window
    lane = torch.arange(64)
    ql = (lane & 15)[:, None].expand(64, 4)
    kl = (4 * (lane >> 4))[:, None] + torch.arange(4)[None, :]
    if w == 8:
        dq = torch.arange(-3, 4)[:, None, None]                 # qt - kt
        drow = 2 * dq + (ql >> 3)[None] - (kl >> 3)[None]
        dcol = ((ql & 7) - (kl & 7))[None].expand(7, 64, 4)
    elif w == 4:
        drow = ((ql >> 2) - (kl >> 2))[None]
        dcol = ((ql & 3) - (kl & 3))[None]
    else:
        raise ValueError(f"window_size={w} unsupported (4 or 8)")
    idx = (drow + w - 1) * (2 * w - 1) + (dcol + w - 1)          # (NB, 64, 4)
    frag = (table if keep_graph else table.detach()).float()[idx.to(table.device)]   # (NB, 64, 4, heads)
    return frag.permute(3, 0, 1, 2).contiguous()


def bias_dense(table: torch.Tensor, window: int, keep_graph: bool = False) -> torch.Tensor:
    """table ((2w-1)^2, heads) -> (heads, N, N) f32, N = w^2: bias[h][q][k] = table[(i_q - i_k + w - 1)(2w - 1) + (j_q - j_k + w - 1)][h]
    (hetero_fusion.py:82-109, 227-233) - the layout of the generic attention kernel (any window, any dim_head: csrc/attn.hip
    k_attention_any), where `bias_frag` of HmvitStageWeights carries this tensor instead of MFMA fragments."""
    w = window
    n = torch.arange(w * w)
    i, j = n // w, n % w
    idx = (i[:, None] - i[None, :] + w - 1) * (2 * w - 1) + (j[:, None] - j[None, :] + w - 1)       # (N, N) [q][k]
    dense = (table if keep_graph else table.detach()).float()[idx.to(table.device)]                   # (N, N, heads)
    return dense.permute(2, 0, 1).contiguous()


def generic_shape(window: int, dim_head: int) -> bool:
    """True for the shapes only the generic exact-f32 attention kernel serves (the tuned kernels: window 4 / 8, dim_head 32)."""
    return window not in (4, 8) or dim_head != 32


def store_row_order(r):
    """Channel (inside a 32-channel tile) computed by MFMA output row r of the projections whose results
    go to memory as f16 (img_q / img_kv).  An accumulator lane (token m, half hi) owns rows 8 j + 4 hi + i;
    with this order they are the channels 16 (j >> 1) + 8 hi + 4 (j & 1) + i, i.e. two runs of eight
    consecutive channels, so the lane stores 2 x 16 bytes instead of 4 x 8 (csrc/chain.hip k_ln_qkv)."""
    j, hi, i = r >> 3, (r >> 2) & 1, r & 3
    return 16 * (j >> 1) + 8 * hi + 4 * (j & 1) + i


# ---------------------------------------------------------------------------------------------
# range normalisation of the split-operand modes (HmvitStageScales, include/hmvit.h)
# ---------------------------------------------------------------------------------------------
TOP = 2.0 ** 14          # operands are carried at a power of two that puts their static bound just below this (f16 max 65504)


def pow2_floor(v: float) -> float:
    """Largest power of two <= v (1 for non-positive / non-finite v); clamped to 2^+-40 so that products of a few of them stay
    far inside the f32 range."""
    import math
    if not (v > 0.0) or math.isinf(v) or math.isnan(v):
        return 1.0
    return 2.0 ** max(-40, min(40, math.floor(math.log2(v))))


def scale_for(bound: float) -> float:
    """Power of two s with bound * s in (TOP / 2, TOP]."""
    return pow2_floor(TOP / bound) if bound > 0 else 1.0


def ln_bound(gamma: torch.Tensor, beta: torch.Tensor) -> float:
    """|LayerNorm(x)_c| <= sqrt(C) max|gamma| + max|beta|: the normalised row has 2-norm <= sqrt(C)."""
    C = gamma.shape[-1]
    return float(C ** 0.5 * gamma.abs().max() + beta.abs().max())


def linear_of_ln_bound(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias=None) -> float:
    """max_n |w_n . LayerNorm(x) + b_n| <= max_n (||w_n * gamma||_2 sqrt(C) + |w_n . beta| + |b_n|)  (Cauchy-Schwarz)."""
    C = w.shape[-1]
    b = (w * gamma).norm(dim=-1) * C ** 0.5 + (w @ beta).abs()
    if bias is not None:
        b = b + bias.abs()
    return float(b.max())


def split_halves(w: torch.Tensor):
    """x = hi + lo with hi = f16(x), lo = f16(x - hi): the operand pair of the split precision mode (22+ mantissa bits)."""
    hi = w.to(torch.float16)
    lo = (w - hi.to(w.dtype)).to(torch.float16)
    return hi, lo


def weight_image(w: torch.Tensor, store_rows: bool = False, linear_k: bool = False, split: bool = False) -> torch.Tensor:
    """(N, K) matrix -> (N/32, K/16, 64, 8) f16 fragment image (include/hmvit.h):
    img[t][kk][lane][4 jj + i] = W[32 t + row(lane & 31)][16 kk + 8 jj + 4 (lane >> 5) + i],
    row(r) = r, or store_row_order(r) for the images of k_ln_qkv.  ``linear_k`` (img_o, whose operand is
    loaded from memory rather than taken from an accumulator) uses the K index
    16 kk + 8 (lane >> 5) + 4 jj + i instead: 8 consecutive input channels per lane = one 16-byte load."""
    if split:
        # (N/32, K/16, 2, 64, 8): per k-step the fragment of the hi half followed by the fragment of the lo half
        hi, lo = split_halves(w.float())
        return torch.stack([weight_image(hi.float(), store_rows, linear_k), weight_image(lo.float(), store_rows, linear_k)],
                           dim=2).contiguous()
    N, K = w.shape
    if N % 32 or K % 16:
        raise ValueError(f"weight_image: ({N}, {K}) must be multiples of (32, 16)")
    dev = w.device
    t = torch.arange(N // 32, device=dev)[:, None, None, None]
    kk = torch.arange(K // 16, device=dev)[None, :, None, None]
    lane = torch.arange(64, device=dev)[None, None, :, None]
    q = torch.arange(8, device=dev)[None, None, None, :]
    r = lane & 31
    n = 32 * t + (store_row_order(r) if store_rows else r)
    if linear_k:
        k = 16 * kk + 8 * (lane >> 5) + q
    else:
        k = 16 * kk + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3)
    return w[n.expand(-1, K // 16, -1, 8), k.expand(N // 32, -1, -1, -1)].to(torch.float16).contiguous()


def weight_image16(w: torch.Tensor) -> torch.Tensor:
    """(N, K) matrix -> (N/16, K/32, 2, 64, 8) f16 split image of the 16-token kernels (csrc/chain.hip "x16", split mode, C = 256):
    img[T][s][half][lane][j] = W_half[16 T + (lane & 15)][32 s + 16 (j >> 2) + 4 (lane >> 4) + (j & 3)]
    (v_mfma_f32_16x16x32_f16 A operand; the k order matches the channels an accumulator lane owns)."""
    N, K = w.shape
    if N % 16 or K % 32:
        raise ValueError(f"weight_image16: ({N}, {K}) must be multiples of (16, 32)")
    dev = w.device
    T = torch.arange(N // 16, device=dev)[:, None, None, None]
    s = torch.arange(K // 32, device=dev)[None, :, None, None]
    lane = torch.arange(64, device=dev)[None, None, :, None]
    j = torch.arange(8, device=dev)[None, None, None, :]
    n = (16 * T + (lane & 15)).expand(-1, K // 32, -1, 8)
    k = (32 * s + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)).expand(N // 16, -1, -1, -1)
    hi, lo = split_halves(w.float())
    return torch.stack([hi[n, k], lo[n, k]], dim=2).contiguous()


def ffn_image16(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """x16 weight stream of Linear(C, C) -> GELU -> Linear(C, C): per hidden tile hc (32 hidden channels) the W_1 chunk (row tiles
    2 hc, 2 hc + 1 x all k-steps) followed by the W_2 slice chunk (all 16 row tiles x k-step hc): (C/32, 2, 16, 2, 64, 8)."""
    C = w1.shape[1]
    if tuple(w1.shape) != (C, C) or tuple(w2.shape) != (C, C) or C != 256:
        raise ValueError("the x16 FFN kernel needs mlp_dim == input_dim == 256")
    i1 = weight_image16(w1)                                   # (16, 8, 2, 64, 8)
    i2 = weight_image16(w2)
    a = i1.reshape(8, 2 * 8, 2, 64, 8)                        # hc: (row tile 2 hc .. 2 hc + 1) x k-step
    b = i2.permute(1, 0, 2, 3, 4).contiguous()                # (k-step hc, row tile T, half, 64, 8)
    return torch.stack([a, b], dim=1).contiguous()


def ffn_image(w1: torch.Tensor, w2: torch.Tensor, split: bool = False) -> torch.Tensor:
    """Interleaved weight stream of Linear(C, C) -> GELU -> Linear(C, C) for k_out_ffn:
    (C/32, 2, C/16, 64, 8): [hc][0] = image of W_1 rows of hidden tile hc, [hc][1] = fragments
    (t, 2 hc + s) of the image of W_2.  split: every fragment becomes a (hi, lo) pair, (C/32, 2, C/16, 2, 64, 8)."""
    C = w1.shape[1]
    if tuple(w1.shape) != (C, C) or tuple(w2.shape) != (C, C):
        raise ValueError("the fused FFN kernel needs mlp_dim == input_dim")
    i1 = weight_image(w1, split=split)
    i2 = weight_image(w2, split=split)
    nt, kk = C // 32, C // 16
    if split:
        i2 = i2.view(nt, nt, 2, 2, 64, 8).permute(1, 0, 2, 3, 4, 5).reshape(nt, kk, 2, 64, 8)
    else:
        i2 = i2.view(nt, nt, 2, 64, 8).permute(1, 0, 2, 3, 4).reshape(nt, kk, 64, 8)
    return torch.stack([i1, i2], dim=1).contiguous()


def fold_stage(sd: Dict[str, torch.Tensor], prefix: str, which: str, dim_head: int, window: int,
               dtype: torch.dtype, keep_graph: bool = False, split: bool = False, log2e: bool | None = None) -> Dict[str, torch.Tensor]:
    """Folded tensors of one stage (which = 'window' | 'grid') in the layout of
    HmvitStageWeights (include/hmvit.h).  Matrices in `dtype`, vectors in f32.  ``keep_graph`` (training, f32): `sd` holds
    the live parameters and the folds stay on the autograd tape, so the gradients the backward kernels return for the folded
    tensors flow back to relation_att / relation_msg / the typed Linears / the bias table through this very algebra."""
    prefix = f"{prefix}." if prefix else ""
    att = f"{prefix}{which}_attention"
    f = (lambda k: sd[k].float()) if keep_graph else (lambda k: sd[k].detach().float())
    C = f(f"{att}.q_linears.0.weight").shape[0]
    M = C // dim_head
    f16 = dtype == torch.float16       # fragment images (f16 mode, or split mode with hi / lo pairs)
    # the persistent attention kernel evaluates the softmax with exp2: log2(e) rides on the q scale and on the position bias
    LOG2E = 1.4426950408889634
    if log2e is None:
        log2e = f16 and not split
    scale = dim_head ** -0.5 * (LOG2E if log2e else 1.0)
    rel_att = f(f"{att}.relation_att")     # (4, M, d, d) [e, h, p, q]
    rel_msg = f(f"{att}.relation_msg")

    # ---- the folded tensors at their true scale, f32 ----
    raw: Dict[str, torch.Tensor] = {}
    stack = lambda fmt: torch.stack([f(fmt.format(t=t)) for t in range(NUM_TYPES)])
    raw["ln_gamma"] = stack(f"{prefix}{which}_norm.net.{{t}}.weight")
    raw["ln_beta"] = stack(f"{prefix}{which}_norm.net.{{t}}.bias")
    raw["w_q"] = stack(f"{att}.q_linears.{{t}}.weight") * scale
    raw["b_q"] = stack(f"{att}.q_linears.{{t}}.bias") * scale
    w_kv = torch.empty(NUM_TYPES, NUM_TYPES, 2 * C, C, device=rel_att.device)
    b_kv = torch.empty(NUM_TYPES, NUM_TYPES, 2 * C, device=rel_att.device)
    for te in range(NUM_TYPES):
        for ts in range(NUM_TYPES):
            e = te * NUM_TYPES + ts
            wk = f(f"{att}.k_linears.{ts}.weight").reshape(M, dim_head, C)
            bk = f(f"{att}.k_linears.{ts}.bias").reshape(M, dim_head)
            wv = f(f"{att}.v_linears.{ts}.weight").reshape(M, dim_head, C)
            bv = f(f"{att}.v_linears.{ts}.bias").reshape(M, dim_head)
            w_kv[te, ts, :C] = torch.einsum("hpq,hqc->hpc", rel_att[e], wk).reshape(C, C)
            b_kv[te, ts, :C] = torch.einsum("hpq,hq->hp", rel_att[e], bk).reshape(C)
            w_kv[te, ts, C:] = torch.einsum("hpq,hpc->hqc", rel_msg[e], wv).reshape(C, C)
            b_kv[te, ts, C:] = torch.einsum("hpq,hp->hq", rel_msg[e], bv).reshape(C)
    raw["w_kv"], raw["b_kv"] = w_kv, b_kv
    if generic_shape(window, dim_head):
        if f16 or split:
            raise ValueError(f"window_size={window} / dim_head={dim_head}: generic shapes run on the exact-f32 kernels only")
        # (training too: the dense table stays on the autograd tape and k_attention_any_bwd returns its gradient in this layout)
        raw["bias_frag"] = bias_dense(sd[f"{att}.relative_position_bias_table.weight"], window, keep_graph)
    else:
        raw["bias_frag"] = bias_fragments(sd[f"{att}.relative_position_bias_table.weight"], window, keep_graph) * (LOG2E if log2e else 1.0)
    raw["w_o"] = stack(f"{att}.a_linears.{{t}}.0.weight")
    raw["b_o"] = stack(f"{att}.a_linears.{{t}}.0.bias")
    raw["ffn_ln_gamma"] = stack(f"{prefix}{which}_ffd.norm.net.{{t}}.weight")
    raw["ffn_ln_beta"] = stack(f"{prefix}{which}_ffd.norm.net.{{t}}.bias")
    raw["w_1"] = stack(f"{prefix}{which}_ffd.fn.net.{{t}}.0.weight")
    raw["w_2"] = stack(f"{prefix}{which}_ffd.fn.net.{{t}}.3.weight")
    raw["b_1"] = stack(f"{prefix}{which}_ffd.fn.net.{{t}}.0.bias")
    raw["b_2"] = stack(f"{prefix}{which}_ffd.fn.net.{{t}}.3.bias")

    out: Dict[str, torch.Tensor] = {}
    if split:
        # split-operand modes: every tensor goes to the kernels at the power of two stage_scales() assigns it (exact)
        sc, mul = stage_scales(raw, planes_scaled=not log2e)
        for k, m in mul.items():
            raw[k] = raw[k] * m
        out["scales"] = sc
    per_type = lambda m, fn: torch.stack([fn(m[t]) for t in range(NUM_TYPES)])
    x16 = split and C == 256          # 16-token split kernels: their own image layout (weight_image16)
    w_q, w_kv, w_o, w_1, w_2 = raw["w_q"], raw["w_kv"], raw["w_o"], raw["w_1"], raw["w_2"]
    if x16:
        out["img_q"] = per_type(w_q, weight_image16)
        out["img_kv"] = torch.stack([per_type(w_kv[te], weight_image16) for te in range(NUM_TYPES)])
        out["img_o"] = per_type(w_o, weight_image16)
        out["img_ffn"] = torch.stack([ffn_image16(w_1[t], w_2[t]) for t in range(NUM_TYPES)])
    elif f16:
        out["img_q"] = per_type(w_q, lambda m: weight_image(m, store_rows=True, split=split))
        out["img_kv"] = torch.stack([per_type(w_kv[te], lambda m: weight_image(m, store_rows=True, split=split)) for te in range(NUM_TYPES)])
        out["img_o"] = per_type(w_o, lambda m: weight_image(m, linear_k=True, split=split))
        out["img_ffn"] = torch.stack([ffn_image(w_1[t], w_2[t], split=split) for t in range(NUM_TYPES)])
    else:
        out["w_q"], out["w_kv"], out["w_o"], out["w_1"], out["w_2"] = w_q, w_kv, w_o, w_1, w_2
    for k in ("ln_gamma", "ln_beta", "b_q", "b_kv", "bias_frag", "b_o", "ffn_ln_gamma", "ffn_ln_beta", "b_1", "b_2"):
        out[k] = raw[k]
    if keep_graph:
        # the backward kernel also forms the un-transposed logit tiles: their bias is the fragment set of the table with
        # negated offsets, i.e. the table flipped along its first axis (index (dr + w - 1)(2w - 1) + dc + w - 1)
        if generic_shape(window, dim_head):
            out["bias_frag_neg"] = torch.zeros(1, dtype=torch.float32, device=sd[f"{att}.relative_position_bias_table.weight"].device)   # unused
        else:
            out["bias_frag_neg"] = bias_fragments(sd[f"{att}.relative_position_bias_table.weight"].detach().flip(0), window)
    return {k: (v.contiguous() if torch.is_tensor(v) else v) for k, v in out.items()}


def stage_scales(raw: Dict[str, torch.Tensor], planes_scaled: bool):
    """Power-of-two range normalisation of one stage for the split-operand modes (HmvitStageScales, include/hmvit.h).
    raw: the folded f32 tensors of fold_stage (ln_*, w_q, b_q, w_kv, b_kv, bias_frag, w_o, b_o, ffn_ln_*, w_1, b_1, w_2, b_2).
    Returns (scales dict of python floats / nested lists, dict of per-tensor multipliers to apply before building the images).
    planes_scaled: the Q / K' / V' / O planes are f32 and carried at their own power of two (split mode); False = true-scale
    planes (mixed mode: f16 planes read by the f16 attention kernels)."""
    T = NUM_TYPES
    C = raw["w_q"].shape[-1]
    g_n, b_n, g_f, b_f = raw["ln_gamma"], raw["ln_beta"], raw["ffn_ln_gamma"], raw["ffn_ln_beta"]
    a_n = [scale_for(ln_bound(g_n[t], b_n[t])) for t in range(T)]
    a_f = [scale_for(ln_bound(g_f[t], b_f[t])) for t in range(T)]
    wmax = lambda m: float(m.abs().max())
    wq_s = [scale_for(wmax(raw["w_q"][t])) for t in range(T)]
    wk_s = [[scale_for(wmax(raw["w_kv"][te, ts, :C])) for ts in range(T)] for te in range(T)]
    wv_s = [[scale_for(wmax(raw["w_kv"][te, ts, C:])) for ts in range(T)] for te in range(T)]
    wo_s = [scale_for(wmax(raw["w_o"][t])) for t in range(T)]
    w1_s = [scale_for(wmax(raw["w_1"][t])) for t in range(T)]
    w2_s = [scale_for(wmax(raw["w_2"][t])) for t in range(T)]
    if planes_scaled:
        s_q = [scale_for(linear_of_ln_bound(raw["w_q"][t], g_n[t], b_n[t], raw["b_q"][t])) for t in range(T)]
        s_k_free = [[scale_for(linear_of_ln_bound(raw["w_kv"][te, ts, :C], g_n[ts], b_n[ts], raw["b_kv"][te, ts, :C]))
                     for ts in range(T)] for te in range(T)]
        # one logit scale per stage (the position bias is a single tensor and one softmax mixes all source types)
        s_qk = min(s_q[te] * s_k_free[te][ts] for te in range(T) for ts in range(T))
        s_k = [[s_qk / s_q[te] for ts in range(T)] for te in range(T)]
        # one value scale per ego type (the attention output sums over the source types)
        s_v = [min(scale_for(linear_of_ln_bound(raw["w_kv"][te, ts, C:], g_n[ts], b_n[ts], raw["b_kv"][te, ts, C:]))
                   for ts in range(T)) for te in range(T)]
    else:
        s_q, s_k, s_v, s_qk = [1.0] * T, [[1.0] * T for _ in range(T)], [1.0] * T, 1.0
    h_bound = [linear_of_ln_bound(raw["w_1"][t], g_f[t], b_f[t], raw["b_1"][t]) for t in range(T)]
    s_g = [scale_for(h_bound[t]) for t in range(T)]
    sc = {
        "c_q": [s_q[t] / (wq_s[t] * a_n[t]) for t in range(T)],
        "c_k": [[s_k[te][ts] / (wk_s[te][ts] * a_n[ts]) for ts in range(T)] for te in range(T)],
        "c_v": [[s_v[te] / (wv_s[te][ts] * a_n[ts]) for ts in range(T)] for te in range(T)],
        "k_logit": 1.0 / s_qk,
        "c_o": [1.0 / (s_v[t] * wo_s[t]) for t in range(T)],
        "c_1": [1.0 / (a_f[t] * w1_s[t]) for t in range(T)],
        "s_g": s_g,
        "k_2": [w2_s[t] * s_g[t] for t in range(T)],
    }
    dev = raw["w_q"].device
    vec = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
    mul = {
        "ln_gamma": vec(a_n)[:, None], "ln_beta": vec(a_n)[:, None],
        "ffn_ln_gamma": vec(a_f)[:, None], "ffn_ln_beta": vec(a_f)[:, None],
        "w_q": vec(wq_s)[:, None, None], "b_q": vec(s_q)[:, None],
        "w_kv": torch.cat([vec(wk_s)[:, :, None, None].expand(T, T, C, 1), vec(wv_s)[:, :, None, None].expand(T, T, C, 1)], dim=2),
        "b_kv": torch.cat([vec(s_k)[:, :, None].expand(T, T, C), vec(s_v)[:, None, None].expand(T, T, C)], dim=2),
        "bias_frag": vec(s_qk),
        "w_o": vec(wo_s)[:, None, None], "b_o": 1.0 / vec(sc["c_o"])[:, None],
        "w_1": vec(w1_s)[:, None, None], "b_1": 1.0 / vec(sc["c_1"])[:, None],
        "w_2": vec(w2_s)[:, None, None], "b_2": vec(sc["k_2"])[:, None],
    }
    return sc, mul


def head_scales(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor):
    """HmvitHeadScales + the multipliers of the two mlp_head weight matrices."""
    T = NUM_TYPES
    w1_s = [scale_for(float(w1[t].abs().max())) for t in range(T)]
    w2_s = [scale_for(float(w2[t].abs().max())) for t in range(T)]
    sc = {"w1": w1_s, "w2": w2_s, "l1": [float(w1[t].abs().sum(-1).max()) for t in range(T)],
          "b1max": [float(b1[t].abs().max()) for t in range(T)]}
    return sc, w1_s, w2_s


def fold_head(sd: Dict[str, torch.Tensor], prefix: str, dtype: torch.dtype, keep_graph: bool = False, split: bool = False) -> Dict[str, torch.Tensor]:
    f = (lambda k: sd[k].float()) if keep_graph else (lambda k: sd[k].detach().float())
    stack = lambda fmt: torch.stack([f(fmt.format(t=t)) for t in range(NUM_TYPES)])
    w1, w2 = stack(f"{prefix}.net.{{t}}.0.weight"), stack(f"{prefix}.net.{{t}}.3.weight")
    out = {"head_b1": stack(f"{prefix}.net.{{t}}.0.bias").contiguous(),
           "head_b2": stack(f"{prefix}.net.{{t}}.3.bias").contiguous()}
    if split:
        sc, w1_s, w2_s = head_scales(w1, out["head_b1"], w2)
        out["head_scales"] = sc
        w1 = w1 * torch.tensor(w1_s, dtype=w1.dtype, device=w1.device)[:, None, None]
        w2 = w2 * torch.tensor(w2_s, dtype=w2.dtype, device=w2.device)[:, None, None]
    if dtype == torch.float16:
        if split and w1.shape[-1] == 256:
            out["head_img_ffn"] = torch.stack([ffn_image16(w1[t], w2[t]) for t in range(NUM_TYPES)])
        else:
            out["head_img_ffn"] = torch.stack([ffn_image(w1[t], w2[t], split=split) for t in range(NUM_TYPES)])
    else:
        out["head_w1"], out["head_w2"] = w1.contiguous(), w2.contiguous()
    return out
