"""CPU oracle for the HM-ViT fusion hot path (TEST INFRASTRUCTURE, not product code).

This file is a plain-PyTorch, fp32, CPU restatement of the reference algorithm for the
path BASELINE.json's north_star names: ``HeteroFusion.forward`` and everything below it
(reference: opencood/models/bevformer_point_pillar_hetero.py:22-49).  It is written from
the reference's *behaviour* (the order of operations the reference performs: typed
LayerNorm -> L^2 BEV warps -> per-ego window / dilated-grid attention with relation
matrices -> typed FFN), vectorised instead of the reference's Python loops, and it takes
the reference's own ``state_dict`` (same key names) so goldens can be replayed.

Who may import this module: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the CPU baseline only.  The
shipped path (``hm-vit_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_goldens.py`` imports the reference from
/root/reference in the build container, runs it on seeded inputs and freezes
inputs+weights+outputs as .npz; ``tests/test_oracle_golden.py`` replays them through this
file.  The reference ships no test of its own for this path.

Third-party arithmetic used exactly as the reference uses it (torch library calls, present
on the GPU box as well): ``F.layer_norm``, ``F.gelu`` (erf), ``F.affine_grid`` +
``F.grid_sample`` (bilinear / nearest, zeros padding, align_corners=True), ``softmax``.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NUM_TYPES = 2  # camera = 0, lidar = 1 (base_camera_lidar_dataset.py:136,178)


# --------------------------------------------------------------------------------------
# typed (per-agent-type) token-wise layers
# --------------------------------------------------------------------------------------
def hetero_layer_norm(x: Tensor, mode: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """HeteroLayerNorm (base_transformer.py:138-177): LayerNorm over the channel axis, eps
    1e-5, with the affine parameters of the agent's type.  x: (B, L, ..., C), mode: (B, L)."""
    C = x.shape[-1]
    out = torch.zeros_like(x)
    for t in range(NUM_TYPES):
        sel = mode == t
        if sel.any():
            out[sel] = F.layer_norm(x[sel], (C,), sd[f"{prefix}.net.{t}.weight"],
                                    sd[f"{prefix}.net.{t}.bias"], 1e-5)
    return out


def hetero_feed_forward(x: Tensor, mode: Tensor, sd: Dict[str, Tensor], prefix: str,
                        mask_hidden: Tensor | None = None, mask_out: Tensor | None = None) -> Tensor:
    """HeteroFeedForward (base_transformer.py:180-192): Linear -> GELU(erf) -> Dropout -> Linear -> Dropout, weights
    picked by agent type.  x: (B, L, ..., C).  Eval mode by default; a training-mode run is replayed by passing the two
    Dropout masks (0 or 1/(1-p), shaped like the hidden / output activations)."""
    out = None
    for t in range(NUM_TYPES):
        sel = mode == t
        if not sel.any():
            continue
        h = F.linear(x[sel], sd[f"{prefix}.net.{t}.0.weight"], sd[f"{prefix}.net.{t}.0.bias"])
        h = F.gelu(h)
        if mask_hidden is not None:
            h = h * mask_hidden[sel]
        h = F.linear(h, sd[f"{prefix}.net.{t}.3.weight"], sd[f"{prefix}.net.{t}.3.bias"])
        if mask_out is not None:
            h = h * mask_out[sel]
        if out is None:
            out = torch.zeros(x.shape[:-1] + (h.shape[-1],), dtype=x.dtype, device=x.device)
        out[sel] = h
    return out


# --------------------------------------------------------------------------------------
# BEV warp (torch_transformation_utils.py)
# --------------------------------------------------------------------------------------
def pixel_affine(t_matrix: Tensor, discrete_ratio: float, downsample_rate: float,
                 H: int, W: int) -> Tensor:
    """(..., 4, 4) metric transforms -> (..., 2, 3) pixel-space affine A with
    dst = A [u, v, 1]^T: rows/cols {0,1}x{0,1,3}, translation divided by
    discrete_ratio*downsample_rate (torch_transformation_utils.py:108-134), rotation taken
    about (W/2, H/2) and the translation added afterwards (:254-297)."""
    m = t_matrix[..., [0, 1], :][..., [0, 1, 3]].to(torch.float32).clone()
    m[..., 2] = m[..., 2] / (discrete_ratio * downsample_rate)
    cx, cy = W / 2, H / 2
    R = m[..., :2]
    A = torch.zeros_like(m)
    A[..., :2] = R
    # S(c) R S(-c): translation part = c - R c
    A[..., 0, 2] = cx - (R[..., 0, 0] * cx + R[..., 0, 1] * cy)
    A[..., 1, 2] = cy - (R[..., 1, 0] * cx + R[..., 1, 1] * cy)
    A[..., 2] = A[..., 2] + m[..., 2]
    return A


def warp_affine(src: Tensor, A: Tensor, mode: str = "bilinear") -> Tensor:
    """Resample src (N, C, H, W) so that dst(u, v) = src(A^-1 [u, v, 1]); zeros outside,
    align_corners=True.  Follows the reference's chain literally
    (torch_transformation_utils.py:317-355): lift A to 3x3, conjugate with the
    pixel->[-1, 1] normalisation, invert in fp32, ``affine_grid`` + ``grid_sample``."""
    N, C, H, W = src.shape
    A = A.cpu()                      # the sampling geometry is ALWAYS the CPU's fp32 arithmetic (see `hetero_fusion(device=)`)
    M = torch.zeros(N, 3, 3, dtype=A.dtype)
    M[:, :2] = A
    M[:, 2, 2] = 1.0
    norm = torch.tensor([[2.0 / (W - 1.0) if W > 1 else 2.0 / 1e-14, 0.0, -1.0],
                         [0.0, 2.0 / (H - 1.0) if H > 1 else 2.0 / 1e-14, -1.0],
                         [0.0, 0.0, 1.0]], dtype=A.dtype)[None]
    dst_norm_from_src_norm = norm @ (M @ torch.inverse(norm))
    src_norm_from_dst_norm = torch.inverse(dst_norm_from_src_norm)
    grid = F.affine_grid(src_norm_from_dst_norm[:, :2], [N, C, H, W], align_corners=True)
    # (the sampling positions are always the reference's fp32 ones; a float64 `src` - the "truth" runs of the precision
    # stress tests - only changes the arithmetic of the blend)
    return F.grid_sample(src, grid.to(device=src.device, dtype=src.dtype), mode=mode, padding_mode="zeros", align_corners=True)


def warp_agents(x: Tensor, t_to_target: Tensor, discrete_ratio: float,
                downsample_rate: float) -> Tensor:
    """SpatialTransformation.forward (spatial_transformation.py:16-44): warp every agent map
    x[b, l] (B, L, C, H, W) with its transform t_to_target[b, l] (B, L, 4, 4)."""
    B, L, C, H, W = x.shape
    A = pixel_affine(t_to_target.cpu(), discrete_ratio, downsample_rate, H, W).reshape(-1, 2, 3)
    return warp_affine(x.reshape(-1, C, H, W), A).reshape(B, L, C, H, W)


def roi_and_cav_mask(H: int, W: int, cav_mask: Tensor, t_to_target: Tensor,
                     discrete_ratio: float, downsample_rate: float) -> Tensor:
    """get_roi_and_cav_mask (torch_transformation_utils.py:11-105): nearest-neighbour warp of
    an all-ones map (which target pixels see source agent l) times agent validity.
    Returns (B, H, W, 1, L) float."""
    B, L = t_to_target.shape[:2]
    A = pixel_affine(t_to_target.cpu(), discrete_ratio, downsample_rate, H, W).reshape(-1, 2, 3)
    ones = torch.ones(B * L, 1, H, W, dtype=A.dtype)
    roi = warp_affine(ones, A, mode="nearest").reshape(B, L, 1, H, W)          # nearest-mode visibility: CPU, whatever the device
    com = roi * cav_mask.cpu().reshape(B, L, 1, 1, 1)
    return com.permute(0, 3, 4, 2, 1)


# --------------------------------------------------------------------------------------
# H3GAT attention (hetero_fusion.py:16-277)
# --------------------------------------------------------------------------------------
def relative_position_index(w: int) -> Tensor:
    """(n, n) index into the ((2w-1)^2, heads) bias table: (dr + w-1)(2w-1) + (dc + w-1),
    token order row-major inside the window (hetero_fusion.py:82-109)."""
    r = torch.arange(w)
    rr, cc = torch.meshgrid(r, r, indexing="ij")
    rr, cc = rr.reshape(-1), cc.reshape(-1)
    dr = rr[:, None] - rr[None, :] + (w - 1)
    dc = cc[:, None] - cc[None, :] + (w - 1)
    return dr * (2 * w - 1) + dc


def hetero_attention(xw: Tensor, mode: Tensor, mask: Tensor, sd: Dict[str, Tensor],
                     prefix: str, dim_head: int, window: int,
                     return_intermediates: bool = False):
    """HeteroAttention.forward (hetero_fusion.py:187-277), exclude_self=False.

    xw   (B, L, X, Y, w, w, C)  partitioned, already-warped, already-normalised features,
                                 agent 0 is the ego of this call
    mode (B, L) int              agent types in the same (ego-first) order
    mask (B, X, Y, w, w, 1, L)   1 = key visible
    ->   (B, 1, X, Y, w, w, C)   ego update (before the residual)
    """
    B, L, X, Y, w1, w2, C = xw.shape
    mode = mode.to(xw.device)
    M = C // dim_head
    n = w1 * w2
    scale = dim_head ** -0.5

    # typed q/k/v projections (:111-140); only the ego's q is kept (:200)
    def typed_linear(name: str, inp: Tensor, types: Tensor) -> Tensor:
        out = torch.zeros_like(inp)
        for t in range(NUM_TYPES):
            sel = types == t
            if sel.any():
                out[sel] = F.linear(inp[sel], sd[f"{prefix}.{name}.{t}.weight"],
                                    sd[f"{prefix}.{name}.{t}.bias"])
        return out

    q = typed_linear("q_linears", xw[:, :1], mode[:, :1])
    k = typed_linear("k_linears", xw, mode)
    v = typed_linear("v_linears", xw, mode)

    # (B, X, Y, M, l, n, d)
    def heads(t: Tensor) -> Tensor:
        l = t.shape[1]
        return t.reshape(B, l, X, Y, n, M, dim_head).permute(0, 2, 3, 5, 1, 4, 6)

    q, k, v = heads(q) * scale, heads(k), heads(v)

    # relation matrices of the ego row: e = type_ego * 2 + type_src (:154-185, 209-210)
    rel = mode[:, :1].long() * NUM_TYPES + mode.long()            # (B, L)
    w_att = sd[f"{prefix}.relation_att"][rel]                      # (B, L, M, d, d)
    w_msg = sd[f"{prefix}.relation_msg"][rel]
    # k'[b,x,y,h,z,e,p] = sum_q W_att[b,z,h,p,q] k[...,q]  ;  sim = q . k'   (:221-223)
    k_rel = torch.einsum("bzhpq,bxyhzeq->bxyhzep", w_att, k)
    sim = torch.einsum("bxyhcp,bxyhzep->bxyhcze", q[:, :, :, :, 0], k_rel)  # (B,X,Y,M,n,L,n)

    # relative position bias, identical for every source agent (:227-233)
    table = sd[f"{prefix}.relative_position_bias_table.weight"]    # ((2w-1)^2, M)
    bias = table[relative_position_index(window).to(table.device)].permute(2, 0, 1)  # (M, n, n)
    sim = sim + bias[None, None, None, :, :, None, :]

    # key mask after the bias, -inf fill, softmax over all L*n keys (:243-251)
    key_mask = mask.to(sim.device).reshape(B, X, Y, n, L).permute(0, 1, 2, 4, 3)   # (B,X,Y,L,n)
    sim = sim.masked_fill(key_mask[:, :, :, None, None] == 0, -float("inf"))
    sim_flat = sim.reshape(B, X, Y, M, n, L * n)
    attn = torch.softmax(sim_flat, dim=-1).reshape(B, X, Y, M, n, L, n)

    # messages: v'[...,q] = sum_p W_msg[b,z,h,p,q] v[...,p]  (:263-267)
    v_msg = torch.einsum("bzhpq,bxyhzep->bxyhzeq", w_msg, v)
    out = torch.einsum("bxyhcze,bxyhzeq->bxyhcq", attn, v_msg)      # (B,X,Y,M,n,d)
    out = out.permute(0, 1, 2, 4, 3, 5).reshape(B, 1, X, Y, w1, w2, C)

    # typed output projection of the ego rows (:142-152); dropout is identity in eval
    res = torch.zeros_like(out)
    for t in range(NUM_TYPES):
        sel = mode[:, :1] == t
        if sel.any():
            res[sel] = F.linear(out[sel], sd[f"{prefix}.a_linears.{t}.0.weight"],
                                sd[f"{prefix}.a_linears.{t}.0.bias"])
    if return_intermediates:
        return res, sim_flat, attn.reshape(B, X, Y, M, n, L * n)
    return res


# --------------------------------------------------------------------------------------
# HeteroFusionBlock (hetero_fusion.py:279-474)
# --------------------------------------------------------------------------------------
def _partition(t: Tensor, w: int, grid: bool) -> Tensor:
    """(B, L, C, H, W) -> (B, L, X, Y, w, w, C).  Local: contiguous w x w windows
    '(x w1) (y w2)' (:387-389); global: dilated grid '(w1 x) (w2 y)' (:430-431)."""
    B, L, C, H, W = t.shape
    X, Y = H // w, W // w
    if grid:
        t = t.reshape(B, L, C, w, X, w, Y).permute(0, 1, 4, 6, 3, 5, 2)
    else:
        t = t.reshape(B, L, C, X, w, Y, w).permute(0, 1, 3, 5, 4, 6, 2)
    return t


def _unpartition(t: Tensor, grid: bool) -> Tensor:
    """(B, l, X, Y, w, w, C) -> (B, l, C, H, W), inverse of _partition."""
    B, l, X, Y, w, _, C = t.shape
    if grid:
        t = t.permute(0, 1, 6, 4, 2, 5, 3)
    else:
        t = t.permute(0, 1, 6, 2, 4, 3, 5)
    return t.reshape(B, l, C, X * w, Y * w)


def fusion_stage(x: Tensor, pairwise_t: Tensor, mask: Tensor, mode: Tensor,
                 record_len: Tensor, sd: Dict[str, Tensor], prefix: str, which: str,
                 cfg: dict, drop=None) -> Tensor:
    """local_/global_spatial_multi_agent_attention (hetero_fusion.py:363-444).
    which = 'window' (local) or 'grid' (global).  drop: None (eval) or the three Dropout masks of a training-mode run,
    (B, L, H, W, C) each: [0] after the attention out-projection (hetero_fusion.py:65-66), [1] FFN hidden, [2] FFN output
    (base_transformer.py:186-192)."""
    B, L, C, H, W = x.shape
    w = cfg["window_size"]
    dr = cfg["spatial_transform"]["voxel_size"][0]
    ds = cfg["spatial_transform"]["downsample_rate"]
    grid = which == "grid"

    xn = hetero_layer_norm(x.permute(0, 1, 3, 4, 2), mode, sd,
                           f"{prefix}.{which}_norm").permute(0, 1, 4, 2, 3)
    max_cav = int(record_len.max())
    updates = []
    for i in range(max_cav):
        # every source l warped into ego i's frame with T[b, l, i]  (:338-361)
        t_li = pairwise_t[:, :, i]
        xi = warp_agents(xn, t_li, dr, ds)[:, :max_cav]
        mi = roi_and_cav_mask(H, W, mask, t_li, dr, ds)[..., :max_cav]  # (B,H,W,1,l)
        order = [i] + [j for j in range(max_cav) if j != i]               # (:329-336)
        xi, mi, mode_i = xi[:, order], mi[..., order], mode[:, :max_cav][:, order]
        xw = _partition(xi, w, grid)
        X, Y = H // w, W // w
        if grid:
            mw = mi.reshape(B, w, X, w, Y, 1, max_cav).permute(0, 2, 4, 1, 3, 5, 6)
        else:
            mw = mi.reshape(B, X, w, Y, w, 1, max_cav).permute(0, 1, 3, 2, 4, 5, 6)
        upd = hetero_attention(xw, mode_i, mw, sd, f"{prefix}.{which}_attention",
                               cfg["dim_head"], w)
        updates.append(_unpartition(upd, grid))
    upd = torch.cat(updates, dim=1)
    if drop is not None:
        upd = upd * drop[0][:, :max_cav].permute(0, 1, 4, 2, 3)
    upd = F.pad(upd, (0, 0, 0, 0, 0, 0, 0, L - max_cav))
    x = upd + x                                                            # (:399,439)
    xt = x.permute(0, 1, 3, 4, 2)
    y = hetero_feed_forward(hetero_layer_norm(xt, mode, sd, f"{prefix}.{which}_ffd.norm"),
                            mode, sd, f"{prefix}.{which}_ffd.fn",
                            None if drop is None else drop[1], None if drop is None else drop[2]) + xt      # (:401,441)
    return y.permute(0, 1, 4, 2, 3)


def split_attn(branches, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """SplitAttn (fusion_modules/split_attn.py:32-67), radix = len(branches): global average
    pool over H, W of the branch sum -> fc1 (no bias) -> LayerNorm -> ReLU -> fc2 (no bias)
    -> softmax across branches per channel -> weighted sum.  branches: [(B, L, H, W, C)]."""
    B, L, H, W, C = branches[0].shape
    r = len(branches)
    gap = sum(branches).mean((2, 3), keepdim=True)
    g = F.linear(gap, sd[f"{prefix}.fc1.weight"])
    g = F.relu(F.layer_norm(g, (C,), sd[f"{prefix}.bn1.weight"], sd[f"{prefix}.bn1.bias"], 1e-5))
    a = F.linear(g, sd[f"{prefix}.fc2.weight"]).reshape(B, L, 1, 1, r, C)
    a = torch.softmax(a, dim=4)
    return sum(branches[i] * a[:, :, :, :, i] for i in range(r))


def hetero_fusion_block(x: Tensor, pairwise_t: Tensor, mode: Tensor, record_len: Tensor,
                        mask: Tensor, sd: Dict[str, Tensor], prefix: str, cfg: dict, drop=None) -> Tensor:
    """HeteroFusionBlock.forward (hetero_fusion.py:446-474).  drop: None or [window masks, grid masks] (fusion_stage)."""
    arch = cfg["architect_mode"]
    if arch == "sequential":
        x = fusion_stage(x, pairwise_t, mask, mode, record_len, sd, prefix, "window", cfg, None if drop is None else drop[0])
        x = fusion_stage(x, pairwise_t, mask, mode, record_len, sd, prefix, "grid", cfg, None if drop is None else drop[1])
        return x
    if arch == "parallel":
        a = fusion_stage(x, pairwise_t, mask, mode, record_len, sd, prefix, "window", cfg)
        b = fusion_stage(x, pairwise_t, mask, mode, record_len, sd, prefix, "grid", cfg)
        y = split_attn([a.permute(0, 1, 3, 4, 2), b.permute(0, 1, 3, 4, 2)], sd,
                       f"{prefix}.split_attn")
        return y.permute(0, 1, 4, 2, 3)
    raise ValueError(f"{arch} not implemented")


def hetero_fusion(x: Tensor, pairwise_t: Tensor, mode: Tensor, record_len: Tensor,
                  mask: Tensor, sd: Dict[str, Tensor], cfg: dict, drop_masks=None,
                  dtype: torch.dtype = torch.float32, device=None) -> Tensor:
    """HeteroFusion.forward (bevformer_point_pillar_hetero.py:39-49).  Plain torch, so torch.autograd differentiates it:
    the gradient checker of the HIP backward pass.  drop_masks: None (eval) or, per iteration, [window, grid] triples of
    Dropout masks replaying a training-mode run (fusion_stage).

    x (B, L, C, H, W) f32; pairwise_t (B, L, L, 4, 4), [b, i, j] maps agent i -> agent j;
    mode (B, L) int 1 = lidar / 0 = camera (padding 0); record_len (B,); mask (B, L) 1/0.
    Returns (B, C, H, W).  dtype=torch.float64 evaluates the same network on the same fp32 sampling geometry in double
    precision: the yardstick that tells fp32 round-off of the reference apart from error of the implementation under test.
    device: where the token arithmetic (LayerNorm, Linears, einsums, softmax, the bilinear blend) runs; default = x's device
    (the CPU in every pinned comparison).  The GPU tests pass "cuda" for the float64 yardstick at the sizes where the CPU
    needs minutes: the sampling positions (fp32 `affine_grid`) and the nearest-mode visibility masks are ALWAYS computed on
    the CPU and copied over, so only the double-precision arithmetic moves
    (tests/test_hip_range.py::test_float64_yardstick_is_the_same_on_both_devices holds the two to 1e-12)."""
    device = x.device if device is None else torch.device(device)
    sd = {k: (v.to(dtype) if v.is_floating_point() else v).to(device) for k, v in sd.items()}
    mode = mode.to(torch.int64).to(device)
    pairwise_t = pairwise_t.to(torch.float32).cpu()
    mask, record_len = mask.cpu(), record_len.cpu()
    x = x.to(device=device, dtype=dtype)
    if drop_masks is not None:
        drop_masks = [[[m.to(device=device, dtype=dtype) for m in stage] for stage in it] for it in drop_masks]
    for it in range(cfg["num_iters"]):
        x = hetero_fusion_block(x, pairwise_t, mode, record_len, mask, sd,
                                "hetero_fusion_block", cfg["hetero_fusion_block"],
                                None if drop_masks is None else drop_masks[it])
    ego = x[:, :1].permute(0, 1, 3, 4, 2)                      # (B, 1, H, W, C)
    y = hetero_feed_forward(ego, mode[:, :1], sd, "mlp_head")
    return y[:, 0].permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# synthetic scenes -- shared by tests and bench so that the oracle
# and the HIP path always see identical inputs
# --------------------------------------------------------------------------------------
def make_config(C: int, window: int, L: int, voxel: float = 0.4, downsample: int = 4,
                num_iters: int = 2, dim_head: int = 32, arch: str = "sequential",
                mlp_dim: int | None = None) -> dict:
    st = {"downsample_rate": downsample, "voxel_size": [voxel, voxel, 4]}
    return {"num_iters": num_iters, "spatial_transform": dict(st),
            "hetero_fusion_block": {"input_dim": C, "mlp_dim": mlp_dim or C, "agent_size": L,
                                    "window_size": window, "dim_head": dim_head,
                                    "drop_out": 0.1, "architect_mode": arch,
                                    "spatial_transform": dict(st)}}


def rigid(yaw: float, tx: float, ty: float) -> Tensor:
    c, s = math.cos(yaw), math.sin(yaw)
    T = torch.eye(4, dtype=torch.float64)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1], T[0, 3], T[1, 3] = c, -s, s, c, tx, ty
    return T


def pairwise_from_poses(poses, L: int) -> Tensor:
    """pairwise[i, j] = inv(T_j) T_i (mixed/intermediate_fusion_dataset.py:163-202);
    identity for padded agents."""
    P = torch.eye(4, dtype=torch.float64).repeat(L, L, 1, 1)
    n = len(poses)
    for i in range(n):
        for j in range(n):
            if i != j:
                P[i, j] = torch.linalg.inv(poses[j]) @ poses[i]
    return P.to(torch.float32)


def _rs(seed: int):
    import numpy as np
    return np.random.RandomState(seed)  # legacy MT19937 stream: stable across numpy versions


def _randn(rs, *shape) -> Tensor:
    import numpy as np
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


def _uniform(rs, bound: float, *shape) -> Tensor:
    import numpy as np
    return torch.from_numpy(rs.uniform(-bound, bound, shape).astype(np.float32))


def synthetic_scene(L: int, C: int, H: int, W: int, modes, n_valid: int | None = None,
                    seed: int = 1, B: int = 1, yaw_step: float = 0.2, tx_step: float = 10.0,
                    ty_step: float = -6.0) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Seeded scene of SURVEY 8(d): x ~ N(0,1); T_0 = I, T_i = Rz(0.2 i) trans(10 i, -6 i) m.
    Padded agents (index >= n_valid) are all-zero maps with identity transforms, mode 0,
    mask 0 -- what regroup / the dataset produce (fuse_utils.py:8-61)."""
    rs = _rs(seed)
    n_valid = L if n_valid is None else n_valid
    x = _randn(rs, B, L, C, H, W)
    x[:, n_valid:] = 0
    poses = [rigid(yaw_step * i, tx_step * i, ty_step * i) for i in range(n_valid)]
    pw = pairwise_from_poses(poses, L)[None].repeat(B, 1, 1, 1, 1)
    mode = torch.tensor(list(modes), dtype=torch.int32)[None].repeat(B, 1)
    mode[:, n_valid:] = 0
    record_len = torch.full((B,), n_valid, dtype=torch.int64)
    mask = torch.zeros(B, L, dtype=torch.int64)
    mask[:, :n_valid] = 1
    return x, pw, mode, record_len, mask


def random_state_dict(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """Random weights with the reference's parameter names and shapes, drawn from
    a numpy legacy stream so that the golden generator (build container) and the tests (GPU
    box) regenerate bit-identical tensors without shipping them.  Scales follow the torch
    defaults (Linear: U(+-1/sqrt(fan_in)); relation matrices: xavier-uniform on the 4-D
    tensor; bias table: N(0,1); LayerNorm perturbed away from (1, 0) so that it matters)."""
    rs = _rs(seed)
    blk = cfg["hetero_fusion_block"]
    C, mlp, w, dh = blk["input_dim"], blk["mlp_dim"], blk["window_size"], blk["dim_head"]
    M = C // dh
    sd: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        sd[f"{name}.weight"] = _uniform(rs, b, out_f, in_f)
        if bias:
            sd[f"{name}.bias"] = _uniform(rs, b, out_f)

    def norm(name):
        sd[f"{name}.weight"] = 1 + 0.1 * _randn(rs, C)
        sd[f"{name}.bias"] = 0.1 * _randn(rs, C)

    p = "hetero_fusion_block"
    for which in ("window", "grid"):
        for t in range(NUM_TYPES):
            norm(f"{p}.{which}_norm.net.{t}")
            for nm in ("q", "k", "v"):
                lin(f"{p}.{which}_attention.{nm}_linears.{t}", C, C)
            lin(f"{p}.{which}_attention.a_linears.{t}.0", C, C)
            norm(f"{p}.{which}_ffd.norm.net.{t}")
            lin(f"{p}.{which}_ffd.fn.net.{t}.0", mlp, C)
            lin(f"{p}.{which}_ffd.fn.net.{t}.3", C, mlp)
        # torch xavier_uniform on (4, M, d, d): fan_in = M*d*d, fan_out = 4*d*d
        a = math.sqrt(6.0 / (M * dh * dh + NUM_TYPES ** 2 * dh * dh))
        sd[f"{p}.{which}_attention.relation_att"] = _uniform(rs, a, 4, M, dh, dh)
        sd[f"{p}.{which}_attention.relation_msg"] = _uniform(rs, a, 4, M, dh, dh)
        sd[f"{p}.{which}_attention.relative_position_bias_table.weight"] = \
            _randn(rs, (2 * w - 1) ** 2, M)
        sd[f"{p}.{which}_attention.relative_position_index"] = relative_position_index(w)
    for t in range(NUM_TYPES):
        lin(f"{p}.aggregate_fc.net.{t}.0", mlp, mlp * 3)
        lin(f"{p}.aggregate_fc.net.{t}.3", mlp, mlp)
    for t in range(NUM_TYPES):
        lin(f"mlp_head.net.{t}.0", C, C)
        lin(f"mlp_head.net.{t}.3", C, C)
    if blk["architect_mode"] == "parallel":
        lin(f"{p}.split_attn.fc1", C, C, bias=False)
        sd[f"{p}.split_attn.bn1.weight"] = 1 + 0.1 * _randn(rs, C)
        sd[f"{p}.split_attn.bn1.bias"] = 0.1 * _randn(rs, C)
        lin(f"{p}.split_attn.fc2", 2 * C, C, bias=False)
    return sd
