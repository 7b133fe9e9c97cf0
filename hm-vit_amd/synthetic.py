"""Seeded synthetic workloads for bench.py and the tuning tools: the HeteroFusion config dict, a scene of
random agent maps with rigid poses T_0 = I, T_i = Rz(yaw_step i) trans(tx_step i, ty_step i) metres, and
``pairwise[i, j] = inv(T_j) T_i`` as the dataset builds it (mixed/intermediate_fusion_dataset.py:163-202).  Weights are
the module's own default initialisation under ``torch.manual_seed``.  No dependency on oracle/."""
from __future__ import annotations

import math

import torch


def make_config(C: int, window: int, L: int, voxel: float = 0.4, downsample: int = 4, num_iters: int = 2, dim_head: int = 32,
                arch: str = "sequential", mlp_dim: int | None = None) -> dict:
    st = {"downsample_rate": downsample, "voxel_size": [voxel, voxel, 4]}
    return {"num_iters": num_iters, "spatial_transform": dict(st),
            "hetero_fusion_block": {"input_dim": C, "mlp_dim": mlp_dim or C, "agent_size": L, "window_size": window,
                                    "dim_head": dim_head, "drop_out": 0.1, "architect_mode": arch,
                                    "spatial_transform": dict(st)}}


def rigid(yaw: float, tx: float, ty: float) -> torch.Tensor:
    c, s = math.cos(yaw), math.sin(yaw)
    T = torch.eye(4, dtype=torch.float64)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1], T[0, 3], T[1, 3] = c, -s, s, c, tx, ty
    return T


def pairwise_from_poses(poses, L: int) -> torch.Tensor:
    P = torch.eye(4, dtype=torch.float64).repeat(L, L, 1, 1)
    for i in range(len(poses)):
        for j in range(len(poses)):
            if i != j:
                P[i, j] = torch.linalg.inv(poses[j]) @ poses[i]
    return P.to(torch.float32)


def synthetic_scene(L: int, C: int, H: int, W: int, modes, seed: int = 1, B: int = 1, yaw_step: float = 0.2,
                    tx_step: float = 10.0, ty_step: float = -6.0):
    """(x (B, L, C, H, W) ~ N(0, 1), pairwise_t_matrix (B, L, L, 4, 4), mode (B, L), record_len (B,), mask (B, L))."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, C, H, W, generator=gen)
    poses = [rigid(yaw_step * i, tx_step * i, ty_step * i) for i in range(L)]
    pw = pairwise_from_poses(poses, L)[None].repeat(B, 1, 1, 1, 1)
    mode = torch.tensor(list(modes), dtype=torch.int32)[None].repeat(B, 1)
    record_len = torch.full((B,), L, dtype=torch.int64)
    mask = torch.ones(B, L, dtype=torch.int64)
    return x, pw, mode, record_len, mask


def seeded_fusion(cfg: dict, precision: str = "split", seed: int = 0):
    """HeteroFusion with its default initialisation drawn under ``torch.manual_seed(seed)`` (default init, bias
    table ~ N(0, 1) = nn.Embedding's default), LayerNorm affine perturbed away from (1, 0) so that it matters."""
    from .fusion import HeteroFusion
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = HeteroFusion(cfg, precision=precision)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(0.1 * torch.randn_like(m.weight))
                m.bias.add_(0.1 * torch.randn_like(m.bias))
    torch.random.set_rng_state(state)
    return net
