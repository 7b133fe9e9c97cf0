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


# ---------------------------------------------------------------------------------------------------------------------
# camera agents of the synthetic scenes (trainer.py / replay.py / bench.py --train): the CVT camera lift in the model's camera
# slot (hm-vit_amd/camera.py: CvtCameraEncoder), images, pinhole intrinsics and camera -> ego extrinsics of four cameras
# ---------------------------------------------------------------------------------------------------------------------
def camera_config(image: int = 64, num_layers: int = 18, bev_h: int = 32, bev_w: int = 32, dim: int = 128) -> dict:
    """CvtCameraEncoder config with the structure of ``opcamera/cvt.yaml:49-84``: ResNet-`num_layers` on `image` x `image` cameras,
    pyramid levels 1 and 3 (``id_pick``), (bev_h / 8) x (bev_w / 8) BEV queries, ``NaiveDecoder(dim -> [256, 256])`` with two x2
    up-samplings, i.e. a (256, bev_h / 2, bev_w / 2) BEV map per camera agent."""
    enc = {"num_layers": num_layers, "pretrained": False, "image_height": image, "image_width": image, "id_pick": [1, 3]}
    h1, h3 = image // 8, image // 32
    cvm = {"dim": dim, "middle": [2, 2], "backbone_output_shape": [[1, 1, 4, 128, h1, h1], [1, 1, 4, 512, h3, h3]],
           "bev_embedding": {"sigma": 1.0, "bev_height": bev_h, "bev_width": bev_w, "h_meters": 100.0, "w_meters": 100.0, "offset": 0.0,
                             "decoder_blocks": [128, 128, 64]},
           "cross_view": {"image_width": image, "image_height": image, "no_image_features": False, "heads": 4, "dim_head": 32,
                          "qkv_bias": True, "skip": True}}
    return {"encoder": enc, "cvm": cvm, "decoder": {"input_dim": dim, "num_layer": 2, "num_ch_dec": [256, 256]}}


def synthetic_cameras(n_agents: int, image: int, seed: int = 0) -> dict:
    """``camera`` (N, 4, image, image, 3) ~ N(0, 1) (the dataset hands over ImageNet-normalised RGB, ``rgb_preprocessor.py:16-30``),
    ``intrinsic`` (N, 4, 3, 3) with the focal length of the reference's test yaml (335.64 px at 800 px, rescaled) and
    ``extrinsic`` (N, 4, 4, 4) camera -> ego of four cameras looking forward / left / back / right from the roof."""
    gen = torch.Generator().manual_seed(seed)
    cam = torch.randn(n_agents, 4, image, image, 3, generator=gen)
    f = 335.64 * image / 800.0
    K = torch.tensor([[f, 0.0, image / 2], [0.0, f, image / 2], [0.0, 0.0, 1.0]])
    ext = torch.zeros(n_agents, 4, 4, 4)
    for a in range(n_agents):
        for c in range(4):
            yaw = c * math.pi / 2 + 0.05 * float(torch.randn(1, generator=gen))
            cs, sn = math.cos(yaw), math.sin(yaw)
            T = torch.eye(4)
            # camera axes (x right, y down, z forward) expressed in the ego frame (x forward, y left, z up)
            T[:3, :3] = torch.tensor([[sn, 0.0, cs], [-cs, 0.0, sn], [0.0, -1.0, 0.0]])
            T[:3, 3] = torch.tensor([1.5 * cs, 1.5 * sn, 1.6 + 0.02 * a])
            ext[a, c] = T
    return {"camera": cam, "intrinsic": K.repeat(n_agents, 4, 1, 1), "extrinsic": ext, "cav2cam_extrinsic": ext.clone()}


# ---------------------------------------------------------------------------------------------------------------------
# LiDAR agents (encoder benchmarks): `n_per_agent` unique random cells of an (nx, ny) pillar grid per agent,
# voxel_features ~ points inside their cell (x, y, z, intensity), 1..32 points per pillar, zero padded
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_pillars(n_agents: int, n_per_agent: int, nx: int, ny: int, lidar_args: dict, seed: int = 3):
    """(voxel_features (Nv, 32, 4) f32, voxel_coords (Nv, 4) int32 [agent, z, y, x], voxel_num_points (Nv,) int32)."""
    gen = torch.Generator().manual_seed(seed)
    vx, vy, vz = lidar_args["voxel_size"]
    x0, y0, z0 = lidar_args["lidar_range"][:3]
    feats, coords, counts = [], [], []
    for a in range(n_agents):
        cells = torch.randperm(nx * ny, generator=gen)[:n_per_agent]
        cy, cx = cells // nx, cells % nx
        n_pts = torch.randint(1, 33, (n_per_agent,), generator=gen)
        u = torch.rand(n_per_agent, 32, 4, generator=gen)
        pts = torch.stack([x0 + (cx[:, None] + u[..., 0]) * vx, y0 + (cy[:, None] + u[..., 1]) * vy, z0 + u[..., 2] * vz, u[..., 3]], -1)
        pts = pts * (torch.arange(32)[None, :] < n_pts[:, None])[..., None]
        feats.append(pts.float())
        coords.append(torch.stack([torch.full_like(cy, a), torch.zeros_like(cy), cy, cx], 1))
        counts.append(n_pts)
    return torch.cat(feats), torch.cat(coords).int(), torch.cat(counts).int()


def resnet_trunk_flops(num_layers: int, image: int) -> float:
    """FLOPs (2 per MAC) of the convolutions of a torchvision-style ResNet-18 / 34 BasicBlock trunk on one image x image input:
    7 x 7 / 2 stem, 3 x 3 / 2 max-pool, stages of [64, 128, 256, 512] channels (the pricing of bench.py's camera-encoder line)."""
    blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}[num_layers]
    hw = (image // 2) ** 2
    flops = 2.0 * 64 * 3 * 49 * hw
    hw //= 4
    cin = 64
    for stage, (n, c) in enumerate(zip(blocks, [64, 128, 256, 512])):
        for b in range(n):
            stride = 2 if (stage > 0 and b == 0) else 1
            hw_out = hw // (stride * stride)
            flops += 2.0 * c * cin * 9 * hw_out + 2.0 * c * c * 9 * hw_out
            if stride != 1 or cin != c:
                flops += 2.0 * c * cin * hw_out
            cin, hw = c, hw_out
    return flops
